#!/bin/bash
O=$PWD/gpurun_out/r3i; mkdir -p $O
export TMPDIR=/tmp; ROOT=$PWD
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_train/sq -- python $ROOT/bench.py --workload train --steps 2 --warmup 1 > $O/pmc_train_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_train/lds -- python $ROOT/bench.py --workload train --steps 2 --warmup 1 > $O/pmc_train_lds.log 2>&1
tail -3 $O/pmc_train_lds.log
cd $ROOT
python tools/pmc_summary.py $O/pmc_train > $O/pmc_train_summary.csv; grep -E "dw_tile_kernel<8>|mlp_backward|mlp_forward" $O/pmc_train_summary.csv
