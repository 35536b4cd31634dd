#!/bin/bash
O=$PWD/gpurun_out/r3z; mkdir -p $O; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_train -o t -- python $ROOT/bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_train.log 2>&1
cd $ROOT
ls $O/pmc_train | head
