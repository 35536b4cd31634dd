#!/bin/bash
# Round 5: SQ / TCC counters of the bf16 distance kernel in its shipped shape (64-point tiles, y' as bf16) and as ddf_rev2_kernel
# (128-point tiles in two column passes, y' as eight bits; and 64-point tiles at three workgroups per CU), one 2^23-point launch each.
ROOT=$PWD
O=$ROOT/gpurun_out/r05e
mkdir -p $O
export TMPDIR=/tmp NEDDF_PROBE_DTYPE=bf16
cd /tmp
for geo in 2x2x4 4x2x4 2x3x4; do
  for c in FETCH_SIZE WRITE_SIZE; do
    NEDDF_REV_GEO_BF16=$geo rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$geo/$c -- python $ROOT/tools/pmc_probe.py 1 > $O/log_${geo}_$c.txt 2>&1
  done
  NEDDF_REV_GEO_BF16=$geo rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_$geo/sq -- python $ROOT/tools/pmc_probe.py 1 > $O/log_${geo}_sq.txt 2>&1
  NEDDF_REV_GEO_BF16=$geo rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_$geo/sq2 -- python $ROOT/tools/pmc_probe.py 1 > $O/log_${geo}_sq2.txt 2>&1
  cd $ROOT; python tools/pmc_summary.py $O/pmc_$geo | grep -i "kernel,counter\|ddf_rev" > $O/pmc_${geo}_summary.csv; cd /tmp
  rm -rf $O/pmc_$geo
done
cat $O/pmc_*_summary.csv | head -80
