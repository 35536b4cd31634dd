#!/bin/bash
# Round 5: fp32 fields wider than 256 on the 512-wide fused training chains (default) against the blocked per-layer route
# (NEDDF_TRAIN_WIDE_FUSED=0) and against the fused chains with the job-parallel weight-gradient launch (NEDDF_TRAIN_WIDE_DW_JOBS=1)
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for w in 512 384; do
  for v in 1 0 1 0; do
    NEDDF_TRAIN_WIDE_FUSED=$v python tools/train_wide_probe.py $w 2>&1 | tail -1 | sed "s/^/WIDE_FUSED=$v  /"
  done
done
NEDDF_TRAIN_WIDE_DW_JOBS=1 python tools/train_wide_probe.py 512 2>&1 | tail -1 | sed "s/^/WIDE_DW_JOBS=1  /"
python tools/train_wide_probe.py 256 2>&1 | tail -1
NEDDF_PROBE_DTYPE=f16_split python tools/train_wide_probe.py 256 2>&1 | tail -1
NEDDF_PROBE_DTYPE=f16_split python tools/train_wide_probe.py 512 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "not run_script and not trainer_run" 2>&1 | tail -3
