#!/bin/bash
# split-fp16 fields wider than 256 on the fused chains over the width (default) against the blocked per-layer route (NEDDF_TRAIN_WIDE_FUSED=0)
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "above_width_256 or random_architectures or other_widths" 2>&1 | tail -3
for v in 1 0 1 0; do
  NEDDF_PROBE_DTYPE=f16_split NEDDF_TRAIN_WIDE_FUSED=$v python tools/train_wide_probe.py 512 2>&1 | tail -1 | sed "s/^/WIDE_FUSED=$v  /"
done
