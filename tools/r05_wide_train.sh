#!/bin/bash
# Round 5: fields wider than 256 on the 512-wide fused training chains (fp32) (NEDDF_TRAIN_WIDE_FUSED=1) against the blocked per-layer route (the default)
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for w in 512 384; do
  for v in 1 0 1 0; do
    NEDDF_TRAIN_WIDE_FUSED=$v python tools/train_wide_probe.py $w 2>&1 | tail -1 | sed "s/^/WIDE_FUSED=$v  /"
  done
done
python tools/train_wide_probe.py 256 2>&1 | tail -1
NEDDF_PROBE_DTYPE=f16_split python tools/train_wide_probe.py 256 2>&1 | tail -1
NEDDF_PROBE_DTYPE=f16_split python tools/train_wide_probe.py 512 2>&1 | tail -1
