#!/bin/bash
# heads' weight gradients inside the backward chain's prologue against the stand-alone kernel (one call, same box)
for mode in fused separate fused separate; do
  NEDDF_TRAIN_HEADS_DW=$mode python bench.py --workload train --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', d['ms_per_step'], d['value'])"
done
