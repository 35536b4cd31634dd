#!/bin/bash
# fp32 training step at width 256: job-parallel weight gradients (default) against one point-major launch per product (NEDDF_TRAIN_DW_JOBS=0)
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for v in 1 0 1 0; do
  NEDDF_TRAIN_DW_JOBS=$v python bench.py --workload train --steps 12 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('DW_JOBS=$v: %d rays/s  %.2f ms/step  final loss %.6f' % (d['value'], d['ms_per_step'], d['final_loss']))"
done
