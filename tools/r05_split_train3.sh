#!/bin/bash
# split-fp16 training: the weight gradients as one job-parallel launch per pass (NEDDF_TRAIN_SPLIT_DW_JOBS=0: one launch per product)
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python bench.py --workload train --steps 8 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32: %d rays/s  %.2f ms/step' % (d['value'], d['ms_per_step']))"
for v in 1 0 1 0 1 0; do
  NEDDF_TRAIN_SPLIT_DW_JOBS=$v python bench.py --workload train --dtype f16_split --steps 16 --warmup 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('SPLIT_DW_JOBS=$v: %d rays/s  %.2f ms/step  final loss %.6f' % (d['value'], d['ms_per_step'], d['final_loss']))"
done
python bench.py --workload train --steps 8 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32: %d rays/s  %.2f ms/step' % (d['value'], d['ms_per_step']))"
