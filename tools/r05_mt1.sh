#!/bin/bash
# probe (lab notebook R5.20; the switch is NOT in the tree: template flag SMALL on mlp_forward_kernel / mlp_backward_kernel, MT = 1,
# __launch_bounds__(256, 3)): the 256-wide fp32 training chains on 32-row tiles, three workgroups per CU (NEDDF_TRAIN_MT1=1)
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for v in 0 1 0 1; do
  NEDDF_TRAIN_MT1=$v python bench.py --workload train --steps 12 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TRAIN_MT1=$v: %d rays/s  %.2f ms/step  final loss %.6f' % (d['value'], d['ms_per_step'], d['final_loss']))"
done
NEDDF_TRAIN_MT1=1 timeout 300 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "fp32 and (field_forward_backward or train_step)" 2>&1 | tail -2
