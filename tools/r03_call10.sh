#!/bin/bash
O=gpurun_out/r3j; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_train.py -x -q -m gpu > $O/tests_train.log 2>&1; grep -E "passed|failed|error" $O/tests_train.log | tail -3; grep -E "^FAILED|^E  " $O/tests_train.log | head
for u in 0; do NEDDF_TRAIN_UNFUSED=$u python bench.py --workload train --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('fp32 unfused=$u', round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['achieved'],1), d['final_loss'])"; done
NEDDF_LIB_PATH=$PWD/tools/bin/libneddf_hip_prev.so python bench.py --workload train --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('prev lib', round(d['value']), round(d['ms_per_step'],2))"
python bench.py --workload train --dtype f16_split --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('split', round(d['value']), round(d['ms_per_step'],2))"
