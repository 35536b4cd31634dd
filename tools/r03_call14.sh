#!/bin/bash
O=gpurun_out/r3x; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_train.py -x -q -m gpu > $O/tests_train.log 2>&1; grep -E "passed|failed|error" $O/tests_train.log | tail -3; grep -E "^FAILED|^E  " $O/tests_train.log | head
for i in 1 2; do timeout 120 python bench.py --workload train --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('fp32', round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['achieved'],1), d['final_loss'])"; done
NEDDF_LIB_PATH=$PWD/tools/bin/libneddf_hip_prev.so timeout 120 python bench.py --workload train --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('prev lib', round(d['value']), round(d['ms_per_step'],2))"
