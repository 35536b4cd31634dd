#!/bin/bash
O=gpurun_out/r3ae; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_train.py -x -q -m gpu > $O/tests_train.log 2>&1; grep -E "passed|failed|error" $O/tests_train.log | tail -3; grep -E "^FAILED|^E  " $O/tests_train.log | head
for v in pmnt new pmnt new; do
  if [ $v = new ]; then L=$PWD/neddf_amd/csrc/libneddf_hip.so; else L=$PWD/tools/bin/libneddf_hip_$v.so; fi
  NEDDF_LIB_PATH=$L timeout 120 python bench.py --workload train --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v', round(d['ms_per_step'],2), d['final_loss'])"; done | tee $O/ab.txt
