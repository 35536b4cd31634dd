#!/bin/bash
O=gpurun_out/r3ag; mkdir -p $O
L=$PWD/tools/bin/libneddf_hip_fastact.so
NEDDF_LIB_PATH=$L timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_c5.py -q -m gpu > $O/tests_fast.log 2>&1; grep -E "passed|failed|error" $O/tests_fast.log | tail -3; grep -E "^FAILED" $O/tests_fast.log | head -20
for v in cur fast cur fast; do
  if [ $v = cur ]; then LL=$PWD/neddf_amd/csrc/libneddf_hip.so; else LL=$L; fi
  NEDDF_LIB_PATH=$LL timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('$v f32', round(d['value']), 'ddf', round(r['avg_launch_ms'],2), 'col', round(r['colour_kernel']['avg_launch_ms'],2), 'psnr', round(d['psnr_vs_oracle_db'],1), d['parity_sample']['gate_margin'], 'split', round(d['alt_operand_policy']['value']))"; done | tee $O/bench.txt
