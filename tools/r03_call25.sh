#!/bin/bash
O=gpurun_out/r3ai; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_c5.py -q -m gpu -x > $O/tests.log 2>&1; grep -E "passed|failed|error" $O/tests.log | tail -3; grep -E "^FAILED|^E  " $O/tests.log | head -20
for v in fast new fast new; do
  if [ $v = new ]; then LL=$PWD/neddf_amd/csrc/libneddf_hip.so; else LL=$PWD/tools/bin/libneddf_hip_fastact.so; fi
  NEDDF_LIB_PATH=$LL timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('$v f32', round(d['value']), 'ddf', round(r['avg_launch_ms'],2), 'col', round(r['colour_kernel']['avg_launch_ms'],2), 'psnr', round(d['psnr_vs_oracle_db'],1), 'split', round(d['alt_operand_policy']['value']))"; done | tee $O/bench.txt
for v in fast new; do
  if [ $v = new ]; then LL=$PWD/neddf_amd/csrc/libneddf_hip.so; else LL=$PWD/tools/bin/libneddf_hip_fastact.so; fi
  NEDDF_LIB_PATH=$LL timeout 200 python bench.py --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']; print('$v bf16', round(d['value']), 'ddf', round(r['avg_launch_ms'],2), 'col', round(r['colour_kernel']['avg_launch_ms'],2), 'psnr', round(d['psnr_vs_oracle_db'],1))"; done | tee -a $O/bench.txt
