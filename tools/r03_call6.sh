#!/bin/bash
O=gpurun_out/r3f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_c5.py -x -q -m gpu > $O/tests.log 2>&1; grep -E "passed|failed|error" $O/tests.log | tail -3; grep -E "^FAILED|^E  " $O/tests.log | head
echo "== previous library"; NEDDF_LIB_PATH=$PWD/tools/bin/libneddf_hip_prev.so python tools/act_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/act_prev.txt
echo "== this tree"; python tools/act_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/act_new.txt
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c2.json 2>$O/bench_c2.err; python -c "
import json; d=json.loads(open('$O/bench_c2.json').read().strip().split('\n')[-1]); r=d['roofline']; print('c2', round(d['value']), r['avg_launch_ms'], round(r['frac'],4), r['colour_kernel']['avg_launch_ms'])" || tail -5 $O/bench_c2.err
