#!/bin/bash
O=gpurun_out/r3al; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py -q -m gpu > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['value']), d['roofline']['frac'])"
