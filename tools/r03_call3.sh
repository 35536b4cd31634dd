#!/bin/bash
O=gpurun_out/r3c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "synth" > $O/tests_synth.log 2>&1; tail -15 $O/tests_synth.log
python tools/r03_margins.py neddf_w128 neddf_w192 neddf_w384 neddf_skips2 > $O/margins.txt 2>&1; cat $O/margins.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests_full.log 2>&1; grep -E "passed|failed|error" $O/tests_full.log | tail -3
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c2.json 2>$O/bench_c2.err; python -c "
import json; d=json.loads(open('$O/bench_c2.json').read().strip().split('\n')[-1]); r=d['roofline']; print('c2', round(d['value']), r['avg_launch_ms'], r['frac'], r['colour_kernel'])"
