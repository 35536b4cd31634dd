#!/bin/bash
O=gpurun_out/r3e; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests_full.log 2>&1; grep -E "passed|failed|error" $O/tests_full.log | tail -3; grep -E "^FAILED|Error" $O/tests_full.log | head
python bench.py --steps 5 --warmup 2 > $O/bench_c2.json 2>$O/bench_c2.err; python -c "
import json; d=json.loads(open('$O/bench_c2.json').read().strip().split('\n')[-1]); r=d['roofline']; print('c2', round(d['value']), r['avg_launch_ms'], round(r['frac'],4), r['colour_kernel'], d['psnr_vs_oracle_db'], d['parity_sample']['gate_margin']); print(json.dumps(d['cpu_baseline'])[:1500])" || tail -5 $O/bench_c2.err
python tools/r03_margins.py > $O/margins.txt 2>&1; cat $O/margins.txt | head -4
