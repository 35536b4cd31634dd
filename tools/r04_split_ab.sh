#!/bin/bash
# split-fp16 row stride: bench + kernel times (run at the tree to measure; the baseline numbers are profiles/r04_bench_c2_f16_split.json)
mkdir -p gpurun_out/r4split
python bench.py --dtype f16_split --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r4split/c2_split.json 2>/dev/null
python bench.py --workload c3 --dtype f16_split --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r4split/c3_split.json 2>/dev/null
python - <<'PY'
import json
for n in ("c2_split","c3_split"):
    d=json.loads(open("gpurun_out/r4split/%s.json"%n).read().strip().splitlines()[-1]); r=d["roofline"]
    print(n, round(d["value"]), d["ms_per_step"], r["avg_launch_ms"], r["frac"], r["colour_kernel"]["avg_launch_ms"], d.get("psnr_vs_oracle_db"))
PY
