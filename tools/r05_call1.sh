#!/bin/bash
# Round-5 first GPU call: the reordered suite once (unbuffered), the default bench invocation, and same-box baselines of the 16-bit policies.
O=gpurun_out/r05a
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu_run1.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_run1.txt
tail -3 $O/pytest_gpu_run1.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 300 env NEDDF_BENCH_PMC=0 python bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c2_bf16.json 2>/dev/null
timeout 300 env NEDDF_BENCH_PMC=0 python bench.py --dtype f16_split --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_c2_f16_split.json 2>/dev/null
python - <<'PY'
import json
for n in ("bench_default","bench_c2_bf16","bench_c2_f16_split"):
    try:
        d=json.load(open("gpurun_out/r05a/%s.json"%n)); r=d["roofline"]
        print(n, round(d["value"]), "ms/step %.1f"%d["ms_per_step"], "ddf %.2f ms frac %.3f"%(r["avg_launch_ms"], r["frac"]), "col %.2f"%r["colour_kernel"]["avg_launch_ms"], "traffic", r.get("traffic"), r.get("traffic_source","")[:60])
    except Exception as e: print(n, "ERR", e)
PY
