#!/bin/bash
# Round 5: the two-pass reverse kernel (ddf_rev2_kernel) in its shapes against the shipped 64-point kernel, same call.
O=gpurun_out/r05b
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1 NEDDF_BENCH_PMC=0
for cfg in "2x2x4 0" "4x2x4 0" "2x3x4 0" "2x4x4 0" "2x2x4 1" "2x2x4 0"; do
  set -- $cfg
  NEDDF_REV_GEO_BF16=$1 NEDDF_REV2=$2 timeout 300 python bench.py --dtype bf16 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_bf16_$1_$2.json 2>$O/bench_bf16_$1_$2.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_bf16_$1_$2.json")); r=d["roofline"]
    print("$1 rev2=$2", round(d["value"]), "ms/step %.1f"%d["ms_per_step"], "ddf %.2f ms frac %.3f"%(r["avg_launch_ms"], r["frac"]), "col %.2f"%r["colour_kernel"]["avg_launch_ms"], "psnr %.1f"%d.get("psnr_vs_oracle_db"))
except Exception as e: print("$1 $2 ERR", e); print(open("$O/bench_bf16_$1_$2.err").read()[-800:])
PY
done
for geo in 2x3x4 2x4x4; do
NEDDF_REV_GEO_BF16=$geo timeout 600 python -m pytest tests/test_gpu_c5.py -x -q -m gpu > $O/pytest_c5_$geo.txt 2>&1; echo "pytest $geo rc=$?"; tail -2 $O/pytest_c5_$geo.txt
done
