#!/bin/bash
# Round 5: phase-offset twin teams (NEDDF_REV_TEAMS: bit 0 bf16, bit 1 split fp16) against two independent workgroups per CU, same call.
O=gpurun_out/r05c
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1 NEDDF_BENCH_PMC=0
for cfg in "bf16 0" "bf16 3" "f16_split 0" "f16_split 3" "bf16 0" "bf16 3"; do
  set -- $cfg
  NEDDF_REV_TEAMS=$2 timeout 300 python bench.py --dtype $1 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_$1_$2.json 2>$O/bench_$1_$2.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$1_$2.json")); r=d["roofline"]
    print("$1 teams=$2", round(d["value"]), "ms/step %.1f"%d["ms_per_step"], "ddf %.2f ms frac %.3f"%(r["avg_launch_ms"], r["frac"]), "col %.2f"%r["colour_kernel"]["avg_launch_ms"], "psnr %.1f"%d.get("psnr_vs_oracle_db"))
except Exception as e: print("$1 $2 ERR", e); print(open("$O/bench_$1_$2.err").read()[-800:])
PY
done
NEDDF_REV_TEAMS=3 timeout 600 python -m pytest tests/test_gpu_c5.py -x -q -m gpu > $O/pytest_c5_teams.txt 2>&1; echo "pytest teams rc=$?"; tail -3 $O/pytest_c5_teams.txt
