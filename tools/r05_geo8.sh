#!/bin/bash
# Round 5 probe: the fp32 field kernels with eight waves on the 64-row tile (four waves per SIMD instead of two).
# The two switches were local to the probe and are NOT in the tree (lab notebook R5.17: distance kernel -3.7 %, colour kernel +0.7 %):
#   geo_rev():   g[0] = parse_geo("NEDDF_REV_GEO_F32", Geo{ 2, 2, 4 }, { { 2, 2, 4 }, { 2, 2, 8 } });
#   launch_ddf_rev():  NEDDF_GEO_CASE(2, 2, 8) return launch_ddf_rev_t<2, 8, 2, OpsF32>(a, grid, s);
#   geo_col(0):  parse_geo("NEDDF_F32_COL_GEO", Geo{ 2, 2, 4 }, { { 2, 2, 4 }, { 2, 2, 8 } });   launch_col(): launch_col_g<2, 2, 8, OpsF32>
ROOT=$PWD
O=$ROOT/gpurun_out/r5geo8
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1 NEDDF_BENCH_PMC=0
run() {   # label, env...
  label=$1; shift
  env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_$label.json 2>$O/err_$label.txt
  python - $O/bench_$label.json $label <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print("%-22s %9d rays/s  distance kernel %.2f ms (frac %.3f)  colour kernel %.2f ms  parity margin %.3f  psnr %.1f" % (sys.argv[2], d["value"], r["avg_launch_ms"], r["frac"], r["colour_kernel"]["avg_launch_ms"], max(d["parity_sample"]["gate_margin"].values()), d["psnr_vs_oracle_db"]))
except Exception as e:
    print(sys.argv[2], "failed:", e); print(open(sys.argv[1].replace("bench_", "err_").replace(".json", ".txt")).read()[-800:])
PY
}
run shipped A=1
run rev_2x2x8 NEDDF_REV_GEO_F32=2x2x8
run col_2x2x8 NEDDF_F32_COL_GEO=2x2x8
run both_2x2x8 NEDDF_REV_GEO_F32=2x2x8 NEDDF_F32_COL_GEO=2x2x8
run shipped_again A=1
