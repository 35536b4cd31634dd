#!/bin/bash
# Tile-geometry sweep on the headline workload (one line per variant): tools/geo_sweep.sh [f32|16bit|all]
cd "$(dirname "$0")/.."
run() {  # dtype, env assignment
  out=$(env $2 python bench.py --dtype $1 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1)
  python - "$1" "$2" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[3])
r = d["roofline"]
print("%-10s %-28s %9.0f rays/s  ddf %.3f ms (%.0f TF, %.1f%%)  col %.3f ms (%.0f TF)  psnr %.1f dB" % (
    sys.argv[1], sys.argv[2], d["value"], r["avg_launch_ms"], r["achieved"], 100 * r["frac"], r["colour_kernel"]["avg_launch_ms"],
    r["colour_kernel"]["achieved"], d.get("psnr_vs_oracle_db", float("nan"))))
PY
}
what=${1:-all}
if [ "$what" = f32 ] || [ "$what" = all ]; then
  run f32 NEDDF_F32_GEO=2x2x4
  run f32 NEDDF_F32_GEO=1x4x4
  run f32 NEDDF_F32_GEO=1x3x4
fi
if [ "$what" = 16bit ] || [ "$what" = all ]; then
  run f16_split NEDDF_SPLIT_GEO=4x1x8
  run f16_split NEDDF_SPLIT_GEO=2x2x4
  run bf16 NEDDF_BF16_GEO=4x2x8
  run bf16 NEDDF_BF16_GEO=4x1x8
  run bf16 NEDDF_BF16_GEO=4x2x4
  run bf16 NEDDF_BF16_GEO=2x2x4
fi
