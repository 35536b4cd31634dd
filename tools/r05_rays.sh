#!/bin/bash
# Round 5: sample points from the rays (NEDDF_RAYS_IN_FIELD, default on) -- bit-identity test, the hot-path parity files, A/B of both routes.
O=gpurun_out/r05g
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1 NEDDF_BENCH_PMC=0
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "from_rays or bunny or end_to_end or c1_ or c2_ or image_small or ragged or neus or nerf_render or invariants or full_size" > $O/pytest_rays.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_rays.txt
line() { python - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]; st=d["stage_ms_per_step"]
    print(sys.argv[2], round(d["value"]), "ms/step %.2f"%d["ms_per_step"], "ddf %.2f ms frac %.3f"%(r["avg_launch_ms"], r["frac"]), "col %.2f"%r["colour_kernel"]["avg_launch_ms"], "sampling", st.get("sampling"), "psnr %.1f"%d.get("psnr_vs_oracle_db"))
except Exception as e: print(sys.argv[2], "ERR", e)
PY
}
for rep in 1 2; do
for dt in f32 bf16; do
  st=4; [ $dt = f32 ] && st=3
  NEDDF_RAYS_IN_FIELD=0 timeout 300 python bench.py --dtype $dt --steps $st --warmup 1 --no-cpu-baseline > $O/b_${dt}_tensors_$rep.json 2>$O/err.txt; line $O/b_${dt}_tensors_$rep.json "$dt tensors"
  timeout 300 python bench.py --dtype $dt --steps $st --warmup 1 --no-cpu-baseline > $O/b_${dt}_rays_$rep.json 2>$O/err.txt; line $O/b_${dt}_rays_$rep.json "$dt rays   "
done
done
