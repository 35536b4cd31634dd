#!/bin/bash
# Round 5: are all workgroup slots of the persistent field grids occupied?  tools/residency_probe.hip (where the workgroups of a
# 512-workgroup grid with the field kernels' footprint land) + bench.py with NEDDF_GRID_SLACK_PCT more workgroups than slots.
# Build the probe first (in the build container; tools/bin/ travels with the snapshot):
#   mkdir -p tools/bin && hipcc --offload-arch=gfx950 -O2 -o tools/bin/residency_probe tools/residency_probe.hip
ROOT=$PWD
O=$ROOT/gpurun_out/r5res
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1 NEDDF_BENCH_PMC=0
tools/bin/residency_probe | tee $O/residency_probe.txt
tools/bin/residency_probe 35000 | tee -a $O/residency_probe.txt
for p in 0 12 25 50 100 0; do
  NEDDF_GRID_SLACK_PCT=$p python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_f32_slack$p.json 2>$O/err_$p.txt
  python - $O/bench_f32_slack$p.json $p <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print("fp32 slack %s%%:" % sys.argv[2], round(d["value"]), "rays/s  ddf launch %.2f ms (frac %.3f)  col launch %.2f ms  parity margin %.3f" % (r["avg_launch_ms"], r["frac"], r["colour_kernel"]["avg_launch_ms"], max(d["parity_sample"]["gate_margin"].values())))
except Exception as e:
    print("slack", sys.argv[2], "failed:", e)
PY
done
for dt in bf16 f16_split; do
  for p in 0 25 50; do
    NEDDF_GRID_SLACK_PCT=$p python bench.py --dtype $dt --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_${dt}_slack$p.json 2>/dev/null
    python - $O/bench_${dt}_slack$p.json $p $dt <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print(sys.argv[3], "slack %s%%:" % sys.argv[2], round(d["value"]), "rays/s  ddf launch %.2f ms (frac %.3f)  col launch %.2f ms" % (r["avg_launch_ms"], r["frac"], r["colour_kernel"]["avg_launch_ms"]))
except Exception as e:
    print("slack", sys.argv[2], "failed:", e)
PY
  done
done
