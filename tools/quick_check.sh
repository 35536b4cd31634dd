# A/B on the GPU box: bench lines of library variants (tools/bin/libneddf_hip_<X>.so, built by hand) against the tree's build
O=gpurun_out/quick; mkdir -p $O
run() { # name, lib, args
  if [ -n "$2" ]; then export NEDDF_LIB_PATH=$PWD/tools/bin/libneddf_hip_$2.so; else unset NEDDF_LIB_PATH; fi
  python bench.py $3 --steps 3 --warmup 1 --no-cpu-baseline > $O/$1.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/$1.json").read().strip().split("\n")[-1])
print("$1", round(d["value"]), round(d["ms_per_step"],1), round(d["roofline"]["avg_launch_ms"],3), round(d["roofline"]["colour_kernel"]["avg_launch_ms"],3), d.get("psnr_vs_oracle_db"))
PY
}
for rep in 1 2; do
run f32_tree_$rep "" ""
run f32_R_$rep R ""
done
run c3_tree "" "--workload c3"
run c3_R R "--workload c3"
