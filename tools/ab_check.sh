O=gpurun_out/quick; mkdir -p $O
run() { # name, lib, args
  if [ -n "$2" ]; then export NEDDF_LIB_PATH=$PWD/tools/bin/libneddf_hip_$2.so; else unset NEDDF_LIB_PATH; fi
  python bench.py $3 --steps 3 --warmup 1 --no-cpu-baseline > $O/$1.json 2>$O/$1.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$1.json").read().strip().split("\n")[-1])
    r=d["roofline"]
    print("$1", round(d["value"]), round(d["ms_per_step"],1), r.get("avg_launch_ms"), (r.get("colour_kernel") or {}).get("avg_launch_ms"), d.get("psnr_vs_oracle_db"))
except Exception as e:
    print("$1 FAILED", e, open("$O/$1.err").read()[-600:])
PY
}
for rep in 1 2; do
run split_tree_$rep "" "--dtype f16_split"
run split_S_$rep S "--dtype f16_split"
done
