O=gpurun_out/r04/skew; mkdir -p $O
run() { # name env args
  env NEDDF_FUSED=0 $2 python bench.py $3 --steps 2 --warmup 1 --no-cpu-baseline > $O/$1.json 2>$O/$1.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$1.json").read().strip().split("\n")[-1]); r=d["roofline"]
    print("$1", "rays/s", round(d["value"]), "ddf launch_ms %.3f" % r.get("avg_launch_ms"), "frac %.4f" % r["frac"], "psnr %.2f" % d.get("psnr_vs_oracle_db"))
except Exception as e:
    print("$1 FAILED", e, open("$O/$1.err").read()[-300:])
PY
}
run bf16_5M "NEDDF_REV_SKEW=0,5000000,0" "--dtype bf16"
run bf16_100k "NEDDF_REV_SKEW=0,100000,0" "--dtype bf16"
