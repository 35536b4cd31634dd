# phase offset between the two workgroups of a CU (NEDDF_REV_SKEW = cycles for f32,bf16,split): C2 bench per policy
O=gpurun_out/r04/skew; mkdir -p $O
run() { # name env args
  env NEDDF_FUSED=0 $2 python bench.py $3 --steps 3 --warmup 1 --no-cpu-baseline > $O/$1.json 2>$O/$1.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$1.json").read().strip().split("\n")[-1]); r=d["roofline"]
    print("$1", "rays/s", round(d["value"]), "ddf launch_ms %.3f" % r.get("avg_launch_ms"), "frac %.4f" % r["frac"], "psnr %.2f" % d.get("psnr_vs_oracle_db"))
except Exception as e:
    print("$1 FAILED", e, open("$O/$1.err").read()[-300:])
PY
}
for k in 0 2000 4000 6000 9000 13000 20000 40000; do
  run bf16_$k "NEDDF_REV_SKEW=0,$k,0" "--dtype bf16"
done
for k in 0 4000 8000 12000 18000 30000; do
  run split_$k "NEDDF_REV_SKEW=0,0,$k" "--dtype f16_split"
done
for k in 0 20000 60000; do
  run f32_$k "NEDDF_REV_SKEW=$k,0,0" "--dtype f32"
done
