# reverse-mode distance kernel: tile shapes under the 16-bit policies (NEDDF_REV_GEO_BF16 / NEDDF_REV_GEO_SPLIT = MTxWPSxNW)
O=gpurun_out/r04/revgeo; mkdir -p $O
run() { # name env args
  env $2 python bench.py $3 --steps 3 --warmup 1 --no-cpu-baseline > $O/$1.json 2>$O/$1.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$1.json").read().strip().split("\n")[-1]); r=d["roofline"]
    print("$1", "rays/s", round(d["value"]), "launch_ms %.3f" % r.get("avg_launch_ms"), "frac %.4f" % r.get("frac"), "psnr", d.get("psnr_vs_oracle_db"))
except Exception as e:
    print("$1 FAILED", e, open("$O/$1.err").read()[-300:])
PY
}
run bf16_2x2x4 "X=1" "--dtype bf16"
run bf16_2x2x8 "NEDDF_REV_GEO_BF16=2x2x8" "--dtype bf16"
run bf16_2x3x4 "NEDDF_REV_GEO_BF16=2x3x4" "--dtype bf16"
run bf16_4x2x4 "NEDDF_REV_GEO_BF16=4x2x4" "--dtype bf16"
run split_2x2x4 "X=1" "--dtype f16_split"
run split_2x2x8 "NEDDF_REV_GEO_SPLIT=2x2x8" "--dtype f16_split"
