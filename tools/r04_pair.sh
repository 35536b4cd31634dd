# the PAIRED shape of the reverse-mode distance kernel (NEDDF_REV_GEO_* = 2x1x4: one workgroup of two anti-phased four-wave groups per CU)
O=gpurun_out/r04/pair; mkdir -p $O
run() { # name env args
  env $2 timeout 300 python bench.py $3 --steps 3 --warmup 1 --no-cpu-baseline > $O/$1.json 2>$O/$1.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$1.json").read().strip().split("\n")[-1]); r=d["roofline"]
    print("$1", "rays/s", round(d["value"]), "ddf launch_ms %.3f" % r.get("avg_launch_ms"), "frac %.4f" % r["frac"], "psnr", d.get("psnr_vs_oracle_db"))
except Exception as e:
    print("$1 FAILED", e, open("$O/$1.err").read()[-600:])
PY
}
run bf16_base "X=1" "--dtype bf16"
run bf16_pair "NEDDF_REV_GEO_BF16=2x1x4" "--dtype bf16"
run split_base "X=1" "--dtype f16_split"
run split_pair "NEDDF_REV_GEO_SPLIT=2x1x4" "--dtype f16_split"
NEDDF_REV_GEO_BF16=2x1x4 NEDDF_REV_GEO_SPLIT=2x1x4 timeout 600 python -m pytest tests/test_gpu_c5.py -x -q -m gpu 2>&1 | tail -3
