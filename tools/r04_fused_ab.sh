# fused (one field kernel) vs two-kernel route: parity suite on the fused default, then C2 bench per policy for both
O=gpurun_out/r04/fused; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_c5.py -x -q -m gpu 2>&1 | tail -15
run() { # name env args
  env $2 python bench.py $3 --steps 3 --warmup 1 --no-cpu-baseline > $O/$1.json 2>$O/$1.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$1.json").read().strip().split("\n")[-1]); r=d["roofline"]
    print("$1", "rays/s", round(d["value"]), "ms/step %.1f" % d["ms_per_step"], "ddf launch_ms %.3f" % r.get("avg_launch_ms"), "col ms", (r.get("colour_kernel") or {}).get("avg_launch_ms"), "psnr", d.get("psnr_vs_oracle_db"), d["stage_ms_per_step"])
except Exception as e:
    print("$1 FAILED", e, open("$O/$1.err").read()[-300:])
PY
}
for dt in f32 f16_split bf16; do
  run ${dt}_fused "NEDDF_FUSED=1" "--dtype $dt"
  run ${dt}_two "NEDDF_FUSED=0" "--dtype $dt"
done
