O=gpurun_out/r04/latency; mkdir -p $O
run() { # name env args
  env $2 python bench.py $3 --steps 3 --warmup 1 --no-cpu-baseline > $O/$1.json 2>$O/$1.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$1.json").read().strip().split("\n")[-1]); r=d["roofline"]
    print("$1", "rays/s", round(d["value"]), "ddf launch_ms %.3f" % r.get("avg_launch_ms"), "frac %.4f" % r["frac"], "col ms %.3f" % (r.get("colour_kernel") or {}).get("avg_launch_ms", 0), "psnr", d.get("psnr_vs_oracle_db"))
except Exception as e:
    print("$1 FAILED", e, open("$O/$1.err").read()[-600:])
PY
}
run bf16 "X=1" "--dtype bf16"
run split "X=1" "--dtype f16_split"
run f32 "X=1" ""
run c5 "X=1" "--workload c5"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_c5.py -x -q -m gpu 2>&1 | tail -3
