# A/B of the tanhExp evaluation modes (tools/act_variants.sh) in ONE call: accuracy on the negative-bias fixture + C2 bench per policy
O=gpurun_out/r04/act; mkdir -p $O
for m in 0 1 2; do
  L=tools/bin/libneddf_hip_act$m.so
  echo "== mode $m"
  NEDDF_LIB_PATH=$L timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "negative_bias_regime or neddf_bunny_field" -s 2>&1 | grep -E "^negbias|passed|failed"
  for dt in f32 f16_split; do
    NEDDF_LIB_PATH=$L python bench.py --dtype $dt --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_${dt}_act$m.json 2>$O/bench_${dt}_act$m.err
    python - <<PY
import json
try:
    d=json.loads(open("$O/bench_${dt}_act$m.json").read().strip().split("\n")[-1]); r=d["roofline"]
    print("mode $m $dt rays/s", round(d["value"]), "launch_ms", r.get("avg_launch_ms"), "frac", r.get("frac"), "colour_ms", (r.get("colour_kernel") or {}).get("avg_launch_ms"), "psnr", d.get("psnr_vs_oracle_db"))
except Exception as e:
    print("mode $m $dt FAILED", e, open("$O/bench_${dt}_act$m.err").read()[-400:])
PY
  done
done
