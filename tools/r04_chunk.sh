# points per field-kernel launch (NEDDF_FIELD_CHUNK_LOG2) x rays per render call (NEDDF_RAYS_PER_CALL): launch tails vs buffer size
O=gpurun_out/r04/chunk; mkdir -p $O
run() { # name env args
  env $2 python bench.py $3 --steps 3 --warmup 1 --no-cpu-baseline > $O/$1.json 2>$O/$1.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$1.json").read().strip().split("\n")[-1]); r=d["roofline"]
    print("$1", "rays/s", round(d["value"]), "ms/step %.1f" % d["ms_per_step"], "ddf launch_ms %.3f x %d" % (r.get("avg_launch_ms"), r["launches"]), "stage", {k: round(v,1) for k,v in d["stage_ms_per_step"].items() if v > 1})
except Exception as e:
    print("$1 FAILED", e, open("$O/$1.err").read()[-300:])
PY
}
for dt in f32 f16_split bf16; do
  run ${dt}_21 "X=1" "--dtype $dt"
  run ${dt}_23 "NEDDF_FIELD_CHUNK_LOG2=23" "--dtype $dt"
  run ${dt}_25_r18 "NEDDF_FIELD_CHUNK_LOG2=25 NEDDF_RAYS_PER_CALL=262144" "--dtype $dt"
done
