# quick regression pass on the GPU box: the -m gpu suite + one bench line per operand policy / workload
# (A/B of library variants: NEDDF_LIB_PATH=<variant .so> in front of the bench command, both runs in the SAME call -- boxes differ by +-2.5 %)
O=gpurun_out/quick; mkdir -p $O
run() { # name, args
  python bench.py $2 --steps 3 --warmup 1 --no-cpu-baseline > $O/$1.json 2>$O/$1.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$1.json").read().strip().split("\n")[-1])
    r=d["roofline"]
    print("$1", round(d["value"]), round(d["ms_per_step"],1), r.get("avg_launch_ms"), r.get("frac"), (r.get("colour_kernel") or {}).get("avg_launch_ms"), d.get("psnr_vs_oracle_db"))
except Exception as e:
    print("$1 FAILED", e, open("$O/$1.err").read()[-600:])
PY
}
timeout 900 python -m pytest tests -x -q -m gpu > $O/tests_full.log 2>&1; grep -E "passed|failed|error" $O/tests_full.log | tail -3
run c2_f32 ""
run c2_bf16 "--dtype bf16"
run c2_split "--dtype f16_split"
run c5 "--workload c5"
run c3 "--workload c3"
