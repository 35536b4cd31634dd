# quick regression pass on the GPU box: the -m gpu suite + one bench line per operand policy / workload
set -x
O=gpurun_out/quick; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/tests_full.log 2>&1; grep -E "passed|failed|error" $O/tests_full.log | tail -3 > $O/tests.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/c2_f32.json 2>$O/c2_f32.err
python bench.py --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline > $O/c2_bf16.json 2>/dev/null
python bench.py --workload train --steps 5 --warmup 2 > $O/train.json 2>/dev/null
python bench.py --workload c5 --steps 2 --warmup 1 > $O/c5.json 2>/dev/null
cat $O/tests.log
for f in c2_f32 c2_bf16 train c5; do python - <<PY
import json
d=json.loads(open("$O/$f.json").read().strip().split("\n")[-1])
print("$f", round(d["value"]), round(d["ms_per_step"],1), d["roofline"].get("avg_launch_ms"), d["roofline"].get("frac"), d["roofline"].get("colour_kernel"), d.get("alt_operand_policy",{}).get("value"))
PY
done
