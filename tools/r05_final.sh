#!/bin/bash
# Round-5 closing pass (on the GPU box, from the repository root): the whole -m gpu suite under the driver's conditions, smoke(),
# the default bench line, the training lines of both operand policies.
ROOT=$PWD
O=$ROOT/gpurun_out/r5final
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
python __graft_entry__.py smoke-only > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python bench.py --workload train --steps 8 --warmup 3 > $O/bench_train_f32.json 2>/dev/null
python bench.py --workload train --dtype f16_split --steps 8 --warmup 3 > $O/bench_train_f16_split.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5final/bench_default.json")); r = d["roofline"]
print(round(d["value"]), d["unit"], "ms/step %.2f" % d["ms_per_step"], "frac %.3f" % r["frac"], "traffic", r.get("traffic"), "col ms", r["colour_kernel"]["avg_launch_ms"], "psnr", d.get("psnr_vs_oracle_db"))
for f in ("bench_train_f32", "bench_train_f16_split"):
    d = json.load(open("gpurun_out/r5final/%s.json" % f)); print(f, round(d["value"]), "rays/s", "%.2f ms/step" % d["ms_per_step"])
PY
