#!/bin/bash
# Round-6 validation pass (GPU box, repository root): the full GPU suite, then the bench lines of the three operand policies.
ROOT=$PWD
O=$ROOT/gpurun_out/${1:-r6val}
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1 NEDDF_BENCH_PMC=0
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
python bench.py --steps 6 --warmup 2 > $O/bench_c2_f32.json 2> $O/bench_c2_f32.err
python bench.py --dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_c2_bf16.json 2>/dev/null
python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c5_bf16.json 2>/dev/null
python bench.py --dtype f16_split --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c2_f16_split.json 2>/dev/null
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get("O","gpurun_out/r6val")+"/bench_*.json") if False else glob.glob("gpurun_out/*/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get("roofline",{})
        print(f, round(d["value"]), d["unit"], "ms/step %.2f"%d["ms_per_step"], "frac", round(r.get("frac",0),4), "ddf ms", d.get("stage_ms_per_step",{}).get("ddf"), "psnr", d.get("psnr_vs_oracle_db"))
    except Exception as e: print(f, "ERR", e)
PY
