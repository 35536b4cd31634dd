#!/bin/bash
# Round 5: the split-fp16 training route on fused, point-major chains (mlp_backward_split_kernel) against the per-layer route
# (NEDDF_TRAIN_SPLIT_FUSED=0): parity tests of the training file, then the training bench under both.
ROOT=$PWD
O=$ROOT/gpurun_out/r5split
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "not run_script and not trainer_run" > $O/pytest_train.txt 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_train.txt
for v in 1 0 1 0; do
  NEDDF_TRAIN_SPLIT_FUSED=$v python bench.py --workload train --dtype f16_split --steps 8 --warmup 3 > $O/bench_train_split_fused$v.json 2>$O/err$v.txt
  python - $O/bench_train_split_fused$v.json $v <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("NEDDF_TRAIN_SPLIT_FUSED=%s: %d rays/s  %.2f ms/step  final loss %.6f" % (sys.argv[2], d["value"], d["ms_per_step"], d["final_loss"]))
except Exception as e:
    print("failed", e); print(open(sys.argv[1].replace("bench_train_split_fused", "err").replace(".json", ".txt")).read()[-1500:])
PY
done
python bench.py --workload train --steps 8 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32: %d rays/s  %.2f ms/step  final loss %.6f' % (d['value'], d['ms_per_step'], d['final_loss']))"
