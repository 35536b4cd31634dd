#!/bin/bash
# the whole -m gpu suite + the training bench lines at the tree with the split-fp16 fused training route
ROOT=$PWD
O=$ROOT/gpurun_out/r5suite2
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
python bench.py --workload train --dtype f16_split --steps 8 --warmup 3 > $O/bench_train_f16_split.json 2>/dev/null
python bench.py --workload train --steps 8 --warmup 3 > $O/bench_train_f32.json 2>/dev/null
for f in bench_train_f16_split bench_train_f32; do python -c "import json; d=json.load(open('$O/$f.json')); print('$f', round(d['value']), 'rays/s', '%.2f ms/step' % d['ms_per_step'], 'whole-step TF %.1f' % d['roofline']['achieved'])"; done
