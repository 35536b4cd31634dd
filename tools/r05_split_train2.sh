#!/bin/bash
# kernel breakdown of the split-fp16 training step on the fused route + the training tests under the bounds probe
ROOT=$PWD
O=$ROOT/gpurun_out/r5split2
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
DTYPES=f16_split bash tools/r05_train_prof.sh
cp gpurun_out/r5train/train_kernel_stats_f16_split.csv $O/train_kernel_stats_f16_split_fused.csv
NEDDF_GUARD=1 timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "not run_script and not trainer_run" > $O/pytest_train_guard.txt 2>&1; echo "guard pytest rc=$?"; tail -4 $O/pytest_train_guard.txt
