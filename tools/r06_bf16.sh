#!/bin/bash
# bf16 policy check + A/B: the configs[4] tests on the shipped library, then bench lines per variant (tools/r06_ab.sh)
ROOT=$PWD
TAG=$1; shift
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1 NEDDF_BENCH_PMC=0
python -m pytest tests/test_gpu_c5.py -m gpu -x -q > $O/pytest_c5.txt 2>&1
tail -15 $O/pytest_c5.txt
bash tools/r06_ab.sh $TAG bf16 "$@"
