#!/bin/bash
# the GPU suite, plain and under the bounds probe (NEDDF_GUARD=1: every workspace between poisoned bands, checked after every test)
ROOT=$PWD
O=$ROOT/gpurun_out/${1:-r6suite}
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1 NEDDF_BENCH_PMC=0
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
tail -4 $O/pytest_gpu.txt
NEDDF_GUARD=1 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_guard.txt 2>&1
tail -4 $O/pytest_gpu_guard.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
