#!/bin/bash
# Round 5: the probe-shape tests, then the whole GPU suite under the bounds probe (NEDDF_GUARD=1: every guard band checked after every test).
O=gpurun_out/r05h
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_c5.py -x -q -m gpu -k "probe_shapes" > $O/pytest_probe_shapes.txt 2>&1; echo "probe shapes rc=$?"; tail -3 $O/pytest_probe_shapes.txt
NEDDF_GUARD=1 timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu_guard.txt 2>&1; echo "guard rc=$?"; tail -3 $O/pytest_gpu_guard.txt
