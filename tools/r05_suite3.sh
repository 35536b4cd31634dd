#!/bin/bash
# The whole GPU suite three times in one lease, unbuffered pipes (the driver's conditions), then smoke().
O=gpurun_out/r05s
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for i in 1 2 3; do
  timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu_run$i.txt 2>&1; echo "run $i rc=$?"; tail -2 $O/pytest_gpu_run$i.txt
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
