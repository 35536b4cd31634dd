#!/bin/bash
O=$PWD/gpurun_out/r3y; mkdir -p $O; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o t -- python $ROOT/bench.py --workload train --steps 8 --warmup 3 --no-cpu-baseline > $O/prof_train.log 2>&1
cd $ROOT
f=$(find $O/prof_train -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -12 "$f" | cut -c1-150; fi
