#!/bin/bash
O=$PWD/gpurun_out/r3h; mkdir -p $O
export TMPDIR=/tmp; ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o t -- python $ROOT/bench.py --workload train --steps 7 --warmup 2 > $O/prof_train.log 2>&1
cd $ROOT
f=$(find $O/prof_train -name "*kernel_stats.csv" | head -1); echo $f; head -16 $f | cut -c1-200
