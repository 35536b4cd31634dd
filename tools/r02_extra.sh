#!/bin/bash
# Round-2 extras (GPU box, repository root): tail-filling probe, kernel stats of the training step.
set -x
ROOT=$PWD
O=$ROOT/gpurun_out/r2extra
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/two_stream_probe.py > $O/two_stream.json 2> $O/two_stream.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o t -- python $ROOT/bench.py --workload train --steps 5 --warmup 2 > $O/prof_train.log 2>&1
cd $ROOT
cat $O/two_stream.json; tail -3 $O/two_stream.err
