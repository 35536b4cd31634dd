#!/bin/bash
O=gpurun_out/r3u; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_train.py -x -q -m gpu > $O/tests_train.log 2>&1; grep -E "passed|failed|error" $O/tests_train.log | tail -3; grep -E "^FAILED|^E  " $O/tests_train.log | head
python bench.py --workload train --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('fp32', round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['achieved'],1), d['final_loss'])"
NEDDF_LIB_PATH=$PWD/tools/bin/libneddf_hip_prev.so python bench.py --workload train --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('prev lib', round(d['value']), round(d['ms_per_step'],2))"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_train -o train -- python $GRAFT_REPO_ROOT/bench.py --workload train --steps 8 --warmup 3 > $GRAFT_REPO_ROOT/$O/prof_train.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof_train -name "*kernel_stats.csv" | head -1); head -25 $f | cut -c1-160
