#!/bin/bash
O=$PWD/gpurun_out/r3af; mkdir -p $O; ROOT=$PWD
for v in pmnt new pmnt new; do
  if [ $v = new ]; then L=$PWD/neddf_amd/csrc/libneddf_hip.so; else L=$PWD/tools/bin/libneddf_hip_$v.so; fi
  NEDDF_LIB_PATH=$L timeout 120 python bench.py --workload train --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v', round(d['ms_per_step'],2), d['final_loss'])"; done | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o t -- python $ROOT/bench.py --workload train --steps 8 --warmup 3 --no-cpu-baseline > $O/prof_train.log 2>&1
cd $ROOT
f=$(find $O/prof_train -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -9 "$f" | cut -c1-150; fi
