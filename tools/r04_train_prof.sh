#!/bin/bash
# Training step under rocprofv3 (kernel stats): where the non-GEMM time of the 1024-ray step goes.
ROOT=$PWD
O=$ROOT/gpurun_out/r4train
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python $ROOT/bench.py --workload train --steps 10 --warmup 3 > $O/train.log 2>&1
cd $ROOT
tail -c 400 $O/train.log
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r4train/prof/**/t_kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step (13 steps):", tot/13e6)
for r in rows[:45]:
    print("%8.3f ms/step %6d calls  %5.1f%%  %s"%(float(r["TotalDurationNs"])/13e6,int(r["Calls"]),float(r["Percentage"]),r["Name"][:110]))
PY
