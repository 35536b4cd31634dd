#!/bin/bash
# Round 5: kernel breakdown of the training step under both operand policies (rocprofv3 --kernel-trace --stats)
ROOT=$PWD
O=$ROOT/gpurun_out/r5train
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
cd /tmp
for dt in ${DTYPES:-f32 f16_split}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$dt -o t -- python $ROOT/bench.py --workload train --dtype $dt --steps 8 --warmup 3 > $O/train_$dt.log 2>&1
  cp $(find $O/prof_$dt -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats_$dt.csv
  rm -rf $O/prof_$dt
  tail -1 $O/train_$dt.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$dt', round(d['value']), 'rays/s', '%.2f ms/step' % d['ms_per_step'])"
  python - $O/train_kernel_stats_$dt.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print("  %-70s calls %5s  %8.3f ms/step  %5.1f %%" % (r["Name"].split("(")[0][-70:], r["Calls"], float(r["TotalDurationNs"]) / 11 / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
print("  all kernels: %.2f ms/step (11 steps incl. warm-up)" % (tot / 11 / 1e6))
PY
done
