#!/bin/bash
# kernel breakdown of the wide (512) training step on the fused chains against the blocked route
ROOT=$PWD
O=$ROOT/gpurun_out/r5wideprof
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
cd /tmp
for v in 1 0; do  # (NEDDF_TRAIN_WIDE_FUSED)
  NEDDF_TRAIN_WIDE_FUSED=$v rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o t -- python $ROOT/tools/train_wide_probe.py 512 265216 3 > $O/log_$v.txt 2>&1
  cp $(find $O/prof_$v -name "*kernel_stats.csv" | head -1) $O/kernel_stats_wide_fused$v.csv; rm -rf $O/prof_$v
  tail -1 $O/log_$v.txt
  python - $O/kernel_stats_wide_fused$v.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:9]:
    print("  %-72s calls %5s  %8.2f ms/step  %5.1f %%" % (r["Name"].split("(")[0][-72:], r["Calls"], float(r["TotalDurationNs"]) / 5 / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
print("  all kernels: %.1f ms/step (5 steps incl. warm-up)" % (tot / 5 / 1e6))
PY
done
