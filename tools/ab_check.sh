# A/B inside one gpurun call (boxes differ by +-2.5 %): kernel stats of the training step under two settings
O=$PWD/gpurun_out/quick; mkdir -p $O; export TMPDIR=/tmp; ROOT=$PWD
cd /tmp
for w in 4 8; do
NEDDF_DW_WAVES=$w timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_w$w -o t -- python $ROOT/bench.py --workload train --steps 5 --warmup 2 > $O/prof_w$w.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/prof_w$w/t_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("dw_waves=$w total kernel ms per step", round(tot/7e6,2))
for r in rows[:6]: print("   %-60s %5s %9.3f %8.4f"%(r["Name"][:60], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6))
PY
done
