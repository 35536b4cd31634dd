# training step: tests + kernel stats (GPU box)
O=$PWD/gpurun_out/quick; mkdir -p $O; export TMPDIR=/tmp; ROOT=$PWD
timeout 600 python -m pytest tests/test_gpu_train.py -x -q -m gpu > $O/tests_train.log 2>&1; grep -E "passed|failed|error" $O/tests_train.log | tail -3
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_t -o t -- python $ROOT/bench.py --workload train --steps 5 --warmup 2 > $O/prof_t.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/prof_t/t_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step", round(tot/7e6,2))
for r in rows[:8]: print("   %-60s %5s %9.3f %8.4f"%(r["Name"][:60], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6))
PY
