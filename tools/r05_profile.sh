#!/bin/bash
# Round-5 measurement pass (on the GPU box, from the repository root): bench lines, rocprofv3 kernel stats, PMC passes, parity report.
ROOT=$PWD
O=$ROOT/gpurun_out/r5prof
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python bench.py --steps 8 --warmup 2 > $O/bench_c2_f32.json 2> $O/bench_c2_f32.err
export NEDDF_BENCH_PMC=0
python bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c3_f32.json 2>/dev/null
python bench.py --workload c5 --steps 3 --warmup 1 > $O/bench_c5_bf16.json 2>/dev/null
python bench.py --dtype f16_split --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c2_f16_split.json 2>/dev/null
NEDDF_BENCH_PMC=1 python bench.py --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c2_bf16.json 2>/dev/null
python bench.py --workload train --steps 8 --warmup 3 > $O/bench_train_f32.json 2>/dev/null
python bench.py --workload train --dtype f16_split --steps 8 --warmup 3 > $O/bench_train_f16_split.json 2>/dev/null
NEDDF_BENCH_FORCE_DIST=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c2_forced_collective.json 2>/dev/null
NEDDF_BENCH_FORCE_DIST=1 python bench.py --scaling strong --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c2_forced_collective_strong.json 2>/dev/null
python tools/parity_report.py > $O/parity_report.json 2>/dev/null
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_bench.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc/$c -- python $ROOT/tools/pmc_probe.py 1 > $O/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc/sq -- python $ROOT/tools/pmc_probe.py 1 > $O/pmc_sq.log 2>&1
cd $ROOT
python tools/pmc_summary.py $O/pmc > $O/pmc_summary.csv
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/pmc $O/prof
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5prof/bench_*.json")):
    try:
        d=json.load(open(f)); r=d.get("roofline",{})
        print(f.split("/")[-1], round(d["value"]), d["unit"], "ms/step %.2f"%d["ms_per_step"], "frac", round(r.get("frac",0),3), "traffic", r.get("traffic"), "psnr", d.get("psnr_vs_oracle_db"))
    except Exception as e: print(f, "ERR", e)
PY
head -5 $O/kernel_stats.csv
