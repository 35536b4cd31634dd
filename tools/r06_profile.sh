#!/bin/bash
# Round-6 measurement pass (on the GPU box, from the repository root): bench lines, rocprofv3 kernel stats, PMC passes, parity report.
ROOT=$PWD
O=$ROOT/gpurun_out/${1:-r6prof}
mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
NEDDF_BENCH_PMC=1 python bench.py --steps 8 --warmup 2 > $O/bench_c2_f32.json 2> $O/bench_c2_f32.err
export NEDDF_BENCH_PMC=0
python bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c3_f32.json 2>/dev/null
python bench.py --workload c5 --steps 3 --warmup 1 > $O/bench_c5_bf16.json 2>/dev/null
python bench.py --dtype f16_split --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c2_f16_split.json 2>/dev/null
NEDDF_BENCH_PMC=1 python bench.py --dtype bf16 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_c2_bf16.json 2>/dev/null
python bench.py --workload train --steps 8 --warmup 3 > $O/bench_train_f32.json 2>/dev/null
python bench.py --workload train --dtype f16_split --steps 8 --warmup 3 > $O/bench_train_f16_split.json 2>/dev/null
NEDDF_BENCH_FORCE_DIST=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c2_forced_collective.json 2>/dev/null
NEDDF_BENCH_FORCE_DIST=1 python bench.py --scaling strong --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c2_forced_collective_strong.json 2>/dev/null
python tools/parity_report.py > $O/parity_report.json 2>/dev/null
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof16 -o b -- python $ROOT/bench.py --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_bench_bf16.log 2>&1
for dt in fp32 bf16; do
  export NEDDF_PROBE_DTYPE=$dt
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$dt/$c -- python $ROOT/tools/pmc_probe.py 1 > $O/pmc_${dt}_$c.log 2>&1
  done
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $O/pmc_$dt/tcc -- python $ROOT/tools/pmc_probe.py 1 > $O/pmc_${dt}_tcc.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_$dt/sq -- python $ROOT/tools/pmc_probe.py 1 > $O/pmc_${dt}_sq.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_$dt/sq2 -- python $ROOT/tools/pmc_probe.py 1 > $O/pmc_${dt}_sq2.log 2>&1
  (cd $ROOT; python tools/pmc_summary.py $O/pmc_$dt > $O/pmc_${dt}_summary.csv)
  rm -rf $O/pmc_$dt
done
unset NEDDF_PROBE_DTYPE
cd $ROOT
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats_f32.csv
cp $(find $O/prof16 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bf16.csv
rm -rf $O/prof $O/prof16
python - "$O" <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get("roofline",{})
        print(f.split("/")[-1], round(d["value"]), d["unit"], "ms/step %.2f"%d["ms_per_step"], "frac", round(r.get("frac",0),4), "traffic", r.get("traffic"), "psnr", d.get("psnr_vs_oracle_db"))
    except Exception as e: print(f, "ERR", e)
PY
head -4 $O/kernel_stats_f32.csv; head -4 $O/kernel_stats_bf16.csv; grep -i "ddf_rev" $O/pmc_bf16_summary.csv
