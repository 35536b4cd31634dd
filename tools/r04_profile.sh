#!/bin/bash
# Round-4 measurement pass (on the GPU box, from the repository root): bench lines, rocprofv3 kernel stats, PMC passes.
ROOT=$PWD
O=$ROOT/gpurun_out/r4prof
mkdir -p $O
export TMPDIR=/tmp
NEDDF_BENCH_PMC=1 python bench.py --steps 8 --warmup 2 > $O/bench_c2_f32.json 2> $O/bench_c2_f32.err
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
for dt in bf16 f16_split; do
  NEDDF_PROBE_DTYPE=$dt rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_$dt/sq -- python $ROOT/tools/pmc_probe.py 1 > $O/pmc_sq_$dt.log 2>&1
  NEDDF_PROBE_DTYPE=$dt rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_$dt/sq2 -- python $ROOT/tools/pmc_probe.py 1 > $O/pmc_sq2_$dt.log 2>&1
done
cd $ROOT
for d in pmc pmc_bf16 pmc_f16_split; do python tools/pmc_summary.py $O/$d > $O/${d}_summary.csv; done
find $O/prof -name "*kernel_stats.csv"
tail -c 600 $O/bench_c2_f32.json
