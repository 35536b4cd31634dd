#!/bin/bash
# Round-3 measurement pass (on the GPU box, from the repository root): bench lines, rocprofv3 kernel stats, PMC passes.
set -x
ROOT=$PWD
O=$ROOT/gpurun_out/r3prof
mkdir -p $O
export TMPDIR=/tmp
python bench.py --steps 8 --warmup 2 > $O/bench_c2_f32.json 2> $O/bench_c2_f32.err
python bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c3_f32.json 2>/dev/null
python bench.py --workload c5 --steps 3 --warmup 1 > $O/bench_c5_bf16.json 2>/dev/null
python bench.py --dtype f16_split --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c2_f16_split.json 2>/dev/null
python bench.py --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c2_bf16.json 2>/dev/null
python bench.py --workload train --steps 8 --warmup 3 > $O/bench_train_f32.json 2>/dev/null
python bench.py --workload train --dtype f16_split --steps 8 --warmup 3 > $O/bench_train_f16_split.json 2>/dev/null
NEDDF_TRAIN_UNFUSED=1 python bench.py --workload train --steps 8 --warmup 3 > $O/bench_train_f32_per_layer.json 2>/dev/null
NEDDF_BENCH_FORCE_DIST=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c2_forced_collective.json 2>/dev/null
for w in 128 384; do python bench.py --width $w --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c2_f32_width$w.json 2>/dev/null; done
python tools/act_probe.py > $O/act_probe.txt 2>/dev/null
python tools/parity_report.py > $O/parity_report.json 2>/dev/null
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o t -- python $ROOT/bench.py --workload train --steps 7 --warmup 2 > $O/prof_train.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc/$c -- python $ROOT/tools/pmc_probe.py 1 > $O/pmc_$c.log 2>&1
  NEDDF_PROBE_DTYPE=bf16 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_bf16/$c -- python $ROOT/tools/pmc_probe.py 1 > $O/pmc_bf16_$c.log 2>&1
  NEDDF_PROBE_ACT=ReLU rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_relu/$c -- python $ROOT/tools/pmc_probe.py 1 > $O/pmc_relu_$c.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc/sq -- python $ROOT/tools/pmc_probe.py 1 > $O/pmc_sq.log 2>&1
cd $ROOT
for d in pmc pmc_bf16 pmc_relu; do python tools/pmc_summary.py $O/$d > $O/${d}_summary.csv; done
find $O/prof $O/prof_train -name "*kernel_stats.csv"
tail -c 400 $O/bench_c2_f32.json
