#!/bin/bash
O=gpurun_out/r3d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "c2_single_pass or width_and_rank or neus_render" > $O/tests_new.log 2>&1; tail -12 $O/tests_new.log
for fl in 2 1026 2050 4098 8194 16386 31746; do
  NEDDF_LIB_PATH=$PWD/neddf_amd/csrc/libneddf_hip_ablate.so NEDDF_SCHED=$fl timeout 300 python tools/ablate_probe.py 2>&1 | tail -1
done > $O/col_ablation.txt 2>&1; cat $O/col_ablation.txt
for w in 128 384 512; do
  python bench.py --width $w --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_w$w.json 2>$O/bench_w$w.err; python -c "
import json; d=json.loads(open('$O/bench_w$w.json').read().strip().split('\n')[-1]); r=d['roofline']; print('w$w', round(d['value']), r['avg_launch_ms'], round(r['achieved'],1), round(r['frac'],3), r['colour_kernel'], d['psnr_vs_oracle_db'], d['parity_sample']['gate_margin'])" || tail -5 $O/bench_w$w.err
done
python bench.py --steps 3 --warmup 1 > $O/bench_c2.json 2>$O/bench_c2.err; tail -c 1500 $O/bench_c2.json
