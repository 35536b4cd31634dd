#!/bin/bash
O=gpurun_out/r3ad; mkdir -p $O
run() { NEDDF_LIB_PATH=$PWD/neddf_amd/csrc/libneddf_hip_ablate.so timeout 120 python bench.py --workload train --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', round(d['ms_per_step'],2), d['final_loss'])"; }
run mt2; NEDDF_FWD_MT1=1 run mt1; run mt2; NEDDF_FWD_MT1=1 run mt1
