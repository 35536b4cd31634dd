#!/bin/bash
# fused forward / backward phase ablation of the training step (ablation library; numbers are timings only)
O=gpurun_out/r3v; mkdir -p $O
run() { NEDDF_LIB_PATH=$PWD/neddf_amd/csrc/libneddf_hip_ablate.so NEDDF_DW_ABLATE=$1 timeout 120 python bench.py --workload train --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('abl $1', round(d['ms_per_step'],2))"; }
for b in 0 16 32 48 64 128 240 256 512 1024 2048 3840 4096 8176; do run $b; done 2>&1 | tee $O/ablate.txt
