#!/bin/bash
O=gpurun_out/r3w; mkdir -p $O
run() { NEDDF_LIB_PATH=$PWD/neddf_amd/csrc/libneddf_hip_ablate.so NEDDF_DW_ABLATE=$1 timeout 120 python bench.py --workload train --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('abl $1', round(d['ms_per_step'],2))"; }
for m in 0 8192 16384 24576 48 768; do run $m; done 2>&1 | tee $O/interleave.txt
