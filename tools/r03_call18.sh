#!/bin/bash
O=gpurun_out/r3ab; mkdir -p $O
NEDDF_LIB_PATH=$PWD/neddf_amd/csrc/libneddf_hip_ablate.so NEDDF_DW_ABLATE=32768 timeout 120 python bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "^fwd wg" | tail -32 | grep "wave 0" | tee $O/fwd_times.txt
