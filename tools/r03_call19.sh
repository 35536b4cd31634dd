#!/bin/bash
O=gpurun_out/r3ac; mkdir -p $O
run() { NEDDF_LIB_PATH=$PWD/neddf_amd/csrc/libneddf_hip_ablate.so NEDDF_DW_ABLATE=$1 timeout 120 python bench.py --workload train --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('units $2', round(d['ms_per_step'],2))"; }
for u in 0 8 12 16 20 24 32 0 16; do run $((u*65536)) $u; done 2>&1 | tee $O/antiphase.txt
NEDDF_LIB_PATH=$PWD/neddf_amd/csrc/libneddf_hip_ablate.so NEDDF_DW_ABLATE=$((32768+16*65536)) timeout 120 python bench.py --workload train --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "^fwd wg" | tail -16 | grep "wave 0" | tee $O/fwd_times.txt
for v in pmnt; do NEDDF_LIB_PATH=$PWD/tools/bin/libneddf_hip_$v.so timeout 120 python bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v', round(d['ms_per_step'],2), d['final_loss'])"; done | tee -a $O/antiphase.txt
timeout 120 python bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('product', round(d['ms_per_step'],2), d['final_loss'])" | tee -a $O/antiphase.txt
