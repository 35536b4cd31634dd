#!/bin/bash
export NEDDF_LIB_PATH=$PWD/neddf_amd/csrc/libneddf_hip_ablate.so
for fl in 0 1 2 4 8 3 7 15; do NEDDF_DW_ABLATE=$fl python bench.py --workload train --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('dw_ablate=$fl', round(d['ms_per_step'],2))"; done
