#!/bin/bash
for b in 16 32 64 96 128 192; do NEDDF_DW_COST_BASE=$b python bench.py --workload train --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('base=$b', round(d['ms_per_step'],2))"; done
