#!/bin/bash
O=gpurun_out/r3aa; mkdir -p $O
timeout 100 python - <<'PY' 2>&1 | tee $O/hbm.txt
import torch, time
x=torch.empty(2<<30, dtype=torch.float32, device='cuda'); y=torch.empty_like(x)
def t(f,n=5):
    f(); torch.cuda.synchronize(); a=torch.cuda.Event(True); b=torch.cuda.Event(True); a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
ms=t(lambda: x.zero_()); print('fill  8 GiB', round(ms,3),'ms', round(x.numel()*4/ms/1e9,2),'TB/s')
ms=t(lambda: y.copy_(x)); print('copy  8+8 GiB', round(ms,3),'ms', round(2*x.numel()*4/ms/1e9,2),'TB/s')
ms=t(lambda: x.sum()); print('read  8 GiB', round(ms,3),'ms', round(x.numel()*4/ms/1e9,2),'TB/s')
PY
for v in head pm pmnt pmnt_wait head pm pmnt pmnt_wait; do NEDDF_LIB_PATH=$PWD/tools/bin/libneddf_hip_$v.so timeout 120 python bench.py --workload train --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v', round(d['ms_per_step'],2), d['final_loss'])"; done | tee $O/ab.txt
