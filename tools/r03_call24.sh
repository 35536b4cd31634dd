#!/bin/bash
O=gpurun_out/r3ah; mkdir -p $O
for v in cur fast; do
  if [ $v = cur ]; then LL=$PWD/neddf_amd/csrc/libneddf_hip.so; else LL=$PWD/tools/bin/libneddf_hip_fastact.so; fi
  echo "== $v"; NEDDF_LIB_PATH=$LL timeout 200 python tools/r03_margins.py neddf_tanhexp 2>&1 | grep -v amdgpu.ids
  NEDDF_LIB_PATH=$LL timeout 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, json, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import neddf_amd
from neddf_amd import Sampling
from neddf_amd.fixtures import BUNNY_SMOKE_CFG, bunny_smoke_weights
g = np.load('tests/golden/bunny_stages.npz'); g64 = np.load('tests/golden/bunny_field_fp64.npz')
dev = torch.device('cuda:0'); T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
net = neddf_amd.NeDDF(**{k: v for k, v in BUNNY_SMOKE_CFG.items() if k != '_target_'}); 
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in bunny_smoke_weights().items()}); net.to(dev); net.set_iter(-1)
with torch.no_grad():
    for mode in ('full', 'minimal'):
        net.output_mode = mode
        for tag in ('c', 'f'):
            o = net(Sampling(T(g[tag+'_pos']), T(g[tag+'_dir']), T(g[tag+'_var'])))
            ex = g64[tag+'_density']; e_ref = np.abs(g[tag+'_density'].astype(np.float64)-ex).max(); e_hip = np.abs(o['density'].cpu().numpy().astype(np.float64)-ex).max()
            dd = np.abs(o['distance'].cpu().numpy().astype(np.float64)-g64[tag+'_distance']).max()
            cc = np.abs(o['color'].cpu().numpy().astype(np.float64)-g[tag+'_color']).max()
            print('bunny', mode, tag, 'density err/ref = %.2f (%.2e / %.2e)' % (e_hip/e_ref, e_hip, e_ref), 'distance vs fp64 %.2e' % dd, 'colour vs golden %.2e' % cc)
PY
done | tee $O/margins.txt
