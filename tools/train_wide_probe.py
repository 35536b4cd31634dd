#!/usr/bin/env python3
"""Time NeDDF.forward + backward (the training-mode field: all five outputs, every parameter gradient) at another hidden width on
synthetic weights.  `python tools/train_wide_probe.py [width=512] [points=265216] [steps=5]`; NEDDF_PROBE_DTYPE=f16_split for the
split-fp16 policy, NEDDF_TRAIN_WIDE_FUSED=1 for the 512-wide fused chains (probe; default: the blocked per-layer route of round 4)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import neddf_amd  # noqa: E402
from neddf_amd import Sampling  # noqa: E402
from neddf_amd.fixtures import synth  # noqa: E402

width = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024 * 259
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
kw = dict(embed_pos_rank=10, embed_dir_rank=4, ddf_layer_count=8, ddf_layer_width=width, col_layer_count=4, col_layer_width=width,
          d_near=0.01, activation_type="tanhExp", skips=[4])
net = neddf_amd.NeDDF(**kw)
sd = synth.neddf_state(10, 4, 8, width, 4, width, (4,), seed=37)
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
net.to(dev)
net.set_iter(2500)
net.weight_dtype = os.environ.get("NEDDF_PROBE_DTYPE", "fp32")
gen = torch.Generator(device="cpu").manual_seed(3)
B, S = n // 259, 259          # [rays, samples, 3] like render_rays hands them over
n = B * S
pos = (torch.rand(B, S, 3, generator=gen) * 2 - 1).to(dev)
d = torch.nn.functional.normalize(torch.randn(B, S, 3, generator=gen), dim=2).to(dev)
var = (torch.rand(B, S, 3, generator=gen) * 1e-4).to(dev)
keys = ("distance", "density", "color", "fields_penalty", "aux_grad")


def step():
    net.zero_grad()
    o = net(Sampling(pos, d, var))
    sum(o[k].mean() for k in keys).backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
g = torch.cat([p.grad.flatten() for p in net.parameters()])
print("width %d, %d points, %s: %.2f ms per forward + backward; |grad| sum %.6e, finite %s" %
      (width, n, net.weight_dtype, ms, float(g.abs().sum()), bool(torch.isfinite(g).all())))
