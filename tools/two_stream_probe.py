#!/usr/bin/env python3
"""Does filling kernel tails pay?  The headline view (800x800, 128 samples/ray) rendered (a) as bench.py does, every
slab on one stream, and (b) with the slabs alternating between two library contexts on two HIP streams, so that the
last workgroups of one slab's kernels share the device with the first of the next slab's.  Prints one JSON line."""
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench                                    # noqa: E402
from neddf_amd._lib import Context              # noqa: E402
from neddf_amd.render import SLOT_FINE          # noqa: E402
import neddf_amd                                # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
render, _ = bench.build_render(dev)
W = H = 800
fx = 0.5 * W / math.tan(0.5 * bench.CAMERA_ANGLE_X)
cam = neddf_amd.Camera(neddf_amd.PinholeCalib(np.array([fx, fx, W / 2.0, H / 2.0])), None).to(dev)
R, T = bench.view_pose(0)
cam.R, cam.T = torch.from_numpy(R).to(dev), torch.from_numpy(T).to(dev)
n = W * H
U = torch.rand(n, 128, device=dev)
idx = torch.arange(n, device=dev)
uv = torch.stack([idx % W, idx // W], 1)
desc = cam.descriptor()
ctxs = [render._ctx(dev), Context(0)]
render.network_fine.upload(ctxs[1], SLOT_FINE)
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
out = dict(color=torch.empty(n, 3, device=dev), depth=torch.empty(n, device=dev), transmittance=torch.empty(n, device=dev))
flag = torch.zeros(1, device=dev, dtype=torch.int32)
params = render._params()
torch.cuda.synchronize()


def view(two, per_call):
    for i, lo in enumerate(range(0, n, per_call)):
        hi = min(n, lo + per_call)
        k = i & 1 if two else 0
        with torch.cuda.stream(streams[k]):
            ctxs[k].render_rays(uv[lo:hi], desc, params, U[lo:hi], None,
                                dict({key: v[lo:hi] for key, v in out.items()}, nan_flag=flag), single_slot=SLOT_FINE)


res = {}
for name, two, per_call in (("one_stream_65536", False, 65536), ("two_streams_65536", True, 65536),
                            ("one_stream_16384", False, 16384), ("two_streams_16384", True, 16384)):
    view(two, per_call)
    torch.cuda.synchronize()
    ref = out["color"].clone() if not res else ref
    t0 = time.perf_counter()
    for _ in range(3):
        view(two, per_call)
    torch.cuda.synchronize()
    res[name] = {"rays_per_s": 3 * n / (time.perf_counter() - t0), "same_pixels": bool(torch.equal(out["color"], ref))}
print(json.dumps(res))
