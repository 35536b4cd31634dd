#!/usr/bin/env python3
"""Small fixed workload for counter collection: one 65 536-ray slab of the C2
configuration (128 samples/ray) = one distance-trunk + one colour-trunk launch of
2^23 points at the default launch size (NEDDF_FIELD_CHUNK_LOG2 = 23).  Used under `rocprofv3 --pmc ... --kernel-trace`."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import neddf_amd  # noqa: E402
import numpy as np  # noqa: E402
import math  # noqa: E402

dev = torch.device("cuda:0")
render, _ = bench.build_render(dev, activation=os.environ.get("NEDDF_PROBE_ACT"))      # NEDDF_PROBE_ACT=ReLU: y' travels as mask bits
render.network_fine.weight_dtype = os.environ.get("NEDDF_PROBE_DTYPE", "fp32")      # "bf16": the configs[4] kernels
fx = 0.5 * 800 / math.tan(0.5 * bench.CAMERA_ANGLE_X)
R, T = bench.view_pose(0)
cam = neddf_amd.Camera(neddf_amd.PinholeCalib(np.array([fx, fx, 400.0, 400.0])), None).to(dev)
cam.R, cam.T = torch.from_numpy(R).to(dev), torch.from_numpy(T).to(dev)
lo = 360 * 800
U = torch.rand(65536, 128, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    out = render.render_image_single_pass(800, 800, cam, 128, U=U, pixel_range=(lo, lo + 65536))
torch.cuda.synchronize()
print("ok", float(out["color"].mean()))
