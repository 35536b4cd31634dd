#!/usr/bin/env python3
"""Wall time of the drop-in path NeRFRender.render_image (800x800, 64+128 hierarchical samples, chunk 1024) in the
reference-compatible "torch_cpu" RNG mode vs the "device" RNG mode."""
import math, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, neddf_amd
dev = torch.device("cuda:0")
render, _ = bench.build_render(dev)
fx = 0.5 * 800 / math.tan(0.5 * bench.CAMERA_ANGLE_X)
R, T = bench.view_pose(0)
cam = neddf_amd.Camera(neddf_amd.PinholeCalib(np.array([fx, fx, 400.0, 400.0])), None).to(dev)
cam.R, cam.T = torch.from_numpy(R).to(dev), torch.from_numpy(T).to(dev)
for mode in ("device", "torch_cpu", "device", "torch_cpu"):
    render.rng = mode
    torch.manual_seed(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    img = render.render_image(800, 800, cam, ["color", "depth"], 1, 1024)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("render_image 800x800 64+128 rng=%-9s %.3f s  %.0f rays/s" % (mode, dt, 640000 / dt))
