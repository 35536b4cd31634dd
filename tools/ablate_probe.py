#!/usr/bin/env python3
"""Timing ablation of the distance-trunk kernel (NEDDF_SCHED bits, see field_kernels.hip); results are invalid by design.
Needs a library built with the ablation switches: `make -C neddf_amd/csrc clean && make -C neddf_amd/csrc ABLATE=1`
(the shipped build compiles them out); rebuild without ABLATE afterwards."""
import os, sys, math
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, neddf_amd
dev = torch.device("cuda:0")
render, _ = bench.build_render(dev)
render.network_fine.weight_dtype = os.environ.get("NEDDF_PROBE_DTYPE", "fp32")      # "bf16" for the configs[4] kernels
fx = 0.5 * 800 / math.tan(0.5 * bench.CAMERA_ANGLE_X)
R, T = bench.view_pose(0)
cam = neddf_amd.Camera(neddf_amd.PinholeCalib(np.array([fx, fx, 400.0, 400.0])), None).to(dev)
cam.R, cam.T = torch.from_numpy(R).to(dev), torch.from_numpy(T).to(dev)
lo, n = 300 * 800, 65536 * 2
U = torch.rand(n, 128, device=dev)
ctx = render._ctx(dev)
render.render_image_single_pass(800, 800, cam, 128, U=U, pixel_range=(lo, lo + n))
torch.cuda.synchronize()
ctx.set_timing(True); ctx.get_timings()
for _ in range(2):
    render.render_image_single_pass(800, 800, cam, 128, U=U, pixel_range=(lo, lo + n))
torch.cuda.synchronize()
tm = ctx.get_timings()
pts = n * 128 * 2
print("dtype=%s " % render.network_fine.weight_dtype + "MT=%s NEDDF_SCHED=%s ddf %.1f TF-equivalent (%.2f ms/launch) col %.1f TF" % (os.environ.get("NEDDF_TILE_MT", "2"), os.environ.get("NEDDF_SCHED", "2"),
      pts * bench.DDF_FLOP_PER_POINT / tm["ddf_ms"] / 1e9, tm["ddf_ms"] / tm["ddf_launches"], pts * bench.COL_FLOP_PER_POINT / tm["col_ms"] / 1e9))
