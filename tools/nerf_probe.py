#!/usr/bin/env python3
"""Throughput of the stand-alone field entry points (NeRF and NeDDF minimal/full) on 2^22 random sample points."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import neddf_amd, synth
from conftest import BUNNY_CFG, golden
dev = torch.device("cuda:0")
N = 1 << 22
g = torch.Generator(device=dev).manual_seed(0)
pos = torch.rand(1, N, 3, device=dev, generator=g) * 2 - 1
d = torch.nn.functional.normalize(torch.randn(1, N, 3, device=dev, generator=g), dim=-1)
var = torch.rand(1, N, 3, device=dev, generator=g) * 1e-4
smp = neddf_amd.Sampling(pos, d, var)
ctx = neddf_amd.Context.get(dev)

def run(net, label, flop):
    net(smp); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        net(smp)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("%-36s %8.2f ms  %6.2f Mpoints/s  %6.1f TFLOP/s (algorithmic %.3f MFLOP/point)" % (label, dt * 1e3, N / dt / 1e6, N * flop / dt / 1e12, flop / 1e6))

w = golden("bunny_weights.npz")
for dtype in ("fp32", "f16_split", "bf16"):
    with torch.no_grad():
        nerf = neddf_amd.NeRF()
        nerf.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_state().items()}); nerf.to(dev); nerf.set_iter(-1)
        nerf.weight_dtype = dtype
        run(nerf, "NeRF 8x256 (value only) " + dtype, 2 * 525952)
        net = neddf_amd.NeDDF(**BUNNY_CFG)
        net.load_state_dict({k: torch.from_numpy(w[k]) for k in w.files}); net.to(dev); net.set_iter(-1)
        net.weight_dtype = dtype
        net.output_mode = "minimal"
        run(net, "NeDDF eval-minimal " + dtype, 2 * (4 * (423936 + 256) + 256 + 219648 + 768))
        net.output_mode = "full"
        run(net, "NeDDF full (penalties) " + dtype, 2 * 4 * (423936 + 512 + 219648 + 768))
        neus = neddf_amd.NeuS()
        neus.to(dev)
        neus.weight_dtype = dtype
        run(neus, "NeuS (random init) " + dtype, 0)
