#!/usr/bin/env python3
"""Eval-minimal NeDDF field throughput by hidden activation (tanhExp / ReLU / LeakyReLU) and operand policy on 2^22 random sample
points: the reverse-mode distance kernel + colour trunk, with per-stage timings from the library.  A/B of library builds with
NEDDF_LIB_PATH (tools/ab_check.sh)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import neddf_amd
from neddf_amd.fixtures import BUNNY_SMOKE_CFG, synth
dev = torch.device("cuda:0")
N = 1 << 22
g = torch.Generator(device=dev).manual_seed(0)
pos = torch.rand(1, N, 3, device=dev, generator=g) * 2 - 1
d = torch.nn.functional.normalize(torch.randn(1, N, 3, device=dev, generator=g), dim=-1)
var = torch.rand(1, N, 3, device=dev, generator=g) * 1e-4
smp = neddf_amd.Sampling(pos, d, var)
ctx = neddf_amd.Context.get(dev)
sd = synth.neddf_state()
for act in sys.argv[1:] or ["tanhExp", "ReLU", "LeakyReLU"]:
    for dtype in ("fp32", "f16_split", "bf16"):
        with torch.no_grad():
            net = neddf_amd.NeDDF(**dict(BUNNY_SMOKE_CFG, activation_type=act))
            net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); net.to(dev); net.set_iter(-1)
            net.weight_dtype = dtype
            net.output_mode = "minimal"
            o = net(smp); torch.cuda.synchronize()
            ctx.set_timing(True); ctx.get_timings()
            for _ in range(3):
                o = net(smp)
            torch.cuda.synchronize()
            st = ctx.get_stage_timings(); ctx.set_timing(False)
            print("%-10s %-10s ddf %7.3f ms/launch  col %6.3f ms/launch   checksum %.6f %.6f" % (
                act, dtype, st["ddf"][0] / st["ddf"][1], st["col"][0] / st["col"][1], float(o["density"].double().mean()), float(o["color"].double().mean())), flush=True)
