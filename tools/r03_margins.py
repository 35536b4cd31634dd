#!/usr/bin/env python3
"""Margins of the synthetic-architecture field gates: max over elements of |a-b| / (rtol |b| + atol) at the north-star tolerance
(1e-4 rel + 1e-5 abs), HIP vs reference golden, for the full (forward-mode) and minimal (reverse-mode) paths; and the density
error relative to the reference's own fp32 error (fp64 yardstick)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import synth, neddf_amd
from neddf_amd import Sampling
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
def margin(a, b, rtol=1e-4, atol=1e-5):
    a = a.astype(np.float64); b = b.astype(np.float64)
    return float((np.abs(a - b) / (rtol * np.abs(b) + atol)).max())
names = sys.argv[1:] or ["neddf_relu", "neddf_tanhexp", "neddf_leaky"]
with torch.no_grad():
    for name in names:
        g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        kw = json.loads(str(g["config"]))
        sd = synth.neddf_state(kw["embed_pos_rank"], kw["embed_dir_rank"], kw["ddf_layer_count"], kw["ddf_layer_width"],
                               kw["col_layer_count"], kw["col_layer_width"], tuple(kw["skips"]), seed=7)
        try:
            net = neddf_amd.NeDDF(**kw)
            net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
            net.to(dev)
            for it, tag in ((-1, "eval"), (2500, "it2500")):
                net.set_iter(it)
                for mode in ("full", "minimal"):
                    net.output_mode = mode
                    o = net(Sampling(T(g["pos"]), T(g["dir"]), T(g["var"])))
                    row = {}
                    for k in o:
                        row[k] = round(margin(o[k].cpu().numpy(), g["%s_%s" % (tag, k)]), 3)
                    ex = g["%s_density_fp64" % tag]
                    e_ref = float(np.abs(g["%s_density" % tag].astype(np.float64) - ex).max())
                    e_hip = float(np.abs(o["density"].cpu().numpy().astype(np.float64) - ex).max())
                    print(name, tag, mode, row, "density err / ref fp32 err = %.2f (%.2e / %.2e)" % (e_hip / max(e_ref, 1e-30), e_hip, e_ref), flush=True)
        except Exception as e:
            print(name, "FAILED:", repr(e)[:300], flush=True)
