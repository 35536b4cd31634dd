#!/usr/bin/env python3
"""Measured deviations of the HIP path from the reference goldens and from the CPU oracle (run on the GPU box).

    python tools/parity_report.py > profiles/rNN_parity_report.json
"""
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import neddf_amd  # noqa: E402
from conftest import BUNNY_CFG, golden  # noqa: E402
from oracle import oracle as orc  # noqa: E402

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)       # the fused inference path (with autograd on, NeDDF modules take the training kernels)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    d = np.abs(a - b)
    gate = d - (1e-4 * np.abs(b) + 1e-5)
    return {"max_abs": float(d.max()), "max_rel_where_abs_ref_gt_1e-3": float((d / np.maximum(np.abs(b), 1e-3)).max()),
            "worst_margin_vs_1e-4rel+1e-5abs": float(gate.max()), "ref_range": [float(b.min()), float(b.max())]}


g = golden("bunny_stages.npz")
wts = golden("bunny_weights.npz")
w = {k: wts[k] for k in wts.files}
cfg = dict(BUNNY_CFG, _target_="neddf.network.NeDDF")
render = neddf_amd.NeRFRender(cfg, sample_coarse=64, sample_fine=128, dist_near=2.0, dist_far=6.0, max_dist=6.0,
                              use_coarse_network=False, sampling_type="cone")
render.network_fine.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
render.to(dev); render.set_iter(-1)
cam = neddf_amd.Camera(neddf_amd.PinholeCalib(g["calib"].astype(np.float64)), None).to(dev)
cam.R, cam.T = T(g["R"]), T(g["T"])
ctx = render._ctx(dev)
report = {"what": "HIP (MI355X) vs goldens produced by the reference PyTorch renderer (64 bunny_smoke rays, 65 coarse + 194 fine samples)"}

o = render._render(ctx, T(g["uv"]), cam, T(g["u_coarse"]), T(g["u_fine"]), full=True)
e2e = {}
for k in ("color", "depth", "transmittance", "color_coarse", "depth_coarse", "transmittance_coarse", "weight_coarse", "fields_penalty"):
    e2e[k] = err(o[k].cpu().numpy(), g["out_" + k])
mse = float(np.mean((o["color"].cpu().numpy() - g["out_color"]) ** 2))
e2e["psnr_vs_reference_db"] = 10 * math.log10(1.0 / max(mse, 1e-30))
report["render_rays_end_to_end"] = e2e

net = render.network_fine
net.output_mode = "full"
fld = {}
for tag in ("c", "f"):
    s = neddf_amd.Sampling(T(g[tag + "_pos"]), T(g[tag + "_dir"]), T(g[tag + "_var"]))
    out = net(s)
    fld[tag] = {k: err(out[k].cpu().numpy(), g[tag + "_" + k]) for k in ("distance", "density", "color", "aux_grad", "fields_penalty")}
report["neddf_field_on_reference_samples"] = fld

# stage-level bit-exactness
rd, ro = ctx.raygen(T(g["uv"]), cam.descriptor())
dc = ctx.sample_coarse(T(g["u_coarse"]), 2.0, 6.0)
wc = T(g["weight_coarse_raw"].copy())
df, ids = ctx.importance_resample(T(g["dists_coarse"]), wc, T(g["u_fine"]), True, want_ids=True)
ord_, _ = orc.create_rays(g["uv"], g["R"], g["T"], g["calib"])
_, oids, _ = orc.sample_pdf(g["dists_coarse"], g["weight_coarse_raw"].copy(), g["u_fine"], True)
report["bit_exact"] = {
    "ray_dir_vs_oracle": bool(np.array_equal(rd.cpu().numpy(), ord_)),
    "stratified_dists_vs_reference": bool(np.array_equal(dc.cpu().numpy(), g["dists_coarse"])),
    "importance_samples_vs_reference": bool(np.array_equal(df.cpu().numpy(), g["dists_fine"])),
    "searchsorted_ids_vs_oracle": bool(np.array_equal(ids.cpu().numpy(), oids)),
}
# end-to-end sample positions (coarse weights differ in the last bits, so knots can flip)
dff = torch.empty(64, 194, device=dev)
ctx.render_rays(T(g["uv"]), cam.descriptor(), render._params(), T(g["u_coarse"]), T(g["u_fine"]),
                dict(color=torch.empty(64, 3, device=dev), dists_fine=dff, nan_flag=torch.zeros(1, dtype=torch.int32, device=dev)))
report["bit_exact"]["end_to_end_fine_dists_identical_fraction"] = float(np.mean(dff.cpu().numpy() == g["dists_fine"]))
report["bit_exact"]["end_to_end_fine_dists_max_abs"] = float(np.abs(dff.cpu().numpy() - g["dists_fine"]).max())

# Where do the non-identical end-to-end fine samples come from?  Stage by stage on the REFERENCE's own intermediate tensors:
#   A  compositing the reference's coarse field outputs            -> weights vs the reference's weights
#   B  resampling the reference's weights                          -> identical (above)
#   C  resampling the weights of A                                 -> fine distances
#   D  the whole HIP chain (own field outputs)                     -> fine distances (above)
# so A/C isolate the compositor (its expf and the wave-order sums), D - C is the field's fp32 noise.
comp, _ = ctx.composite(T(g["dists_coarse"]), T(g["c_density"]), T(g["c_color"]), 6.0)
wA = comp["weight"].contiguous()
raw = g["weight_coarse_raw"]
fA = float(np.mean(wA.cpu().numpy() == raw))
wA2 = wA.clone()
dC = ctx.importance_resample(T(g["dists_coarse"]), wA2, T(g["u_fine"]), True)
val = net(neddf_amd.Sampling(T(g["c_pos"]), T(g["c_dir"]), T(g["c_var"])))
compF, _ = ctx.composite(T(g["dists_coarse"]), val["density"].reshape(64, 65).contiguous(), val["color"].reshape(64, 65, 3).contiguous(), 6.0)
report["fine_sample_identity_by_stage"] = {
    "A_composite_weights_identical_fraction": fA,
    "A_composite_weights_max_rel": float(np.max(np.abs(wA.cpu().numpy() - raw) / np.maximum(np.abs(raw), 1e-6))),
    "B_resample_on_reference_weights_identical": bool(np.array_equal(df.cpu().numpy(), g["dists_fine"])),
    "C_resample_on_A_weights_identical_fraction": float(np.mean(dC.cpu().numpy() == g["dists_fine"])),
    "C_max_abs": float(np.abs(dC.cpu().numpy() - g["dists_fine"]).max()),
    "D_end_to_end_identical_fraction": report["bit_exact"]["end_to_end_fine_dists_identical_fraction"],
    "field_density_identical_fraction": float(np.mean(val["density"].cpu().numpy().reshape(64, 65) == g["c_density"])),
    "weights_from_own_field_identical_fraction": float(np.mean(compF["weight"].cpu().numpy() == raw)),
    "reading": "a fine sample is lerp(bins, (u - cdf[i-1]) / (cdf[i] - cdf[i-1])): it moves in its last bits whenever any coarse weight "
               "of the ray does, without any searchsorted index changing",
}

# C1 frame (400x400, bunny pose 0): 4096 random rays, HIP vs the CPU oracle on the same uniforms
rng = np.random.default_rng(7)
idx = rng.choice(400 * 400, 4096, replace=False)
uv = np.stack([idx % 400, idx // 400], 1).astype(np.int64)
uc = rng.uniform(0, 1, (4096, 65)).astype(np.float32); uf = rng.uniform(0, 1, (4096, 129)).astype(np.float32)
o = render._render(ctx, T(uv), cam, T(uc), T(uf), full=False)
torch.cuda.synchronize()
t0 = time.time()
onet = orc.NeDDFOracle(w, **BUNNY_CFG)
ref = orc.render_rays(onet, onet, uv, g["R"], g["T"], g["calib"], uc, uf, 2.0, 6.0, 6.0, "cone")
c1 = {k: err(o[k].cpu().numpy(), ref[k]) for k in ("color", "depth", "transmittance")}
mse = float(np.mean((o["color"].cpu().numpy() - ref["color"]) ** 2))
c1["psnr_vs_oracle_db"] = 10 * math.log10(1.0 / max(mse, 1e-30))
c1["oracle_seconds"] = time.time() - t0
report["c1_400x400_4096_rays_vs_oracle"] = c1
print(json.dumps(report, indent=1))
