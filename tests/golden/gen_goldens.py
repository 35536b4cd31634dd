#!/usr/bin/env python3
"""Golden-vector generator.  Runs ONLY in the build container, where the
reference checkout exists at /root/reference; nothing of the reference travels.

It imports the reference PyTorch renderer (with three stub modules for the
absent cv2 / hydra / omegaconf packages), drives its
public functions on seeded inputs and stores inputs + expected outputs as
.npz fixtures next to this file.  The fixtures are data only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_goldens.py
"""
import importlib
import json
import math
import os
import sys
import types

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, REF)
sys.path.insert(0, HERE)
sys.path.append(os.path.join(HERE, "..", ".."))      # neddf_amd.fixtures.synth (behind the reference on the path: `neddf` must resolve to it)
sys.dont_write_bytecode = True

import synth  # noqa: E402


# --------------------------------------------------------------------------
# stub modules (only what neddf.render imports at module scope)
def _install_stubs():
    cv2 = types.ModuleType("cv2")
    cv2.COLORMAP_JET = 2
    cv2.IMREAD_UNCHANGED = -1
    cv2.applyColorMap = lambda img, cm: img        # identity: the goldens store what goes INTO the colour map

    def imread(path, flags=None):
        """cv2.imread(path, IMREAD_UNCHANGED) semantics through PIL: uint8, channels in B,G,R(,A) order."""
        from PIL import Image
        img = np.asarray(Image.open(path))
        if img.ndim == 3 and img.shape[2] == 4:
            return np.ascontiguousarray(img[:, :, [2, 1, 0, 3]])
        return np.ascontiguousarray(img[:, :, ::-1]) if img.ndim == 3 else img

    cv2.imread = imread
    cv2.imwrite = lambda *a, **k: None
    sys.modules["cv2"] = cv2
    oc = types.ModuleType("omegaconf")

    class DictConfig(dict):
        pass

    oc.DictConfig = DictConfig
    sys.modules["omegaconf"] = oc
    hydra = types.ModuleType("hydra")
    utils = types.ModuleType("hydra.utils")

    def instantiate(cfg, **kw):
        cfg = dict(cfg)
        cfg.update(kw)
        target = cfg.pop("_target_")
        cfg.pop("_recursive_", None)
        mod, cls = target.rsplit(".", 1)
        return getattr(importlib.import_module(mod), cls)(**cfg)

    utils.instantiate = instantiate
    hydra.utils = utils
    sys.modules["hydra"] = hydra
    sys.modules["hydra.utils"] = utils


_install_stubs()
from neddf.camera import Camera, PinholeCalib  # noqa: E402
from neddf.network import NeDDF, NeRF, NeuS  # noqa: E402
from neddf.nn_module import PositionalEncoding  # noqa: E402
from neddf.nn_module.with_grad import (  # noqa: E402
    LeakyReLUGradFunction, LinearGradFunction, PositionalEncodingGradLayer,
    ReLUGradFunction, SigmoidGradFunction, SoftplusGradFunction, TanhExpGradFunction)
from neddf.ray import Ray, Sampling  # noqa: E402
from neddf.render import NeRFRender  # noqa: E402
from scipy.spatial.transform import Rotation  # noqa: E402

torch.set_grad_enabled(False)


def npy(t):
    return t.detach().cpu().numpy().copy()


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez(path, **arrs)
    print("wrote %s (%.1f KB)" % (name, os.path.getsize(path) / 1024))


def to_torch_sd(sd):
    return {k: torch.from_numpy(v) for k, v in sd.items()}


def make_camera(width, height, frame, angle_x):
    """Mirrors the pose/intrinsics maths of nerf_synthetic_dataset.py:49-62."""
    focal = 0.5 * width / math.tan(0.5 * angle_x)
    m = np.array(frame["transform_matrix"], dtype=np.float64)
    rotvec = Rotation.from_matrix(m[:3, :3]).as_rotvec()
    t = m[:3, 3]
    calib = np.array([focal, focal, width / 2.0, height / 2.0])
    cam = Camera(PinholeCalib(calib), np.r_[rotvec, t].astype(np.float32))
    return cam, calib.astype(np.float32)


# --------------------------------------------------------------------------
def gen_bunny():
    cfg = yaml.safe_load(open(os.path.join(REF, "pretrained/bunny_smoke/.hydra/config.yaml")))
    rcfg = dict(cfg["render"])
    rcfg.pop("_target_")
    render = NeRFRender(network_config=cfg["network"], **rcfg)
    sd = torch.load(os.path.join(REF, "pretrained/bunny_smoke/models/model_02000.pth"),
                    map_location="cpu")
    print(render.load_state_dict(sd))
    render.set_iter(-1)
    render.network_fine.eval()
    net_sd = {k: npy(v) for k, v in render.network_fine.state_dict().items()}
    np.savez(os.path.join(HERE, "..", "..", "neddf_amd", "fixtures", "bunny_smoke_weights.npz"), **net_sd)     # ships with the product

    tf = json.load(open(os.path.join(REF, "data/bunny_smoke/transforms_test.json")))
    W = H = 400
    cam, calib = make_camera(W, H, tf["frames"][0], tf["camera_angle_x"])
    cam.update_transform()
    rng = np.random.default_rng(2024)
    uv_c = rng.integers(120, 280, (48, 2))
    uv_a = rng.integers(0, 400, (16, 2))
    uv = torch.from_numpy(np.concatenate([uv_c, uv_a]).astype(np.int64))
    B = uv.shape[0]
    Sc, Sf = render.sample_coarse, render.sample_fine

    # --- end-to-end render_rays with captured uniforms (SURVEY App.B step 4)
    torch.manual_seed(0)
    state = torch.get_rng_state()
    u_coarse = torch.rand(B, Sc + 1)
    u_fine = torch.rand(B, Sf + 1)
    torch.set_rng_state(state)
    out = render.render_rays(uv, cam)

    # --- stage by stage on the same inputs (all public functions)
    rays = cam.create_rays(uv)
    dists_c = (torch.linspace(render.dist_near, render.dist_far, Sc + 1).reshape(1, -1).expand(B, -1)
               + u_coarse * ((render.dist_far - render.dist_near) / Sc))
    radius = 1.0 / 1111 / math.sqrt(12)
    samp_c = rays.get_sampling_cones(dists_c, radius)
    val_c = render.network_coarse(samp_c)
    int_c = render.integrate_volume_render(dists_c, val_c["density"], val_c["color"])
    w_c_raw = npy(int_c["weight"])
    torch.set_rng_state(state)
    _ = torch.rand(B, Sc + 1)
    dists_f = render.sample_pdf(dists_c, int_c["weight"], Sf + 1)  # mutates weight in place
    samp_f = rays.get_sampling_cones(dists_f, radius)
    val_f = render.network_fine(samp_f)
    int_f = render.integrate_volume_render(dists_f, val_f["density"], val_f["color"])
    for k in ("color", "depth", "transmittance", "weight"):
        assert torch.equal(int_f[k], out[k]), k
    pen_c = torch.sum((dists_c[:, 1:] - dists_c[:, :-1]) * val_c["fields_penalty"][:, :-1], dim=1)
    pen_f = torch.sum((dists_f[:, 1:] - dists_f[:, :-1]) * val_f["fields_penalty"][:, :-1], dim=1)
    assert torch.equal(pen_f, out["fields_penalty"])
    # point sampling of the same coarse distances (ray.py:88)
    samp_p = rays.get_sampling_points(dists_c)
    arrs = dict(
        uv=npy(uv), R=npy(cam.R), T=npy(cam.T), calib=calib,
        dist_near=np.float32(render.dist_near), dist_far=np.float32(render.dist_far),
        max_dist=np.float32(render.max_dist), ray_radius=np.float64(radius),
        sample_coarse=np.int32(Sc), sample_fine=np.int32(Sf),
        ray_dir=npy(rays.ray_dir), ray_orig=npy(rays.ray_orig),
        u_coarse=npy(u_coarse), u_fine=npy(u_fine),
        dists_coarse=npy(dists_c), dists_fine=npy(dists_f),
        c_pos=npy(samp_c.sample_pos), c_dir=npy(samp_c.sample_dir), c_var=npy(samp_c.diag_variance),
        p_pos=npy(samp_p.sample_pos), p_var=npy(samp_p.diag_variance),
        f_pos=npy(samp_f.sample_pos), f_dir=npy(samp_f.sample_dir), f_var=npy(samp_f.diag_variance),
        weight_coarse_raw=w_c_raw, weight_coarse=npy(int_c["weight"]),
        pen_coarse=npy(pen_c), pen_fine=npy(pen_f),
        num_threads=np.int32(torch.get_num_threads()),
    )
    for k, v in val_c.items():
        arrs["c_" + k] = npy(v)
    for k, v in val_f.items():
        arrs["f_" + k] = npy(v)
    for k in ("color", "depth", "transmittance"):
        arrs["ic_" + k] = npy(int_c[k])
    for k, v in out.items():
        arrs["out_" + k] = npy(v)
    save("bunny_stages.npz", **arrs)

    # --- render_image on a tiny frame: pins uv order, chunking and RNG draw order
    w, h, chunk = 12, 10, 50
    cam2, calib2 = make_camera(w, h, tf["frames"][7], tf["camera_angle_x"])
    cam2.update_transform()
    torch.manual_seed(0)
    img = render.render_image(w, h, cam2, ["color", "depth", "transmittance"], 1, chunk)
    torch.manual_seed(0)
    img2 = render.render_image(2 * w, 2 * h, cam2, ["color", "depth"], 2, 64)
    save("bunny_image_small.npz", R=npy(cam2.R), T=npy(cam2.T), calib=calib2,
         width=np.int32(w), height=np.int32(h), chunk=np.int32(chunk), seed=np.int32(0),
         color=npy(img["color"]), depth=npy(img["depth"]), transmittance=npy(img["transmittance"]),
         ds_color=npy(img2["color"]), ds_depth=npy(img2["depth"]), ds_chunk=np.int32(64))
    return render


# --------------------------------------------------------------------------
def gen_ops():
    """Unit goldens for the (value, Jacobian) ops on the inputs the reference's
    own tests use (tests/nn_module/with_grad/*.py: torch.manual_seed(1))."""
    a = {}
    torch.manual_seed(1)
    x = torch.rand(10, 15) * 8 - 4          # wider than the test's [0,1) to hit both signs
    J = torch.rand(10, 3, 15) * 2 - 1
    x[0, 0] = 25.0                          # threshold branch (x > 20)
    x[0, 1] = -30.0
    a["act_x"], a["act_J"] = npy(x), npy(J)
    for name, fn in (("leaky", LeakyReLUGradFunction), ("relu", ReLUGradFunction),
                     ("softplus", SoftplusGradFunction), ("tanhexp", TanhExpGradFunction)):
        y, G = fn.apply(x.clone(), J.clone())
        a[name + "_y"], a[name + "_G"] = npy(y), npy(G)
    xs, Js = x[:, :1].clone(), J[:, :, :1].clone()
    y, G = SigmoidGradFunction.apply(xs, Js)
    a["sigmoid_y"], a["sigmoid_G"] = npy(y), npy(G)
    # linear (test_linear.py inputs)
    torch.manual_seed(1)
    x = torch.rand(10, 3)
    J = torch.eye(3).unsqueeze(0).expand(10, 3, 3).contiguous()
    w = torch.rand(3, 128)
    b = torch.rand(128)
    y, G = LinearGradFunction.apply(x, J, w, b)
    a.update(lin_x=npy(x), lin_J=npy(J), lin_w=npy(w), lin_b=npy(b), lin_y=npy(y), lin_G=npy(G))
    # PE with grad (test_positional_encoding.py inputs) and a scaled rank-10 case
    torch.manual_seed(1)
    x = torch.rand(10, 3)
    J = torch.rand(10, 3, 3)
    y, G = PositionalEncodingGradLayer(4)(x, J)
    a.update(pe4_x=npy(x), pe4_J=npy(J), pe4_y=npy(y), pe4_G=npy(G))
    layer = PositionalEncodingGradLayer(10)
    x = torch.rand(12, 3) * 4 - 2
    var = torch.rand(12, 1, 3) * 1e-3
    smp = Sampling(x.reshape(12, 1, 3), x.reshape(12, 1, 3), var)
    wts = smp.get_pe_weights(layer.freq)
    eye = torch.eye(3).unsqueeze(0).expand(12, 3, 3)
    y, G = layer(x, eye, layer.get_grad_scale() * wts)
    a.update(pe10_x=npy(x), pe10_var=npy(var.reshape(12, 3)), pe10_w=npy(wts), pe10_y=npy(y), pe10_G=npy(G),
             pe10_gradscale=npy(layer.get_grad_scale()))
    for alpha in (3.25, 9.5, 10.0):
        a["lowpass_%g" % alpha] = npy(layer.get_lowpass_scale(alpha))
    a["pedir_y"] = npy(PositionalEncoding(4)(x))
    save("ops.npz", **a)


# --------------------------------------------------------------------------
def gen_fields():
    """Field goldens on synthetic seeded weights (tests/golden/synth.py)."""
    pos, d, var = synth.random_sampling(6, 40, seed=5, cone=True)
    smp = Sampling(torch.from_numpy(pos), torch.from_numpy(d), torch.from_numpy(var))
    cases = {
        # reference test fixture config (tests/conftest.py:77-99)
        "neddf_relu": dict(embed_pos_rank=6, embed_dir_rank=4, ddf_layer_count=8, ddf_layer_width=256,
                           col_layer_count=8, col_layer_width=256, d_near=0.01, activation_type="ReLU",
                           density_activation_type="ReLU", skips=[4], lowpass_alpha_offset=10,
                           penalty_weight={"constraints_aux_grad": 0.05, "constraints_dDdt": 0.05,
                                           "constraints_color": 0.0, "range_distance": 1.0,
                                           "range_aux_grad": 1.0}),
        # shipped architecture (config/network/neddf.yaml) on synthetic weights
        "neddf_tanhexp": dict(embed_pos_rank=10, embed_dir_rank=4, ddf_layer_count=8, ddf_layer_width=256,
                              col_layer_count=4, col_layer_width=256, d_near=0.001, activation_type="tanhExp",
                              density_activation_type="ReLU", skips=[4], lowpass_alpha_offset=10,
                              penalty_weight={"constraints_aux_grad": 0.05, "constraints_dDdt": 1.0,
                                              "constraints_color": 0.0001, "range_distance": 1.0,
                                              "range_aux_grad": 1.0, "range_color": 0.1}),
        "neddf_leaky": dict(embed_pos_rank=8, embed_dir_rank=3, ddf_layer_count=6, ddf_layer_width=256,
                            col_layer_count=3, col_layer_width=256, d_near=0.01, activation_type="LeakyReLU",
                            density_activation_type="tanhExp", skips=[2], lowpass_alpha_offset=3),
    }
    # other hidden widths and more than one skip connection: the constructors take any (neddf.py:52-66, nerf.py:34-44)
    cases.update({
        "neddf_w128": dict(embed_pos_rank=6, embed_dir_rank=4, ddf_layer_count=6, ddf_layer_width=128,
                           col_layer_count=4, col_layer_width=128, d_near=0.01, activation_type="tanhExp",
                           density_activation_type="ReLU", skips=[2], lowpass_alpha_offset=10),
        "neddf_w384": dict(embed_pos_rank=10, embed_dir_rank=4, ddf_layer_count=8, ddf_layer_width=384,
                           col_layer_count=4, col_layer_width=384, d_near=0.01, activation_type="ReLU",
                           density_activation_type="ReLU", skips=[4], lowpass_alpha_offset=10),
        "neddf_w192": dict(embed_pos_rank=8, embed_dir_rank=4, ddf_layer_count=7, ddf_layer_width=192,
                           col_layer_count=3, col_layer_width=192, d_near=0.01, activation_type="tanhExp",
                           density_activation_type="LeakyReLU", skips=[3], lowpass_alpha_offset=5),
        "neddf_skips2": dict(embed_pos_rank=8, embed_dir_rank=4, ddf_layer_count=8, ddf_layer_width=256,
                             col_layer_count=4, col_layer_width=256, d_near=0.01, activation_type="tanhExp",
                             density_activation_type="ReLU", skips=[1, 4], lowpass_alpha_offset=10),
    })
    for name, kw in cases.items():
        sd = synth.neddf_state(kw["embed_pos_rank"], kw["embed_dir_rank"], kw["ddf_layer_count"],
                               kw["ddf_layer_width"], kw["col_layer_count"], kw["col_layer_width"],
                               tuple(kw["skips"]), seed=7)
        arrs = dict(pos=pos, dir=d, var=var, config=np.array(json.dumps(kw)))
        # fp32 = the golden; fp64 (torch default dtype float64: same code, same weights) = the yardstick for fp32 noise --
        # density = (1 - |grad D, aux|) / D amplifies rounding, so its gate is stated relative to the reference's OWN fp32 error
        for dtype, suffix in ((torch.float32, ""), (torch.float64, "_fp64")):
            torch.set_default_dtype(dtype)
            net = NeDDF(**kw)
            print(name, dtype, net.load_state_dict({k: v.to(dtype) for k, v in to_torch_sd(sd).items()}))
            smp_t = Sampling(*(torch.from_numpy(x).to(dtype) for x in (pos, d, var)))
            for it in (-1, 2500):   # eval, and a warm-up iteration (lowpass + aux_grad_scale live)
                net.set_iter(it)
                out = net(smp_t)
                tag = "eval" if it == -1 else "it%d" % it
                for k, v in out.items():
                    arrs["%s_%s%s" % (tag, k, suffix)] = npy(v)
            torch.set_default_dtype(torch.float32)
        save(name + ".npz", **arrs)

    kw = dict(embed_pos_rank=10, embed_dir_rank=4, layer_count=8, layer_width=256, activation_type="ReLU",
              density_activation_type="ReLU", skips=[4], lowpass_alpha_offset=10)
    for name, kw2 in (("nerf_relu", kw),
                      ("nerf_tanhexp", dict(kw, activation_type="tanhExp", density_activation_type="LeakyReLU",
                                            embed_pos_rank=6, layer_count=6, skips=[2],
                                            lowpass_alpha_offset=2)),
                      ("nerf_w128", dict(kw, layer_width=128)),
                      ("nerf_w384", dict(kw, layer_width=384, activation_type="tanhExp", embed_pos_rank=8, layer_count=6,
                                         skips=[2])),
                      ("nerf_skips2", dict(kw, skips=[1, 4]))):
        net = NeRF(**kw2)
        sd = synth.nerf_state(kw2["embed_pos_rank"], kw2["embed_dir_rank"], kw2["layer_count"],
                              kw2["layer_width"], tuple(kw2["skips"]), seed=11)
        print(name, net.load_state_dict(to_torch_sd(sd)))
        arrs = dict(pos=pos, dir=d, var=var, config=np.array(json.dumps(kw2)))
        for it in (-1, 2500):
            net.set_iter(it)
            out = net(smp)
            tag = "eval" if it == -1 else "it%d" % it
            for k, v in out.items():
                arrs["%s_%s" % (tag, k)] = npy(v)
        save(name + ".npz", **arrs)

    # NeRFRender with a NeRF network pair, point sampling, 2 separate nets (config/render/nerf_render.yaml)
    ncfg = dict(kw, _target_="neddf.network.NeRF")
    render = NeRFRender(network_config=ncfg, sample_coarse=32, sample_fine=48, dist_near=2.0, dist_far=6.0,
                        max_dist=6.0, use_coarse_network=True, sampling_type="point")
    render.network_coarse.load_state_dict(to_torch_sd(synth.nerf_state(seed=21)))
    render.network_fine.load_state_dict(to_torch_sd(synth.nerf_state(seed=22)))
    render.set_iter(-1)
    calib = np.array([100.0, 100.0, 320.0, 240.0])
    cam = Camera(PinholeCalib(calib), np.array([0.02, 0.04, 0.06, 0.1, 0.2, 0.3], dtype=np.float32))
    us = torch.linspace(0, 480, 32)
    vs = torch.linspace(0, 640, 32)
    uvf = torch.stack([us, vs], 1)            # float uv as in tests/render/test_nerf_render.py:51-53
    torch.manual_seed(3)
    state = torch.get_rng_state()
    u_c = torch.rand(32, 33)
    u_f = torch.rand(32, 49)
    torch.set_rng_state(state)
    out = render.render_rays(uvf, cam)
    arrs = dict(uv=npy(uvf), R=npy(cam.R), T=npy(cam.T), calib=calib.astype(np.float32),
                u_coarse=npy(u_c), u_fine=npy(u_f))
    for k, v in out.items():
        arrs["out_" + k] = npy(v)
    save("nerf_render_rays.npz", **arrs)


# --------------------------------------------------------------------------
def gen_render_edges(render):
    """integrate_volume_render / sample_pdf on edge-case inputs."""
    a = {}
    rng = np.random.default_rng(99)
    B, S = 9, 33
    dists = np.sort(rng.uniform(2, 6, (B, S)).astype(np.float32), axis=1)
    dens = rng.uniform(-5, 30, (B, S)).astype(np.float32)
    dens[0] = 0.0
    dens[1] = 1e4                       # saturating opacity
    dens[2] = -3.0                      # negative density -> T > 1
    dists[3, 10:14] = dists[3, 10]      # repeated distances (zero deltas)
    col = rng.uniform(-0.5, 1.2, (B, S, 3)).astype(np.float32)
    r = render.integrate_volume_render(torch.from_numpy(dists), torch.from_numpy(dens), torch.from_numpy(col))
    a.update(iv_dists=dists, iv_dens=dens, iv_col=col, iv_weight=npy(r["weight"]), iv_depth=npy(r["depth"]),
             iv_color=npy(r["color"]), iv_trans=npy(r["transmittance"]), iv_max_dist=np.float32(render.max_dist))
    # test_nerf_render.py:17-30 inputs (constant density/colour)
    d2 = torch.linspace(0.0, 2.0, 64).unsqueeze(0).expand(32, 64).contiguous()
    r = render.integrate_volume_render(d2, torch.ones(32, 64), torch.ones(32, 64, 3))
    a.update(ivc_depth=npy(r["depth"]), ivc_color=npy(r["color"]), ivc_trans=npy(r["transmittance"]),
             ivc_weight=npy(r["weight"]))

    B, Sc, Sf = 8, 16, 21
    dists = np.sort(rng.uniform(2, 6, (B, Sc + 1)).astype(np.float32), axis=1)
    w = rng.uniform(0, 1, (B, Sc)).astype(np.float32) ** 3
    w[0] = 0.0                                  # flat pdf
    w[1, :] = 0.0
    w[1, 5] = 1.0                               # single spike
    w[2, ::2] = -0.3                            # negative weights -> sanitised
    w[3, 4] = np.nan                            # NaN weight -> sanitised
    w[4] = 1e-9
    for cat in (True, False):
        torch.manual_seed(5)
        state = torch.get_rng_state()
        u = torch.rand(B, Sf)
        torch.set_rng_state(state)
        wt = torch.from_numpy(w.copy())
        out = render.sample_pdf(torch.from_numpy(dists), wt, Sf, cat_coarse=cat)
        tag = "cat" if cat else "nocat"
        a.update({"sp_%s_out" % tag: npy(out), "sp_%s_wafter" % tag: npy(wt)})
        a["sp_u"] = npy(u)
    a.update(sp_dists=dists, sp_w=w)
    save("render_edges.npz", **a)


def gen_neus():
    """NeuS field (neus.py:101-162; normals by torch.autograd.grad) on synthetic seeded weights."""
    pos, d, var = synth.random_sampling(6, 40, seed=5, cone=False)
    cases = {
        "neus_relu": dict(embed_pos_rank=6, embed_dir_rank=4, sdf_layer_count=8, sdf_layer_width=256, col_layer_count=8,
                          col_layer_width=256, init_variance=0.3, activation_type="ReLU", skips=[4]),     # tests/conftest.py:61-74
        "neus_tanhexp": dict(embed_pos_rank=10, embed_dir_rank=3, sdf_layer_count=6, sdf_layer_width=256, col_layer_count=3,
                             col_layer_width=256, init_variance=0.7, activation_type="tanhExp", skips=[2]),
        # the two trunks of a NeuS may have different widths (neus.py:80-99)
        "neus_w128_384": dict(embed_pos_rank=6, embed_dir_rank=4, sdf_layer_count=6, sdf_layer_width=128, col_layer_count=3,
                              col_layer_width=384, init_variance=0.5, activation_type="tanhExp", skips=[2]),
        "neus_w320_64": dict(embed_pos_rank=8, embed_dir_rank=4, sdf_layer_count=7, sdf_layer_width=320, col_layer_count=4,
                             col_layer_width=64, init_variance=0.4, activation_type="ReLU", skips=[1, 4]),
    }
    for name, kw in cases.items():
        net = NeuS(**kw)
        sd = synth.neus_state(kw["embed_pos_rank"], kw["embed_dir_rank"], kw["sdf_layer_count"], kw["sdf_layer_width"],
                              kw["col_layer_count"], kw["col_layer_width"], tuple(kw["skips"]), kw["init_variance"], seed=13)
        print(name, net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}))
        with torch.enable_grad():
            out = net(Sampling(torch.from_numpy(pos.copy()), torch.from_numpy(d), torch.from_numpy(var)))
        arrs = dict(pos=pos, dir=d, var=var, config=np.array(json.dumps(kw)))
        for k, v in out.items():
            arrs["eval_" + k] = npy(v)
        save(name + ".npz", **arrs)

    # NeRFRender over one NeuS network (point samples): the whole render_rays pipeline with the sdf-derived density
    kw = cases["neus_relu"]
    render = NeRFRender(network_config=dict(kw, _target_="neddf.network.NeuS"), sample_coarse=32, sample_fine=48, dist_near=2.0,
                        dist_far=6.0, max_dist=6.0, use_coarse_network=False, sampling_type="point")
    sd = synth.neus_state(kw["embed_pos_rank"], kw["embed_dir_rank"], kw["sdf_layer_count"], kw["sdf_layer_width"],
                          kw["col_layer_count"], kw["col_layer_width"], tuple(kw["skips"]), kw["init_variance"], seed=17)
    render.network_fine.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    render.set_iter(-1)
    calib = np.array([100.0, 100.0, 320.0, 240.0])
    cam = Camera(PinholeCalib(calib), np.array([0.02, 0.04, 0.06, 0.1, 0.2, 0.3], dtype=np.float32))
    uv = torch.stack([torch.linspace(40, 600, 24), torch.linspace(30, 450, 24)], 1).to(torch.int64)
    torch.manual_seed(4)
    with torch.enable_grad():
        out = render.render_rays(uv, cam)
    arrs = dict(uv=npy(uv), R=npy(cam.R), T=npy(cam.T), calib=calib.astype(np.float32), config=np.array(json.dumps(kw)))
    for k, v in out.items():
        arrs["out_" + k] = npy(v)
    save("neus_render_rays.npz", **arrs)


def gen_train():
    """One training step of the reference (nerf_trainer.py:81-140) on 12 bunny_smoke rays: render_rays with autograd
    through the hand-written backward passes of neddf/nn_module/with_grad, the three losses of config/loss/neddf_loss.yaml,
    loss.backward().  Stores inputs, losses and the parameter gradients (selected tensors in full, all of them as norms +
    a fixed random projection)."""
    from neddf.loss import ColorLoss, FieldsConstraintLoss, MaskBCELoss
    cfg = yaml.safe_load(open(os.path.join(REF, "pretrained/bunny_smoke/.hydra/config.yaml")))
    rcfg = dict(cfg["render"]); rcfg.pop("_target_")
    # density activation ReLU (config/network/neddf.yaml default) instead of the checkpoint's LeakyReLU: with negative
    # densities the reference's in-place weight sanitisation (base_neural_render.py:52-55) invalidates its own autograd
    # graph ("modified by an inplace operation"), i.e. the reference cannot take a training step in that regime
    ncfg = dict(cfg["network"], density_activation_type="ReLU")
    render = NeRFRender(network_config=ncfg, **rcfg)
    render.load_state_dict(torch.load(os.path.join(REF, "pretrained/bunny_smoke/models/model_02000.pth"), map_location="cpu"))
    render.set_iter(1500)           # a warm-up iteration: aux_grad_scale 0.15, low-pass all-pass (alpha 11.5)
    tf = json.load(open(os.path.join(REF, "data/bunny_smoke/transforms_test.json")))
    cam, calib = make_camera(400, 400, tf["frames"][3], tf["camera_angle_x"])
    cam.update_transform()
    rng = np.random.default_rng(77)
    uv = torch.from_numpy(rng.integers(140, 260, (12, 2)).astype(np.int16))
    target = {"color": torch.from_numpy(rng.uniform(0, 1, (12, 3)).astype(np.float32)),
              "mask": torch.from_numpy((rng.uniform(0, 1, 12) > 0.5).astype(np.float32)),
              "fields_penalty": torch.zeros(12)}
    losses = [ColorLoss(weight=1.0, weight_coarse=0.1), MaskBCELoss(weight=0.05, weight_coarse=0.005),
              FieldsConstraintLoss(weight=0.01, weight_coarse=0.01)]
    torch.manual_seed(9)
    state = torch.get_rng_state()
    u_c, u_f = torch.rand(12, 65), torch.rand(12, 129)
    torch.set_rng_state(state)
    with torch.enable_grad():
        render.zero_grad()
        out = render.render_rays(uv, cam)
        ld = {}
        for f in losses:
            ld.update(f(out, target))
        loss = torch.sum(torch.stack(list(ld.values())))
        loss.backward()
    arrs = dict(uv=npy(uv), R=npy(cam.R), T=npy(cam.T), calib=calib, u_coarse=npy(u_c), u_fine=npy(u_f),
                target_color=npy(target["color"]), target_mask=npy(target["mask"]), loss=npy(loss), iteration=np.int32(1500))
    for k, v in ld.items():
        arrs["loss_" + k] = npy(v)
    for k, v in out.items():
        arrs["out_" + k] = npy(v)
    proj = np.random.default_rng(123)
    full = ("layers_ddf.0.weight", "layers_ddf.5.weight", "layers_ddf.6.bias", "layers_col.0.weight", "layers_col.2.weight",
            "layer_ddf_out.weight", "layer_ddf_out.bias", "layer_aux_out.weight", "layer_aux_out.bias", "layer_col_out.weight",
            "layer_col_out.bias", "layers_ddf.0.bias", "layers_col.0.bias")
    names = []
    for k, p_ in render.network_fine.named_parameters():
        gnp = npy(p_.grad)
        names.append(k)
        arrs["gnorm_" + k] = np.float64(np.linalg.norm(gnp.astype(np.float64)))
        arrs["gproj_" + k] = np.float64(np.sum(gnp.astype(np.float64) * proj.standard_normal(gnp.shape)))
        if k in full:
            arrs["grad_" + k] = gnp
    arrs["param_names"] = np.array(json.dumps(names))

    # stage: integrate_volume_render backward alone
    B, S = 7, 40
    d = torch.sort(torch.from_numpy(rng.uniform(2, 6, (B, S)).astype(np.float32)), dim=1)[0]
    dens = torch.from_numpy(rng.uniform(-2, 15, (B, S)).astype(np.float32)).requires_grad_(True)
    col = torch.from_numpy(rng.uniform(-0.2, 1.1, (B, S, 3)).astype(np.float32)).requires_grad_(True)
    g_col = torch.from_numpy(rng.standard_normal((B, 3)).astype(np.float32))
    g_dep = torch.from_numpy(rng.standard_normal(B).astype(np.float32))
    g_tr = torch.from_numpy(rng.standard_normal(B).astype(np.float32))
    g_w = torch.from_numpy(rng.standard_normal((B, S - 1)).astype(np.float32))
    with torch.enable_grad():
        r = render.integrate_volume_render(d, dens, col)
        obj = (r["color"] * g_col).sum() + (r["depth"] * g_dep).sum() + (r["transmittance"] * g_tr).sum() + (r["weight"] * g_w).sum()
        obj.backward()
    arrs.update(cb_dists=npy(d), cb_dens=npy(dens), cb_col=npy(col), cb_g_color=npy(g_col), cb_g_depth=npy(g_dep),
                cb_g_trans=npy(g_tr), cb_g_weight=npy(g_w), cb_grad_dens=npy(dens.grad), cb_grad_col=npy(col.grad))

    # stage: field backward alone on 40 sample points with random upstream gradients
    pos, dd, var = synth.random_sampling(2, 20, seed=31, cone=True)
    smp = Sampling(torch.from_numpy(pos), torch.from_numpy(dd), torch.from_numpy(var))
    ups = {k: torch.from_numpy(rng.standard_normal((2, 20) + ((3,) if k == "color" else ())).astype(np.float32))
           for k in ("distance", "density", "color", "fields_penalty", "aux_grad")}
    with torch.enable_grad():
        render.zero_grad()
        o = render.network_fine(smp)
        obj = sum((o[k] * ups[k]).sum() for k in ups)
        obj.backward()
    arrs.update(fb_pos=pos, fb_dir=dd, fb_var=var)
    for k, v in ups.items():
        arrs["fb_g_" + k] = npy(v)
    for k, v in o.items():
        arrs["fb_out_" + k] = npy(v)
    proj = np.random.default_rng(456)
    for k, p_ in render.network_fine.named_parameters():
        gnp = npy(p_.grad)
        arrs["fb_gnorm_" + k] = np.float64(np.linalg.norm(gnp.astype(np.float64)))
        arrs["fb_gproj_" + k] = np.float64(np.sum(gnp.astype(np.float64) * proj.standard_normal(gnp.shape)))
        if k in full:
            arrs["fb_grad_" + k] = gnp
    # further architectures at field level (synthetic weights): ReLU / LeakyReLU hidden activations (zero second derivative),
    # LeakyReLU / tanhExp density, other encoding ranks (padding of the GEMM operands), two skip connections
    for tag, kw in (("relu", dict(embed_pos_rank=6, embed_dir_rank=3, ddf_layer_count=6, col_layer_count=3, skips=[1, 3],
                                  activation_type="ReLU", density_activation_type="LeakyReLU")),
                    ("leaky", dict(embed_pos_rank=10, embed_dir_rank=4, ddf_layer_count=8, col_layer_count=4, skips=[4],
                                   activation_type="LeakyReLU", density_activation_type="tanhExp"))):
        net = NeDDF(ddf_layer_width=256, col_layer_width=256, d_near=0.01, lowpass_alpha_offset=10,
                    penalty_weight={"constraints_aux_grad": 0.05, "constraints_dDdt": 0.5, "range_color": 0.1}, **kw)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.neddf_state(
            embed_pos_rank=kw["embed_pos_rank"], embed_dir_rank=kw["embed_dir_rank"], ddf_layer_count=kw["ddf_layer_count"],
            col_layer_count=kw["col_layer_count"], skips=tuple(kw["skips"]), seed=23).items()})
        net.set_iter(2500)
        pos, dd, var = synth.random_sampling(3, 11, seed=37, cone=True)
        ups = {k: torch.from_numpy(rng.standard_normal((3, 11) + ((3,) if k == "color" else ())).astype(np.float32))
               for k in ("distance", "density", "color", "fields_penalty", "aux_grad")}
        with torch.enable_grad():
            net.zero_grad()
            o = net(Sampling(torch.from_numpy(pos), torch.from_numpy(dd), torch.from_numpy(var)))
            sum((o[k] * ups[k]).sum() for k in ups).backward()
        pre = "fx_%s_" % tag
        arrs.update({pre + "pos": pos, pre + "dir": dd, pre + "var": var})
        for k in ups:
            arrs[pre + "g_" + k] = npy(ups[k])
            arrs[pre + "out_" + k] = npy(o[k])
        _grad_records(arrs, pre, net, 789, ("layers_ddf.0.weight", "layers_ddf.0.bias", "layer_aux_out.weight", "layer_col_out.bias"))
    save("train_step.npz", **arrs)


def _grad_records(arrs, prefix, module, proj_seed, full=()):
    proj = np.random.default_rng(proj_seed)
    names = []
    for k, p_ in module.named_parameters():
        gnp = npy(p_.grad)
        names.append(k)
        arrs[prefix + "gnorm_" + k] = np.float64(np.linalg.norm(gnp.astype(np.float64)))
        arrs[prefix + "gproj_" + k] = np.float64(np.sum(gnp.astype(np.float64) * proj.standard_normal(gnp.shape)))
        if k in full:
            arrs[prefix + "grad_" + k] = gnp
    arrs[prefix + "param_names"] = np.array(json.dumps(names))


def gen_train_nerf():
    """NeRF fields under the reference's plain torch autograd (nerf.py:107-165): (a) two configurations on 40 points with
    random upstream gradients, (b) one training step of a two-network NeRFRender (config/render/nerf_render.yaml:
    use_coarse_network, point sampling) with ColorLoss + MaskBCELoss (config/loss/nerf_loss.yaml) on 10 rays."""
    from neddf.loss import ColorLoss, MaskBCELoss
    arrs = {}
    full = ("layers.0.weight", "layers.7.bias", "outL_density.weight", "outL_density.bias", "outL_color.0.bias",
            "outL_color.2.weight", "outL_color.2.bias")       # the large matrices are pinned by norm + random projection
    rng = np.random.default_rng(2024)
    for tag, act, dact, it, shape in (("relu", "ReLU", "ReLU", -1, (2, 20)), ("tanhexp", "tanhExp", "LeakyReLU", 1500, (2, 20)),
                                      ("relu250", "ReLU", "ReLU", 500, (10, 25))):      # 250 rows: ragged GEMM tiles
        net = NeRF(activation_type=act, density_activation_type=dact)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_state(seed=11).items()})
        net.set_iter(it)
        pos, dd, var = synth.random_sampling(shape[0], shape[1], seed=41, cone=True)
        ups = {"density": torch.from_numpy(rng.standard_normal(shape).astype(np.float32)),
               "color": torch.from_numpy(rng.standard_normal(shape + (3,)).astype(np.float32))}
        with torch.enable_grad():
            net.zero_grad()
            o = net(Sampling(torch.from_numpy(pos), torch.from_numpy(dd), torch.from_numpy(var)))
            sum((o[k] * ups[k]).sum() for k in ups).backward()
        pre = "fb_%s_" % tag
        arrs.update({pre + "pos": pos, pre + "dir": dd, pre + "var": var, pre + "iteration": np.int32(it)})
        for k in ups:
            arrs[pre + "g_" + k] = npy(ups[k])
            arrs[pre + "out_" + k] = npy(o[k])
        _grad_records(arrs, pre, net, 456, full)

    # embed_pos_rank 4 (top frequency 8): the sample positions of this build and of the reference differ in the last bit
    # (ray direction = R . normalised pixel direction, summed in a different order), and with rank 10 the 512x frequency
    # plus the ReLU kinks of random weights turn that into 1e-2 differences of the early-layer gradients -- a property of
    # the inputs, not of either implementation (the rank-10 field is pinned on identical positions by the fb_* records)
    ncfg = {"_target_": "neddf.network.NeRF", "embed_pos_rank": 4, "embed_dir_rank": 4, "layer_count": 8, "layer_width": 256,
            "activation_type": "ReLU", "density_activation_type": "ReLU", "lowpass_alpha_offset": 10, "skips": [4]}
    render = NeRFRender(network_config=ncfg, sample_coarse=24, sample_fine=40, dist_near=2.0, dist_far=6.0, max_dist=6.0,
                        use_coarse_network=True, sampling_type="point")
    render.network_coarse.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_state(embed_pos_rank=4, seed=11).items()})
    render.network_fine.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_state(embed_pos_rank=4, seed=12).items()})
    render.set_iter(500)
    tf = json.load(open(os.path.join(REF, "data/bunny_smoke/transforms_test.json")))
    cam, calib = make_camera(400, 400, tf["frames"][5], tf["camera_angle_x"])
    cam.update_transform()
    # MaskBCELoss is -log(clamp(1 - T, 1e-6, 1 - 1e-6)): for a ray whose 1 - T sits at the clamp the gradient jumps between 0
    # and ~1e6, so fp32-level differences in T would decide the whole step.  Keep rays that are clearly inside.
    cand = torch.from_numpy(rng.integers(120, 280, (200, 2)).astype(np.int16))
    torch.manual_seed(5)
    with torch.no_grad():
        pre_out = render.render_rays(cand, cam)
    ok = ((1 - pre_out["transmittance"]) > 1e-2) & ((1 - pre_out["transmittance_coarse"]) > 1e-2) & \
         (pre_out["transmittance"] > 1e-2) & (pre_out["transmittance_coarse"] > 1e-2)
    assert int(ok.sum()) >= 10, int(ok.sum())
    uv = cand[ok][:10].contiguous()
    target = {"color": torch.from_numpy(rng.uniform(0, 1, (10, 3)).astype(np.float32)),
              "mask": torch.from_numpy((rng.uniform(0, 1, 10) > 0.5).astype(np.float32))}
    losses = [ColorLoss(weight=1.0, weight_coarse=0.1), MaskBCELoss(weight=0.05, weight_coarse=0.005)]
    torch.manual_seed(21)
    with torch.enable_grad():
        render.zero_grad()
        out = render.render_rays(uv, cam)
        ld = {}
        for f in losses:
            ld.update(f(out, target))
        loss = torch.sum(torch.stack(list(ld.values())))
        loss.backward()
    arrs.update(uv=npy(uv), R=npy(cam.R), T=npy(cam.T), calib=calib, target_color=npy(target["color"]), target_mask=npy(target["mask"]),
                loss=npy(loss), iteration=np.int32(500), seed=np.int32(21))
    for k, v in ld.items():
        arrs["loss_" + k] = npy(v)
    for k, v in out.items():
        arrs["out_" + k] = npy(v)
    _grad_records(arrs, "coarse_", render.network_coarse, 123, full)
    _grad_records(arrs, "fine_", render.network_fine, 124, full)
    save("train_nerf.npz", **arrs)


def gen_train_neus():
    """NeuS fields under the reference's autograd (neus.py:101-162: the normal comes from torch.autograd.grad with
    create_graph=True, so the parameter gradients below are a double backward) on 40 and 250 points with random upstream
    gradients on sdf, density and colour."""
    arrs = {}
    rng = np.random.default_rng(2025)
    cases = {
        "relu": (dict(embed_pos_rank=6, embed_dir_rank=4, sdf_layer_count=8, sdf_layer_width=256, col_layer_count=8,
                      col_layer_width=256, init_variance=0.3, activation_type="ReLU", skips=[4]), (2, 20)),
        "tanhexp": (dict(embed_pos_rank=10, embed_dir_rank=3, sdf_layer_count=6, sdf_layer_width=256, col_layer_count=3,
                         col_layer_width=256, init_variance=0.7, activation_type="tanhExp", skips=[2]), (2, 20)),
        "relu250": (dict(embed_pos_rank=6, embed_dir_rank=4, sdf_layer_count=8, sdf_layer_width=256, col_layer_count=8,
                         col_layer_width=256, init_variance=0.3, activation_type="ReLU", skips=[4]), (10, 25)),
    }
    for tag, (kw, shape) in cases.items():
        net = NeuS(**kw)
        sd = synth.neus_state(kw["embed_pos_rank"], kw["embed_dir_rank"], kw["sdf_layer_count"], kw["sdf_layer_width"],
                              kw["col_layer_count"], kw["col_layer_width"], tuple(kw["skips"]), kw["init_variance"], seed=13)
        net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        pos, dd, var = synth.random_sampling(shape[0], shape[1], seed=43, cone=False)
        ups = {"sdf": torch.from_numpy(rng.standard_normal(shape).astype(np.float32)),
               "density": torch.from_numpy(rng.standard_normal(shape).astype(np.float32)),
               "color": torch.from_numpy(rng.standard_normal(shape + (3,)).astype(np.float32))}
        with torch.enable_grad():
            net.zero_grad()
            o = net(Sampling(torch.from_numpy(pos.copy()), torch.from_numpy(dd), torch.from_numpy(var)))
            sum((o[k] * ups[k]).sum() for k in ups).backward()
        pre = "fb_%s_" % tag
        arrs.update({pre + "pos": pos, pre + "dir": dd, pre + "var": var, pre + "config": np.array(json.dumps(kw))})
        for k in ups:
            arrs[pre + "g_" + k] = npy(ups[k])
            arrs[pre + "out_" + k] = npy(o[k])
        n_sdf, n_col = kw["sdf_layer_count"], kw["col_layer_count"]
        full = ("variance", "layers_sdf.0.weight", "layers_sdf.%d.bias" % (n_sdf - 1), "layers_col.0.bias",
                "layers_col.%d.weight" % n_col, "layers_col.%d.bias" % n_col)
        _grad_records(arrs, pre, net, 457, full)
        for k, p_ in net.named_parameters():
            print(tag, k, float(p_.grad.abs().max()))

    # one training step of NeRFRender over a NeuS network (point samples, ColorLoss + MaskBCELoss) on 10 rays
    from neddf.loss import ColorLoss, MaskBCELoss
    kw = cases["relu"][0]
    render = NeRFRender(network_config=dict(kw, _target_="neddf.network.NeuS"), sample_coarse=24, sample_fine=40, dist_near=2.0,
                        dist_far=6.0, max_dist=6.0, use_coarse_network=False, sampling_type="point")
    sd = synth.neus_state(kw["embed_pos_rank"], kw["embed_dir_rank"], kw["sdf_layer_count"], kw["sdf_layer_width"],
                          kw["col_layer_count"], kw["col_layer_width"], tuple(kw["skips"]), kw["init_variance"], seed=17)
    render.network_fine.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    render.set_iter(500)
    tf = json.load(open(os.path.join(REF, "data/bunny_smoke/transforms_test.json")))
    cam, calib = make_camera(400, 400, tf["frames"][5], tf["camera_angle_x"])
    cam.update_transform()
    cand = torch.from_numpy(rng.integers(120, 280, (200, 2)).astype(np.int16))
    torch.manual_seed(5)
    with torch.enable_grad():
        pre_out = render.render_rays(cand, cam)
    tr_, trc = pre_out["transmittance"].detach(), pre_out["transmittance_coarse"].detach()
    ok = (tr_ > 1e-2) & (tr_ < 1 - 1e-2) & (trc > 1e-2) & (trc < 1 - 1e-2)     # away from MaskBCELoss's clamp (see gen_train_nerf)
    assert int(ok.sum()) >= 10, int(ok.sum())
    uv = cand[ok][:10].contiguous()
    target = {"color": torch.from_numpy(rng.uniform(0, 1, (10, 3)).astype(np.float32)),
              "mask": torch.from_numpy((rng.uniform(0, 1, 10) > 0.5).astype(np.float32))}
    losses = [ColorLoss(weight=1.0, weight_coarse=0.1), MaskBCELoss(weight=0.05, weight_coarse=0.005)]
    torch.manual_seed(21)
    with torch.enable_grad():
        render.zero_grad()
        out = render.render_rays(uv, cam)
        ld = {}
        for f in losses:
            ld.update(f(out, target))
        loss = torch.sum(torch.stack(list(ld.values())))
        loss.backward()
    arrs.update(step_uv=npy(uv), step_R=npy(cam.R), step_T=npy(cam.T), step_calib=calib, step_target_color=npy(target["color"]),
                step_target_mask=npy(target["mask"]), step_loss=npy(loss), step_config=np.array(json.dumps(kw)))
    for k, v in ld.items():
        arrs["step_loss_" + k] = npy(v)
    for k, v in out.items():
        arrs["step_out_" + k] = npy(v)
    n_sdf, n_col = kw["sdf_layer_count"], kw["col_layer_count"]
    _grad_records(arrs, "step_", render.network_fine, 458, ("variance", "layers_sdf.0.weight", "layers_sdf.%d.bias" % (n_sdf - 1),
                                                          "layers_col.0.bias", "layers_col.%d.weight" % n_col, "layers_col.%d.bias" % n_col))
    save("train_neus.npz", **arrs)


def gen_train_widths():
    """Field-level backward goldens at hidden widths other than 256 (the constructors take any: neddf.py:52-66, nerf.py:34-44,
    neus.py:30-41): NeDDF 128 (tanhExp) and 192 (ReLU) under the reference's hand-written (value, Jacobian) backward passes, NeRF
    128 under plain autograd, NeuS with a 128-wide sdf trunk and a 64-wide colour trunk under its double backward; random upstream
    gradients on every output, 33 sample points each."""
    arrs = {}
    rng = np.random.default_rng(2026)
    shape = (3, 11)
    pos, dd, var = synth.random_sampling(shape[0], shape[1], seed=47, cone=True)
    arrs.update(pos=pos, dir=dd, var=var)
    smp = lambda: Sampling(torch.from_numpy(pos.copy()), torch.from_numpy(dd), torch.from_numpy(var))
    for tag, kw in (("neddf128", dict(embed_pos_rank=6, embed_dir_rank=4, ddf_layer_count=6, ddf_layer_width=128, col_layer_count=4,
                                      col_layer_width=128, d_near=0.01, activation_type="tanhExp", density_activation_type="ReLU", skips=[2],
                                      lowpass_alpha_offset=10, penalty_weight={"constraints_aux_grad": 0.05, "constraints_dDdt": 0.5, "range_color": 0.1})),
                    ("neddf192", dict(embed_pos_rank=8, embed_dir_rank=3, ddf_layer_count=7, ddf_layer_width=192, col_layer_count=3,
                                      col_layer_width=192, d_near=0.01, activation_type="ReLU", density_activation_type="LeakyReLU", skips=[1, 4],
                                      lowpass_alpha_offset=10))):
        net = NeDDF(**kw)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.neddf_state(
            kw["embed_pos_rank"], kw["embed_dir_rank"], kw["ddf_layer_count"], kw["ddf_layer_width"], kw["col_layer_count"],
            kw["col_layer_width"], tuple(kw["skips"]), seed=29).items()})
        net.set_iter(2500)
        ups = {k: torch.from_numpy(rng.standard_normal(shape + ((3,) if k == "color" else ())).astype(np.float32))
               for k in ("distance", "density", "color", "fields_penalty", "aux_grad")}
        with torch.enable_grad():
            net.zero_grad()
            o = net(smp())
            sum((o[k] * ups[k]).sum() for k in ups).backward()
        pre = tag + "_"
        arrs[pre + "config"] = np.array(json.dumps(kw))
        for k in ups:
            arrs[pre + "g_" + k] = npy(ups[k])
            arrs[pre + "out_" + k] = npy(o[k])
        _grad_records(arrs, pre, net, 790, ("layers_ddf.0.weight", "layers_ddf.1.weight", "layers_ddf.0.bias", "layers_col.0.weight",
                                           "layer_aux_out.weight", "layer_ddf_out.weight", "layer_col_out.weight", "layer_col_out.bias"))
    kn = dict(embed_pos_rank=6, embed_dir_rank=4, layer_count=6, layer_width=128, activation_type="tanhExp", density_activation_type="ReLU",
              skips=[2], lowpass_alpha_offset=10)
    net = NeRF(**kn)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_state(6, 4, 6, 128, (2,), seed=31).items()})
    net.set_iter(1500)
    ups = {"density": torch.from_numpy(rng.standard_normal(shape).astype(np.float32)),
           "color": torch.from_numpy(rng.standard_normal(shape + (3,)).astype(np.float32))}
    with torch.enable_grad():
        net.zero_grad()
        o = net(smp())
        sum((o[k] * ups[k]).sum() for k in ups).backward()
    arrs["nerf128_config"] = np.array(json.dumps(kn))
    for k in ups:
        arrs["nerf128_g_" + k] = npy(ups[k])
        arrs["nerf128_out_" + k] = npy(o[k])
    _grad_records(arrs, "nerf128_", net, 791, ("layers.0.weight", "layers.3.weight", "layers.5.bias", "outL_density.weight", "outL_color.0.weight",
                                              "outL_color.0.bias", "outL_color.2.weight", "outL_color.2.bias"))
    ks = dict(embed_pos_rank=6, embed_dir_rank=4, sdf_layer_count=5, sdf_layer_width=128, col_layer_count=3, col_layer_width=64,
              init_variance=0.4, activation_type="tanhExp", skips=[2])
    net = NeuS(**ks)
    sd = synth.neus_state(6, 4, 5, 128, 3, 64, (2,), 0.4, seed=33)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    ups = {"sdf": torch.from_numpy(rng.standard_normal(shape).astype(np.float32)),
           "density": torch.from_numpy(rng.standard_normal(shape).astype(np.float32)),
           "color": torch.from_numpy(rng.standard_normal(shape + (3,)).astype(np.float32))}
    with torch.enable_grad():
        net.zero_grad()
        o = net(Sampling(torch.from_numpy(pos.copy()), torch.from_numpy(dd), torch.from_numpy(np.zeros_like(var))))
        sum((o[k] * ups[k]).sum() for k in ups).backward()
    arrs["neus128_config"] = np.array(json.dumps(ks))
    for k in ups:
        arrs["neus128_g_" + k] = npy(ups[k])
        arrs["neus128_out_" + k] = npy(o[k])
    _grad_records(arrs, "neus128_", net, 792, ("variance", "layers_sdf.0.weight", "layers_sdf.3.weight", "layers_sdf.4.bias", "layers_col.0.weight",
                                              "layers_col.0.bias", "layers_col.3.weight", "layers_col.3.bias"))
    save("train_widths.npz", **arrs)


def gen_train_wide():
    """Field-level backward goldens ABOVE hidden width 256 (the reference trains whatever it constructs, neddf.py:52-66): NeDDF 384
    (tanhExp, one skip: trains zero-padded to 512) and NeDDF 512 (ReLU, two skips, `embed_dir_rank` 6 -- above the 4 the training
    kernels took until round 4) under the reference's hand-written (value, Jacobian) backward passes; random upstream gradients on
    every output, 33 sample points."""
    arrs = {}
    rng = np.random.default_rng(2027)
    shape = (3, 11)
    pos, dd, var = synth.random_sampling(shape[0], shape[1], seed=53, cone=True)
    arrs.update(pos=pos, dir=dd, var=var)
    for tag, kw in (("neddf384", dict(embed_pos_rank=6, embed_dir_rank=4, ddf_layer_count=5, ddf_layer_width=384, col_layer_count=3,
                                      col_layer_width=384, d_near=0.01, activation_type="tanhExp", density_activation_type="ReLU", skips=[1],
                                      lowpass_alpha_offset=10, penalty_weight={"constraints_aux_grad": 0.05, "constraints_dDdt": 0.5, "range_color": 0.1})),
                    ("neddf512", dict(embed_pos_rank=8, embed_dir_rank=6, ddf_layer_count=6, ddf_layer_width=512, col_layer_count=4,
                                      col_layer_width=512, d_near=0.01, activation_type="ReLU", density_activation_type="LeakyReLU", skips=[1, 3],
                                      lowpass_alpha_offset=10))):
        net = NeDDF(**kw)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.neddf_state(
            kw["embed_pos_rank"], kw["embed_dir_rank"], kw["ddf_layer_count"], kw["ddf_layer_width"], kw["col_layer_count"],
            kw["col_layer_width"], tuple(kw["skips"]), seed=37).items()})
        net.set_iter(2500)
        ups = {k: torch.from_numpy(rng.standard_normal(shape + ((3,) if k == "color" else ())).astype(np.float32))
               for k in ("distance", "density", "color", "fields_penalty", "aux_grad")}
        with torch.enable_grad():
            net.zero_grad()
            o = net(Sampling(torch.from_numpy(pos.copy()), torch.from_numpy(dd), torch.from_numpy(var)))
            sum((o[k] * ups[k]).sum() for k in ups).backward()
        pre = tag + "_"
        arrs[pre + "config"] = np.array(json.dumps(kw))
        for k in ups:
            arrs[pre + "g_" + k] = npy(ups[k])
            arrs[pre + "out_" + k] = npy(o[k])
        # (full gradients only of the small tensors -- the wide ones are pinned by their norm and a random projection)
        _grad_records(arrs, pre, net, 793, ("layers_ddf.0.weight", "layers_ddf.0.bias", "layers_ddf.2.bias", "layers_col.1.bias",
                                           "layer_aux_out.weight", "layer_ddf_out.weight", "layer_col_out.weight", "layer_col_out.bias"))
    save("train_wide.npz", **arrs)


def gen_train_wide_nerf():
    """The NeRF companion of gen_train_wide (nerf.py:34-44 builds any width; the colour head's hidden width is layer_width // 2):
    NeRF 384 (tanhExp, one skip: trains zero-padded to 512, colour head 192 -> 256) and NeRF 512 (ReLU, two skips,
    `embed_dir_rank` 6) under plain torch autograd (nerf.py:107-165); random upstream gradients on both outputs, 33 sample points."""
    arrs = {}
    rng = np.random.default_rng(2028)
    shape = (3, 11)
    pos, dd, var = synth.random_sampling(shape[0], shape[1], seed=59, cone=True)
    arrs.update(pos=pos, dir=dd, var=var)
    for tag, kw in (("nerf384", dict(embed_pos_rank=6, embed_dir_rank=4, layer_count=5, layer_width=384, activation_type="tanhExp",
                                     density_activation_type="ReLU", skips=[2], lowpass_alpha_offset=10)),
                    ("nerf512", dict(embed_pos_rank=8, embed_dir_rank=6, layer_count=6, layer_width=512, activation_type="ReLU",
                                     density_activation_type="LeakyReLU", skips=[1, 3], lowpass_alpha_offset=10))):
        net = NeRF(**kw)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_state(
            kw["embed_pos_rank"], kw["embed_dir_rank"], kw["layer_count"], kw["layer_width"], tuple(kw["skips"]), seed=41).items()})
        net.set_iter(1500)
        ups = {"density": torch.from_numpy(rng.standard_normal(shape).astype(np.float32)),
               "color": torch.from_numpy(rng.standard_normal(shape + (3,)).astype(np.float32))}
        with torch.enable_grad():
            net.zero_grad()
            o = net(Sampling(torch.from_numpy(pos.copy()), torch.from_numpy(dd), torch.from_numpy(var)))
            sum((o[k] * ups[k]).sum() for k in ups).backward()
        pre = tag + "_"
        arrs[pre + "config"] = np.array(json.dumps(kw))
        for k in ups:
            arrs[pre + "g_" + k] = npy(ups[k])
            arrs[pre + "out_" + k] = npy(o[k])
        _grad_records(arrs, pre, net, 797, ("layers.0.weight", "layers.0.bias", "layers.3.bias", "outL_density.weight", "outL_color.0.bias",
                                           "outL_color.2.weight", "outL_color.2.bias"))
    save("train_wide_nerf.npz", **arrs)


def gen_train_random():
    """Forty-two architectures drawn at random from what the reference's constructors accept (synth.random_arch(seed, train=True): field kind,
    hidden width 8 .. 512, 2 .. 7 layers, up to three skips, any activation, encoding ranks 1 .. 10) through the reference's own
    forward + backward (NeDDF: hand-written (value, Jacobian) backward passes; NeRF: autograd; NeuS: double backward) with random
    upstream gradients on every output, a ragged number of points each.  Per parameter tensor: gradient norm + one random projection."""
    arrs = {}
    kinds = {"neddf": (NeDDF, ("distance", "density", "color", "fields_penalty", "aux_grad"), 2500),
             "nerf": (NeRF, ("density", "color"), 1500), "neus": (NeuS, ("sdf", "density", "color"), None)}
    for seed in range(100, 142):
        kind, kw = synth.random_arch(seed, train=True)
        cls, keys, it = kinds[kind]
        rng = np.random.default_rng(5000 + seed)
        shape = (int(rng.integers(1, 6)), int(rng.integers(1, 30)))
        pos, dd, var = synth.random_sampling(shape[0], shape[1], seed=6000 + seed, cone=kind != "neus")
        net = cls(**kw)
        net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.arch_state(kind, kw, 7000 + seed).items()})
        if it is not None:
            net.set_iter(it)
        ups = {k: torch.from_numpy(rng.standard_normal(shape + ((3,) if k == "color" else ())).astype(np.float32)) for k in keys}
        with torch.enable_grad():
            net.zero_grad()
            o = net(Sampling(torch.from_numpy(pos.copy()), torch.from_numpy(dd), torch.from_numpy(var)))
            sum((o[k] * ups[k]).sum() for k in ups).backward()
        pre = "s%d_" % seed
        arrs.update({pre + "pos": pos, pre + "dir": dd, pre + "var": var, pre + "config": np.array(json.dumps(dict(kind=kind, kw=kw)))})
        for k in ups:
            arrs[pre + "g_" + k] = npy(ups[k])
            arrs[pre + "out_" + k] = npy(o[k])
        _grad_records(arrs, pre, net, 8000 + seed)
        print("  train_random seed %d %s %s" % (seed, kind, json.dumps(kw)))
    save("train_random.npz", **arrs)


def gen_fields_random():
    """The sixty architectures of the random rendering sweep (synth.random_arch(seed), seeds 0 .. 59: tests/test_gpu_parity.py
    test_random_architectures_vs_oracle) through the REFERENCE's forward in evaluation mode, on that test's own points: pins the
    oracle -- and the HIP path directly -- on every one of them, not only on the fixed architectures of gen_fields."""
    arrs = {}
    classes = {"neddf": NeDDF, "nerf": NeRF, "neus": NeuS}
    for seed in range(60):
        kind, kw = synth.random_arch(seed)
        rng = np.random.default_rng(1000 + seed)
        rays, samples = int(rng.integers(1, 9)), int(rng.integers(1, 50))
        pos, dd, var = synth.random_sampling(rays, samples, seed=2000 + seed)
        net = classes[kind](**kw)
        net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.arch_state(kind, kw, 3000 + seed).items()})
        if kind != "neus":
            net.set_iter(-1)
        net.eval()
        smp_ = Sampling(torch.from_numpy(pos.copy()), torch.from_numpy(dd), torch.from_numpy(var))
        if kind == "neus":              # its normal is torch.autograd.grad of the sdf (neus.py:127-139): needs grad mode even in evaluation
            with torch.enable_grad():
                o = net(smp_)
        else:
            with torch.no_grad():
                o = net(smp_)
        pre = "s%d_" % seed
        arrs[pre + "config"] = np.array(json.dumps(dict(kind=kind, kw=kw)))
        for k, v in o.items():
            arrs[pre + "out_" + k] = npy(v.detach())
    save("fields_random.npz", **arrs)


def gen_render_random():
    """The twenty-four configurations of the rendering sweep (synth.random_render_config) through the REFERENCE's `render_rays`
    (nerf_render.py:109-188).  Its uniforms are the two torch.rand draws of the call, captured by drawing them first and rewinding the
    generator; the pose is what Camera.update_transform makes of the drawn rotation vector (stored as R, T: what the reference used)."""
    arrs = {}
    for seed in range(24):
        c = synth.random_render_config(seed)
        kind, kw = c["kind"], c["kw"]
        render = NeRFRender(network_config=dict(kw, _target_=c["target"]), sample_coarse=c["n_c"], sample_fine=c["n_f"], dist_near=c["near"],
                            dist_far=c["far"], max_dist=c["max_dist"], use_coarse_network=c["two"], sampling_type="cone" if c["cone"] else "point")
        sd_f = synth.arch_state(kind, kw, 500 + seed)
        render.network_fine.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_f.items()})
        if c["two"]:
            render.network_coarse.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.arch_state(kind, kw, 600 + seed).items()})
        render.set_iter(-1)
        cam = Camera(PinholeCalib(c["calib"]), np.r_[c["rotvec"], c["t"]].astype(np.float32))
        cam.update_transform()
        uv = torch.from_numpy(c["uv"])
        torch.manual_seed(50 + seed)
        state = torch.get_rng_state()
        U_c, U_f = torch.rand(c["n"], c["n_c"] + 1), torch.rand(c["n"], c["n_f"] + 1)
        torch.set_rng_state(state)
        pre = "s%d_" % seed
        try:
            if kind == "neus":
                with torch.enable_grad():
                    out = render.render_rays(uv, cam)
            else:
                with torch.no_grad():
                    out = render.render_rays(uv, cam)
        except RuntimeError as e:
            # the reference cannot run every combination its constructors accept: NeDDF over POINT samples dies in
            # `sample_dir.view(-1, 3)` on the expanded direction tensor (neddf.py:210, ray.py:118-126); such a seed keeps its inputs
            # and is compared against the oracle only
            out = {}
            arrs[pre + "reference_error"] = np.array(str(e)[:200])
            print("  render_random seed %d: the reference raised %s" % (seed, str(e)[:80]))
        arrs.update({pre + "R": npy(cam.R), pre + "T": npy(cam.T), pre + "u_coarse": npy(U_c), pre + "u_fine": npy(U_f),
                     pre + "config": np.array(json.dumps(dict(kind=kind, kw=kw, n_c=c["n_c"], n_f=c["n_f"], two=c["two"], cone=c["cone"])))})
        for k, v in out.items():
            arrs[pre + "out_" + k] = npy(v.detach())
        print("  render_random seed %d %s rays %d coarse %d fine %d two %d cone %d" % (seed, kind, c["n"], c["n_c"], c["n_f"], c["two"], c["cone"]))
    save("render_random.npz", **arrs)


def gen_stages_random():
    """The two per-ray stages of the reference on random shapes and hostile inputs: `sample_pdf` (base_neural_render.py:27-115; its
    uniforms captured by drawing first and rewinding the generator) for 2 .. 130 knots, 1 .. 200 samples, with and without the coarse
    knots, weights with zeros, negatives, NaNs, spikes and denormal-small values, repeated distances; `integrate_volume_render`
    (:117-172) for 2 .. 300 samples with negative and saturating densities."""
    render = NeRFRender(network_config=dict(_target_="neddf.network.NeRF", embed_pos_rank=2, embed_dir_rank=1, layer_count=2, layer_width=8,
                                            activation_type="ReLU", density_activation_type="ReLU", skips=[], lowpass_alpha_offset=10),
                        sample_coarse=4, sample_fine=4, dist_near=2.0, dist_far=6.0, max_dist=6.0, use_coarse_network=False, sampling_type="point")
    arrs = {}
    for seed in range(16):
        rng = np.random.default_rng(8100 + seed)
        B, n, nf = int(rng.integers(1, 13)), int(rng.integers(3, 131)), int(rng.integers(1, 201))
        cat = bool(seed % 2 == 0)
        dists = np.sort(rng.uniform(0.5, 7.0, (B, n + 1)).astype(np.float32), axis=1)
        w = (rng.uniform(0, 1, (B, n)) ** int(rng.integers(1, 9))).astype(np.float32)
        kind = seed % 4
        if kind == 1:
            w[rng.uniform(0, 1, w.shape) < 0.3] = 0.0
            w[0] = 0.0
        if kind == 2:
            w[rng.uniform(0, 1, w.shape) < 0.2] *= -1.0
            w[B - 1, n // 2] = np.nan
        if kind == 3:
            w *= 1e-12
            dists[0, 1:4] = dists[0, 1]             # repeated knots
        torch.manual_seed(900 + seed)
        state = torch.get_rng_state()
        u = torch.rand(B, nf)
        torch.set_rng_state(state)
        wt = torch.from_numpy(w.copy())
        out = render.sample_pdf(torch.from_numpy(dists), wt, nf, cat_coarse=cat)
        pre = "sp%d_" % seed
        arrs.update({pre + "dists": dists, pre + "w": w, pre + "u": npy(u), pre + "cat": np.int32(cat), pre + "out": npy(out),
                     pre + "wafter": npy(wt)})
        S = int(rng.integers(2, 301))
        d2 = np.sort(rng.uniform(2.0, 6.0, (B, S)).astype(np.float32), axis=1)
        dens = rng.uniform(-2.0, 30.0, (B, S)).astype(np.float32)
        if kind == 1:
            dens[0] = 1e4
        if kind == 2:
            dens[:, ::3] = -5.0
        col = rng.uniform(-1.0, 2.0, (B, S, 3)).astype(np.float32)
        r = render.integrate_volume_render(torch.from_numpy(d2), torch.from_numpy(dens), torch.from_numpy(col))
        pre = "iv%d_" % seed
        arrs.update({pre + "dists": d2, pre + "dens": dens, pre + "col": col, pre + "weight": npy(r["weight"]), pre + "depth": npy(r["depth"]),
                     pre + "color": npy(r["color"]), pre + "trans": npy(r["transmittance"])})
    arrs["max_dist"] = np.float32(render.max_dist)
    save("stages_random.npz", **arrs)


def gen_rays_random():
    """`Camera.create_rays` (camera.py:155-187, pinhole_calib.py:51-74) and `Ray.get_sampling_cones / get_sampling_points`
    (ray.py:88-194) of the reference for twelve random pinhole cameras, poses, pixel sets of every dtype the callers use (int64 in
    evaluation, int16 in training, float in the reference's tests) and 2 .. 64 sorted distances per ray, with random cone radii."""
    from scipy.spatial.transform import Rotation
    arrs = {}
    for seed in range(12):
        rng = np.random.default_rng(8300 + seed)
        n, S = int(rng.integers(1, 25)), int(rng.integers(2, 65))
        W, H = int(rng.integers(8, 1200)), int(rng.integers(8, 1200))
        calib = np.array([rng.uniform(0.4, 3.0) * W, rng.uniform(0.4, 3.0) * W, 0.5 * W + rng.uniform(-5, 5), 0.5 * H + rng.uniform(-5, 5)])
        rotvec = Rotation.random(random_state=int(rng.integers(0, 1 << 30))).as_rotvec().astype(np.float32)
        t = rng.uniform(-4.0, 4.0, 3).astype(np.float32)
        uv_i = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], 1)
        kind = ("int64", "int16", "float32", "int32")[seed % 4]
        uv = (uv_i.astype(np.float32) + rng.uniform(0, 1, (n, 2)).astype(np.float32)) if kind == "float32" else uv_i.astype(kind)
        cam = Camera(PinholeCalib(calib), np.r_[rotvec, t].astype(np.float32))
        cam.update_transform()
        rays = cam.create_rays(torch.from_numpy(uv))
        dists = np.sort(rng.uniform(0.1, 9.0, (n, S)).astype(np.float32), axis=1)
        radius = float(rng.uniform(1e-4, 2e-3))
        sc = rays.get_sampling_cones(torch.from_numpy(dists), radius)
        sp = rays.get_sampling_points(torch.from_numpy(dists))
        pre = "s%d_" % seed
        arrs.update({pre + "calib": calib, pre + "R": npy(cam.R), pre + "T": npy(cam.T), pre + "uv": uv, pre + "dists": dists,
                     pre + "radius": np.float64(radius), pre + "ray_dir": npy(rays.ray_dir), pre + "ray_orig": npy(rays.ray_orig),
                     pre + "cone_pos": npy(sc.sample_pos), pre + "cone_var": npy(sc.diag_variance), pre + "point_pos": npy(sp.sample_pos)})
        # (sample_dir is the ray direction broadcast, the point samples' variance is zero: asserted here, not stored)
        assert torch.equal(sc.sample_dir, rays.ray_dir.unsqueeze(1).expand_as(sc.sample_dir)) and float(sp.diag_variance.abs().max()) == 0.0
    save("rays_random.npz", **arrs)


def gen_train_render_random():
    """`render_rays` under autograd + backward of the REFERENCE (nerf_render.py:109-188 with the hand-written / autograd backward passes)
    on eight random configurations (synth.random_train_render_config: kinds, one or two networks, cone / point, sample counts, cameras),
    random upstream gradients on every returned tensor.  Stores the outputs and, per parameter tensor of the fine (and coarse) network,
    gradient norm + one random projection; the uniforms are replayed in the test by the same torch seed."""
    arrs = {}
    for seed in range(8):
        c = synth.random_train_render_config(seed)
        kind, kw = c["kind"], c["kw"]
        render = NeRFRender(network_config=dict(kw, _target_=c["target"]), sample_coarse=c["n_c"], sample_fine=c["n_f"], dist_near=c["near"],
                            dist_far=c["far"], max_dist=c["max_dist"], use_coarse_network=c["two"], sampling_type="cone" if c["cone"] else "point")
        render.network_fine.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.arch_state(kind, kw, 500 + seed).items()})
        if c["two"]:
            render.network_coarse.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.arch_state(kind, kw, 600 + seed).items()})
        if kind != "neus":
            render.set_iter(1500)
        cam = Camera(PinholeCalib(c["calib"]), np.r_[c["rotvec"], c["t"]].astype(np.float32))
        cam.update_transform()
        uv = torch.from_numpy(c["uv"])
        rng = np.random.default_rng(8500 + seed)
        torch.manual_seed(70 + seed)
        with torch.enable_grad():
            render.zero_grad()
            out = render.render_rays(uv, cam)
            ups = {k: torch.from_numpy(rng.standard_normal(tuple(v.shape)).astype(np.float32)) for k, v in out.items()}
            sum((out[k] * ups[k]).sum() for k in out).backward()
        pre = "s%d_" % seed
        arrs.update({pre + "R": npy(cam.R), pre + "T": npy(cam.T),
                     pre + "config": np.array(json.dumps(dict(kind=kind, kw=kw, n_c=c["n_c"], n_f=c["n_f"], two=c["two"], cone=c["cone"]))),
                     pre + "keys": np.array(json.dumps(list(out.keys())))})
        for k in out:
            arrs[pre + "out_" + k] = npy(out[k].detach())
            arrs[pre + "g_" + k] = npy(ups[k])
        _grad_records(arrs, pre + "fine_", render.network_fine, 8600 + seed)
        if c["two"]:
            _grad_records(arrs, pre + "coarse_", render.network_coarse, 8700 + seed)
        print("  train_render_random seed %d %s rays %d coarse %d fine %d two %d cone %d" % (seed, kind, c["n"], c["n_c"], c["n_f"], c["two"], c["cone"]))
    save("train_render_random.npz", **arrs)


# --------------------------------------------------------------------------
def gen_dataset():
    """the reference's own NeRFSyntheticDataset (nerf_synthetic_dataset.py:25-84, with cv2.imread backed
    by PIL) on a two-frame crop of data/bunny_smoke's test split.  The crop (tests/golden/bunny_mini/: two 72x56 RGBA
    PNGs + a two-frame transforms_test.json) is data of the reference's dataset, written here once."""
    from PIL import Image
    from neddf.dataset import NeRFSyntheticDataset
    src = os.path.join(REF, "data/bunny_smoke")
    dst = os.path.join(HERE, "bunny_mini")
    os.makedirs(os.path.join(dst, "test"), exist_ok=True)
    tf = json.load(open(os.path.join(src, "transforms_test.json")))
    keep = [tf["frames"][0], tf["frames"][7]]
    mini = {"camera_angle_x": tf["camera_angle_x"], "frames": []}
    for i, fr in enumerate(keep):
        img = Image.open(os.path.join(src, fr["file_path"] + ".png"))
        # non-square window across the silhouette edge: alpha takes 0, 255 and in-between values
        crop = img.crop((150, 200, 150 + 72, 200 + 56))
        name = "./test/m_%d" % i
        crop.save(os.path.join(dst, name + ".png"))
        mini["frames"].append({"file_path": name, "transform_matrix": fr["transform_matrix"]})
    json.dump(mini, open(os.path.join(dst, "transforms_test.json"), "w"))
    arrs = {}
    for tag, use_mask in (("mask", True), ("nomask", False)):
        ds = NeRFSyntheticDataset(dst, "test", use_mask=use_mask)
        item = ds[1]
        arrs[tag + "_calib"] = np.asarray(ds.camera_calib_params)
        arrs[tag + "_camera_params"] = np.asarray(ds.camera_params)
        arrs[tag + "_rgb_images"] = np.asarray(ds.rgb_images)
        arrs[tag + "_mask_images"] = np.asarray(ds.mask_images)
        arrs[tag + "_item1_rgb"] = np.asarray(item["rgb_images"])
        arrs[tag + "_item1_camera_params"] = np.asarray(item["camera_params"])
    a = arrs["mask_mask_images"]
    assert a.min() == 0 and a.max() == 255 and ((a > 0) & (a < 255)).any(), "crop must straddle the silhouette"
    save("dataset_bunny_mini.npz", **arrs)


def gen_eval_harness():
    """the reference's OWN eval harness -- `NeRFTrainer` built from the shipped bunny_smoke config with
    the dataset pointed at tests/golden/bunny_mini (test split), `load_pretrained_model` on the shipped checkpoint, then
    `set_iter(-1)` + `render_test(dir, camera_id, 1)` exactly as `render_all` drives it (base_trainer.py:123-188) -- on the CPU.
    Nothing of the conversion is restated here: `cv2.imwrite` is a stub that RECORDS the arrays the reference hands it (file name
    -> uint8 array, channels in the reference's B,G,R order), so the colour / depth / ground-truth images below are what the
    reference would have written.  72 x 56 = 4 032 rays at chunk 1024: three full chunks and a short one of 960.

    skimage is absent here.  `peak_signal_noise_ratio` / `structural_similarity` are stubs that record their ARGUMENTS (so the
    fixture proves which arrays the reference compares, and in which order); the PSNR stored beside them is the published
    definition skimage implements for uint8 inputs, 10 log10(255^2 / mean((a - b)^2)) in float64.  SSIM stays unpinned."""
    captured, metric_args = {}, []
    sys.modules["cv2"].imwrite = lambda path, arr: captured.__setitem__(os.path.basename(str(path)), np.array(arr, copy=True))
    sk, skm = types.ModuleType("skimage"), types.ModuleType("skimage.metrics")

    def psnr(a, b):
        metric_args.append(("psnr", np.array(a, copy=True), np.array(b, copy=True)))
        assert a.dtype == np.uint8 and b.dtype == np.uint8
        mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
        return 10.0 * np.log10(255.0 ** 2 / mse)

    def ssim(a, b, channel_axis=None):
        metric_args.append(("ssim", np.array(a, copy=True), np.array(b, copy=True), channel_axis))
        return float("nan")

    skm.peak_signal_noise_ratio, skm.structural_similarity = psnr, ssim
    sk.metrics = skm
    sys.modules["skimage"], sys.modules["skimage.metrics"] = sk, skm
    tb = types.ModuleType("torch.utils.tensorboard")

    class SummaryWriter:
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

    tb.SummaryWriter = SummaryWriter
    sys.modules["torch.utils.tensorboard"] = tb

    class Cfg(dict):
        """attribute access like omegaconf's DictConfig (base_trainer.py reads config.dataset, config.loss.functions, ...)"""
        def __getattr__(self, k):
            v = self[k]
            return Cfg(v) if isinstance(v, dict) else v

    from neddf.trainer import NeRFTrainer
    import io
    import contextlib
    from pathlib import Path
    cfg = yaml.safe_load(open(os.path.join(REF, "pretrained/bunny_smoke/.hydra/config.yaml")))
    cfg["dataset"]["dataset_dir"] = os.path.join(HERE, "bunny_mini")
    cfg["dataset"]["data_split"] = "test"                  # run_eval.py:27 override
    tcfg = dict(cfg["trainer"], device="cpu")
    tcfg.pop("_target_")
    trainer = NeRFTrainer(global_config=Cfg(cfg), **tcfg)
    # the checkpoint was saved from cuda:0 and base_trainer.py:121 calls torch.load without map_location: on this GPU-less
    # container the storages have to be mapped to the CPU (an accommodation of the environment, not of the algorithm)
    torch_load = torch.load
    torch.load = lambda f, *a, **k: torch_load(f, *a, **dict(k, map_location="cpu"))
    try:
        trainer.load_pretrained_model(Path(REF) / "pretrained/bunny_smoke/models/model_02000.pth")
    finally:
        torch.load = torch_load
    trainer.neural_render.set_iter(-1)                      # render_all, base_trainer.py:185
    arrs = {"seed": np.int32(11), "chunk": np.int32(trainer.chunk), "num_threads": np.int32(torch.get_num_threads())}
    for cam_id in (0, 1):
        torch.manual_seed(11 + cam_id)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            trainer.render_test(Path("/nonexistent"), cam_id, 1)
        line = [ln for ln in buf.getvalue().splitlines() if ln.startswith("psnr:")][0]
        for suffix in ("rgb", "rgb_gt", "depth"):
            arrs["cam%d_%s" % (cam_id, suffix)] = captured.pop("%03d_%s.png" % (cam_id, suffix))
        assert not captured
        (k0, a0, b0), (k1, a1, b1, axis) = metric_args[-2:]
        assert k0 == "psnr" and k1 == "ssim" and axis == 2
        for a in (a0, a1):
            assert np.array_equal(a, arrs["cam%d_rgb" % cam_id])                     # first argument: the render
        for b in (b0, b1):
            assert np.array_equal(b, arrs["cam%d_rgb_gt" % cam_id])                  # second: the ground truth
        arrs["cam%d_psnr" % cam_id] = np.float64(line.split("psnr: ")[1].split(",")[0])
        arrs["cam%d_printout" % cam_id] = np.array(line)
        print("  eval harness camera %d: %s  rgb %s depth %s" % (cam_id, line, arrs["cam%d_rgb" % cam_id].shape,
                                                                  arrs["cam%d_depth" % cam_id].shape))
    # half-resolution render of camera 0 (the trainer's periodic test rendering, nerf_trainer.py: render_test(dir, id, 2)):
    # no metrics printed, the ground truth written at full size
    torch.manual_seed(13)
    trainer.render_test(Path("/nonexistent"), 0, 2)
    for suffix in ("rgb", "rgb_gt", "depth"):
        arrs["ds2_%s" % suffix] = captured.pop("000_%s.png" % suffix)
    save("eval_harness.npz", **arrs)


def gen_grids(render):
    """NeRFRender.render_field_slice (nerf_render.py:263-336; scalar fields as the uint8 image handed to
    cv2.applyColorMap) and BaseNeuralField.voxelize (base_neuralfield.py:49-79) of the shipped bunny_smoke field."""
    arrs = {}
    for tag, (t, size, res) in {"a": (0.1, 1.1, 48), "b": (-0.25, 0.8, 20)}.items():
        sl = render.render_field_slice(t, size, res)
        arrs["slice_%s_args" % tag] = np.array([t, size, res], np.float64)
        for k, v in sl.items():
            arrs["slice_%s_%s" % (tag, k)] = np.asarray(v)
    net = render.network_fine
    net.eval()
    arrs["vox_density"] = net.voxelize("density", 1.1, 12, chunk=500)
    arrs["vox_distance"] = net.voxelize("distance", 0.9, 9, chunk=65536)
    net.train(True)
    save("bunny_grids.npz", **arrs)


def gen_config_digests():
    """Digest of the VALUES of every file under the reference's config/ (canonical JSON of the parsed YAML), so that the
    shipped config/*.yaml can be checked value for value without the reference tree.  The one deliberate difference is
    recorded as an exclusion: trainer/test.yaml runs on "cpu" in the reference, this build has no CPU compute path."""
    import hashlib
    out = {}
    root = os.path.join(REF, "config")
    for d, _, files in os.walk(root):
        for f in sorted(files):
            if not f.endswith(".yaml"):
                continue
            rel = os.path.relpath(os.path.join(d, f), root)
            val = yaml.safe_load(open(os.path.join(d, f)))
            if rel == "trainer/test.yaml":
                val.pop("device")
            out[rel] = hashlib.sha256(json.dumps(val, sort_keys=True).encode()).hexdigest()
    json.dump(out, open(os.path.join(HERE, "config_digests.json"), "w"), indent=1, sort_keys=True)
    print("wrote config_digests.json (%d files)" % len(out))


def gen_fp64():
    """The reference field evaluated in DOUBLE precision (torch default dtype float64: same code, same checkpoint, the
    stage-golden sample points) -- the yardstick for fp32 noise: density = (1 - |grad D, aux|) / D amplifies rounding, and
    the reference's own fp32 result differs from this by ~1e-4 abs on a range of +-40."""
    torch.set_default_dtype(torch.float64)
    cfg = yaml.safe_load(open(os.path.join(REF, "pretrained/bunny_smoke/.hydra/config.yaml")))
    rcfg = dict(cfg["render"])
    rcfg.pop("_target_")
    render = NeRFRender(network_config=cfg["network"], **rcfg)
    render.load_state_dict(torch.load(os.path.join(REF, "pretrained/bunny_smoke/models/model_02000.pth"), map_location="cpu"))
    render.set_iter(-1)
    render.network_fine.eval()
    assert next(render.network_fine.parameters()).dtype == torch.float64
    g = np.load(os.path.join(HERE, "bunny_stages.npz"))
    arrs = {}
    for tag in ("c", "f"):
        smp = Sampling(*(torch.from_numpy(g[tag + "_" + k]).double() for k in ("pos", "dir", "var")))
        out = render.network_fine(smp)
        for k in ("density", "distance"):
            arrs["%s_%s" % (tag, k)] = npy(out[k])
            assert arrs["%s_%s" % (tag, k)].dtype == np.float64
    save("bunny_field_fp64.npz", **arrs)
    torch.set_default_dtype(torch.float32)


NEGBIAS_KW = dict(embed_pos_rank=10, embed_dir_rank=4, ddf_layer_count=8, ddf_layer_width=256, col_layer_count=4,
                  col_layer_width=256, d_near=0.001, activation_type="tanhExp", density_activation_type="LeakyReLU",
                  skips=[4], lowpass_alpha_offset=10,
                  penalty_weight={"constraints_aux_grad": 0.05, "constraints_dDdt": 1.0, "constraints_color": 0.0001,
                                  "range_distance": 1.0, "range_aux_grad": 1.0, "range_color": 0.1})


def gen_negbias():
    """The regime in which the fused kernels' reduced-cost arithmetic differs from the reference's:
    tanhExp networks whose pre-activations sit at e^x << 0.3 on most units (`synth.neddf_state_negbias`: 80 % of the hidden
    biases in [-15, -1]) -- where 1 - 2 / (e^(2 e^x) + 1) has lost its relative accuracy --, a small D (d_near = 1e-3, distance
    head biased to softplus ~ 0.05: the 1/D of neddf.py:239 amplifies), sample positions out to |pos| = 6 under a rank-10
    encoding (sincos arguments to 2^9 * 6 rad) and zero-variance points (the high frequencies keep their weight).  fp32
    outputs of the reference AND its evaluation in double; the pre-activation quantiles the reference saw are stored so that the
    fixture documents its own regime.  Plus one 64-ray render_rays through NeRFRender on the same network."""
    kw = NEGBIAS_KW
    pos, d, var = synth.wide_sampling(8, 48, seed=5)
    sd = synth.neddf_state_negbias(kw["embed_pos_rank"], kw["embed_dir_rank"], kw["ddf_layer_count"], kw["ddf_layer_width"],
                                   kw["col_layer_count"], kw["col_layer_width"], tuple(kw["skips"]), seed=7)
    arrs = dict(pos=pos, dir=d, var=var, config=np.array(json.dumps(kw)))
    for dtype, suffix in ((torch.float32, ""), (torch.float64, "_fp64")):
        torch.set_default_dtype(dtype)
        net = NeDDF(**kw)
        print("negbias", dtype, net.load_state_dict({k: v.to(dtype) for k, v in to_torch_sd(sd).items()}))
        seen = []
        inner = net.activation

        def spy(x, J, inner=inner, seen=seen):
            seen.append(npy(x).reshape(-1))
            return inner(x, J)
        net.activation = spy
        smp_t = Sampling(*(torch.from_numpy(x).to(dtype) for x in (pos, d, var)))
        for it in (-1, 2500):
            net.set_iter(it)
            del seen[:]
            out = net(smp_t)
            tag = "eval" if it == -1 else "it%d" % it
            for k, v in out.items():
                arrs["%s_%s%s" % (tag, k, suffix)] = npy(v)
            if dtype == torch.float32 and it == -1:
                z = np.concatenate(seen)
                arrs["preact_quantiles"] = np.quantile(z, [0.0, 0.05, 0.25, 0.5, 0.75, 0.95, 1.0]).astype(np.float32)
                arrs["preact_frac_below_m1"] = np.float32((z < -1.0).mean())
                print("pre-activations: quantiles", arrs["preact_quantiles"], "fraction below -1:", arrs["preact_frac_below_m1"])
        torch.set_default_dtype(torch.float32)
    for k in ("distance", "density", "aux_grad", "color", "fields_penalty"):
        a, b = arrs["eval_" + k], arrs["eval_" + k + "_fp64"]
        print("  %-15s range [%.4g, %.4g]  reference fp32-vs-fp64 max abs %.3g" % (k, a.min(), a.max(), np.abs(a - b).max()))
    save("neddf_negbias.npz", **arrs)

    # 64 rays through the whole renderer on that network (cone sampling, 32 + 64 samples, camera at radius ~4: |pos| up to ~6)
    ncfg = dict(kw, _target_="neddf.network.NeDDF")
    render = NeRFRender(network_config=ncfg, sample_coarse=32, sample_fine=64, dist_near=2.0, dist_far=6.0, max_dist=6.0,
                        use_coarse_network=False, sampling_type="cone")
    render.network_fine.load_state_dict(to_torch_sd(sd))
    render.set_iter(-1)
    calib = np.array([1111.1, 1111.1, 400.0, 400.0])
    cam = Camera(PinholeCalib(calib), np.array([0.9, -0.5, 0.3, 1.2, -2.9, 2.4], dtype=np.float32))
    cam.update_transform()
    rs = np.random.RandomState(4)
    uv = torch.from_numpy(rs.randint(0, 800, (64, 2)).astype(np.int64))
    torch.manual_seed(11)
    state = torch.get_rng_state()
    u_c = torch.rand(64, 33)
    u_f = torch.rand(64, 65)
    torch.set_rng_state(state)
    out = render.render_rays(uv, cam)
    r = dict(uv=npy(uv), R=npy(cam.R), T=npy(cam.T), calib=calib.astype(np.float32), u_coarse=npy(u_c), u_fine=npy(u_f))
    for k, v in out.items():
        r["out_" + k] = npy(v)
        print("  render_rays %-22s range [%.4g, %.4g]" % (k, float(v.min()), float(v.max())))
    save("neddf_negbias_render_rays.npz", **r)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "train_nerf":
        gen_train_nerf()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "train":
        gen_train()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "neus":
        gen_neus()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "train_neus":
        gen_train_neus()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "train_wide":
        from neddf.ray import Sampling  # noqa: F401
        gen_train_wide()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "train_wide_nerf":
        from neddf.ray import Sampling  # noqa: F401
        gen_train_wide_nerf()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "train_render_random":
        gen_train_render_random()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "rays_random":
        gen_rays_random()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "stages_random":
        gen_stages_random()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "render_random":
        gen_render_random()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "fields_random":
        from neddf.ray import Sampling  # noqa: F401
        gen_fields_random()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "train_random":
        from neddf.ray import Sampling  # noqa: F401
        gen_train_random()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "train_widths":
        gen_train_widths()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "fp64":
        gen_fp64()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "fields":
        from neddf.ray import Sampling  # noqa: F401  (gen_bunny normally imports the reference first)
        gen_fields()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "negbias":
        from neddf.ray import Sampling  # noqa: F401
        gen_negbias()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "configs":
        gen_config_digests()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "dataset":
        gen_dataset()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "eval_harness":
        gen_eval_harness()
        sys.exit(0)
    r = gen_bunny()
    if len(sys.argv) > 1 and sys.argv[1] == "grids":
        gen_grids(r)
        sys.exit(0)
    gen_grids(r)
    gen_dataset()
    gen_config_digests()
    gen_ops()
    gen_fields()
    gen_render_edges(r)
    gen_neus()
    gen_train()
    gen_train_nerf()
    gen_train_neus()
    gen_train_widths()
    gen_train_wide()
    gen_negbias()
    gen_fp64()          # last: switches torch's default dtype while it runs
    gen_train_wide_nerf()
    gen_train_random()
    gen_fields_random()
    gen_render_random()
    gen_stages_random()
    gen_rays_random()
    gen_train_render_random()
    gen_eval_harness()  # replaces cv2.imwrite / skimage / tensorboard stubs: keep it last
