"""Pins the CPU oracle (oracle/) against golden vectors generated from the
reference PyTorch renderer (tests/golden/gen_goldens.py).  CPU only."""
import json

import numpy as np
import pytest
from conftest import BUNNY_CFG, assert_close, golden

import synth
from oracle import oracle as orc


def test_raygen(bunny_stages):
    g = bunny_stages
    d, o = orc.create_rays(g["uv"], g["R"], g["T"], g["calib"])
    assert_close(d, g["ray_dir"], 1e-6, 1e-7, "ray_dir")
    assert np.array_equal(o, g["ray_orig"])


def test_sample_coarse_bitexact(bunny_stages):
    g = bunny_stages
    dc = orc.sample_coarse(g["u_coarse"], float(g["dist_near"]), float(g["dist_far"]))
    assert np.array_equal(dc, g["dists_coarse"])


def test_sampling(bunny_stages):
    g = bunny_stages
    pos, d, var = orc.sampling(g["ray_dir"], g["ray_orig"], g["dists_coarse"], float(g["ray_radius"]))
    assert_close(pos, g["c_pos"], 1e-6, 1e-7, "cone pos")
    assert np.array_equal(d, g["c_dir"])
    assert_close(var, g["c_var"], 2e-5, 1e-12, "cone var")
    pos, d, var = orc.sampling(g["ray_dir"], g["ray_orig"], g["dists_fine"], float(g["ray_radius"]))
    assert_close(pos, g["f_pos"], 1e-6, 1e-7, "cone pos fine")
    assert_close(var, g["f_var"], 1e-4, 1e-12, "cone var fine")
    pos, d, var = orc.sampling(g["ray_dir"], g["ray_orig"], g["dists_coarse"], None)
    assert_close(pos, g["p_pos"], 1e-6, 1e-7, "point pos")
    assert np.array_equal(var, g["p_var"])


def test_ops():
    g = golden("ops.npz")
    for kind, key in (("LeakyReLU", "leaky"), ("ReLU", "relu"), ("tanhExp", "tanhexp"), ("softplus", "softplus")):
        y, G = orc.activation_grad(kind, g["act_x"], g["act_J"])
        assert_close(y, g[key + "_y"], 2e-6, 1e-7, key + " y")
        assert_close(G, g[key + "_G"], 2e-6, 1e-7, key + " G")
    y, G = orc.activation_grad("sigmoid", g["act_x"][:, :1], g["act_J"][:, :, :1])
    assert_close(y, g["sigmoid_y"], 2e-6, 1e-7, "sigmoid y")
    assert_close(G, g["sigmoid_G"], 2e-6, 1e-7, "sigmoid G")
    y, G = orc.linear_grad(g["lin_x"], g["lin_J"], g["lin_w"], g["lin_b"])
    assert_close(y, g["lin_y"], 1e-6, 1e-7, "linear y")
    assert_close(G, g["lin_G"], 1e-6, 1e-7, "linear G")
    y, G = orc.pe_grad(g["pe4_x"], g["pe4_J"], None, 4)
    assert_close(y, g["pe4_y"], 1e-6, 2e-7, "pe4 y")
    assert_close(G, g["pe4_G"], 1e-6, 1e-6, "pe4 G")
    w = orc.pe_weights(g["pe10_var"], 10)
    assert_close(w, g["pe10_w"], 1e-6, 1e-30, "pe weights")
    eye = np.broadcast_to(np.eye(3, dtype=np.float32), (12, 3, 3))
    y, G = orc.pe_grad(g["pe10_x"], eye, g["pe10_gradscale"] * g["pe10_w"], 10)
    assert_close(y, g["pe10_y"], 1e-6, 3e-7, "pe10 y")
    assert_close(G, g["pe10_G"], 1e-6, 3e-7, "pe10 G")
    assert_close(orc.pe(g["pe10_x"], None, 4), g["pedir_y"], 1e-6, 2e-7, "pe dir")
    for alpha in (3.25, 9.5, 10.0):
        lp = np.repeat(orc.lowpass_scale(alpha, 10), 3)[None]
        assert np.array_equal(lp, g["lowpass_%g" % alpha])


def test_integrate(bunny_stages):
    g = bunny_stages
    r = orc.integrate(g["dists_coarse"], g["c_density"], g["c_color"], float(g["max_dist"]))
    assert not r["nan"]
    assert_close(r["weight"], g["weight_coarse_raw"], 1e-5, 1e-7, "weight")
    assert_close(r["color"], g["ic_color"], 1e-5, 1e-6, "color")
    assert_close(r["depth"], g["ic_depth"], 1e-5, 1e-6, "depth")
    assert_close(r["transmittance"], g["ic_transmittance"], 1e-5, 1e-7, "trans")
    assert_close(orc.integrate_penalty(g["dists_coarse"], g["c_fields_penalty"]), g["pen_coarse"], 1e-5, 1e-7)
    e = golden("render_edges.npz")
    r = orc.integrate(e["iv_dists"], e["iv_dens"], e["iv_col"], float(e["iv_max_dist"]))
    assert_close(r["weight"], e["iv_weight"], 2e-5, 1e-7, "edge weight")
    assert_close(r["color"], e["iv_color"], 2e-5, 1e-6, "edge color")
    assert_close(r["depth"], e["iv_depth"], 2e-5, 1e-6, "edge depth")
    assert_close(r["transmittance"], e["iv_trans"], 2e-5, 1e-12, "edge trans")
    d2 = np.broadcast_to(np.linspace(0.0, 2.0, 64, dtype=np.float32), (32, 64))
    r = orc.integrate(d2, np.ones((32, 64), np.float32), np.ones((32, 64, 3), np.float32), float(e["iv_max_dist"]))
    assert_close(r["color"], e["ivc_color"], 1e-6, 1e-7)
    assert_close(r["depth"], e["ivc_depth"], 1e-6, 1e-7)


def test_sample_pdf_bitexact(bunny_stages):
    """Same weights/dists/uniforms => bit-identical samples (SURVEY App.A N4)."""
    g = bunny_stages
    w = g["weight_coarse_raw"].copy()
    out, ids, fb = orc.sample_pdf(g["dists_coarse"], w, g["u_fine"], True)
    assert not fb
    assert np.array_equal(out, g["dists_fine"])
    assert np.array_equal(w, g["weight_coarse"])          # in-place sanitisation
    assert ids.min() >= 1 and ids.max() <= 64
    e = golden("render_edges.npz")
    for cat, tag in ((True, "cat"), (False, "nocat")):
        w = e["sp_w"].copy()
        out, _, fb = orc.sample_pdf(e["sp_dists"], w, e["sp_u"], cat)
        assert np.array_equal(out, e["sp_%s_out" % tag]), tag
        assert np.array_equal(w, e["sp_%s_wafter" % tag], equal_nan=True)


def test_sample_pdf_nan_fallback():
    d = np.sort(np.random.default_rng(0).uniform(2, 6, (3, 9)).astype(np.float32), axis=1)
    d[1, 3] = np.nan
    w = np.ones((3, 8), np.float32)
    u = np.random.default_rng(1).uniform(0, 1, (3, 5)).astype(np.float32)
    out, _, fb = orc.sample_pdf(d, w, u, True)
    assert fb and np.isfinite(out).all()
    assert np.allclose(out[2], np.linspace(d[0, 0], d[0, -1], 14), rtol=1e-6)


def test_neddf_bunny_field(bunny_weights, bunny_stages):
    g = bunny_stages
    net = orc.NeDDFOracle(bunny_weights, **BUNNY_CFG)
    for tag in ("c", "f"):
        o = net.forward(g[tag + "_pos"], g[tag + "_dir"], g[tag + "_var"])
        assert_close(o["distance"], g[tag + "_distance"], 1e-4, 1e-6, tag + " distance")
        assert_close(o["aux_grad"], g[tag + "_aux_grad"], 1e-4, 1e-6, tag + " aux")
        assert_close(o["color"], g[tag + "_color"], 1e-4, 2e-5, tag + " color")
        # density = (1 - |grad|)/D amplifies fp32 noise (SURVEY App.A N7): abs 5e-5 on range ~40
        assert_close(o["density"], g[tag + "_density"], 1e-4, 3e-4, tag + " density")
        assert_close(o["fields_penalty"], g[tag + "_fields_penalty"], 1e-4, 1e-5, tag + " penalty")


@pytest.mark.parametrize("name", ["neddf_relu", "neddf_tanhexp", "neddf_leaky", "neddf_w128", "neddf_w384", "neddf_w192", "neddf_skips2"])
def test_neddf_synth(name):
    g = golden(name + ".npz")
    kw = json.loads(str(g["config"]))
    sd = synth.neddf_state(kw["embed_pos_rank"], kw["embed_dir_rank"], kw["ddf_layer_count"], kw["ddf_layer_width"],
                           kw["col_layer_count"], kw["col_layer_width"], tuple(kw["skips"]), seed=7)
    net = orc.NeDDFOracle(sd, **kw)
    for it, tag in ((-1, "eval"), (2500, "it2500")):
        net.set_iter(it)
        o = net.forward(g["pos"], g["dir"], g["var"])
        for k in ("distance", "aux_grad", "color", "density", "fields_penalty"):
            assert_close(o[k], g["%s_%s" % (tag, k)], 1e-4, 1e-5, "%s %s %s" % (name, tag, k))


@pytest.mark.parametrize("name", ["nerf_relu", "nerf_tanhexp", "nerf_w128", "nerf_w384", "nerf_skips2"])
def test_nerf_synth(name):
    g = golden(name + ".npz")
    kw = json.loads(str(g["config"]))
    sd = synth.nerf_state(kw["embed_pos_rank"], kw["embed_dir_rank"], kw["layer_count"], kw["layer_width"],
                          tuple(kw["skips"]), seed=11)
    net = orc.NeRFOracle(sd, **kw)
    for it, tag in ((-1, "eval"), (2500, "it2500")):
        net.set_iter(it)
        o = net.forward(g["pos"], g["dir"], g["var"])
        for k in ("density", "color"):
            assert_close(o[k], g["%s_%s" % (tag, k)], 1e-4, 1e-5, "%s %s %s" % (name, tag, k))


def test_cpu_baseline_implementation_matches_the_oracle(bunny_weights, bunny_stages):
    """bench.py's CPU baseline (oracle/neddf_cpu_fast.c: eval-minimal, reverse-mode distance gradient, blocked GEMM micro-kernels,
    vectorised activations) computes the same field values as the operation-order-exact oracle: the 1e-4 gates of the HIP
    path (density with the 3e-4 abs term of its fp32 noise), on the shipped network and on a synthetic one with
    another width and two skip connections."""
    g = bunny_stages
    net = orc.NeDDFOracle(bunny_weights, **BUNNY_CFG)
    for tag in ("c", "f"):
        a = net.forward(g[tag + "_pos"], g[tag + "_dir"], g[tag + "_var"])
        b = net.forward_fast(g[tag + "_pos"], g[tag + "_dir"], g[tag + "_var"])
        assert "fields_penalty" not in b
        for k in ("distance", "aux_grad", "color"):
            assert_close(b[k], a[k], 1e-4, 1e-5, "%s %s" % (tag, k))
        assert_close(b["density"], a["density"], 1e-4, 3e-4, tag + " density")
    for name in ("neddf_w192", "neddf_skips2", "neddf_relu"):
        gg = golden(name + ".npz")
        kw = json.loads(str(gg["config"]))
        sd = synth.neddf_state(kw["embed_pos_rank"], kw["embed_dir_rank"], kw["ddf_layer_count"], kw["ddf_layer_width"],
                               kw["col_layer_count"], kw["col_layer_width"], tuple(kw["skips"]), seed=7)
        n2 = orc.NeDDFOracle(sd, **kw)
        n2.set_iter(2500)
        b = n2.forward_fast(gg["pos"], gg["dir"], gg["var"])
        for k in ("distance", "aux_grad", "color", "density"):
            assert_close(b[k], gg["it2500_" + k], 1e-4, 1e-5, "%s %s vs reference golden" % (name, k))


def test_render_rays_end_to_end(bunny_weights, bunny_stages):
    g = bunny_stages
    net = orc.NeDDFOracle(bunny_weights, **BUNNY_CFG)
    out = orc.render_rays(net, net, g["uv"], g["R"], g["T"], g["calib"], g["u_coarse"], g["u_fine"],
                          float(g["dist_near"]), float(g["dist_far"]), float(g["max_dist"]), "cone")
    assert_close(out["color"], g["out_color"], 1e-4, 1e-5, "color")
    assert_close(out["depth"], g["out_depth"], 1e-4, 1e-5, "depth")
    assert_close(out["transmittance"], g["out_transmittance"], 1e-4, 1e-5, "transmittance")
    assert_close(out["color_coarse"], g["out_color_coarse"], 1e-4, 1e-5, "color_coarse")
    assert_close(out["depth_coarse"], g["out_depth_coarse"], 1e-4, 1e-5, "depth_coarse")
    # end-to-end sample positions: identical up to knot flips ("bit-exact indices")
    flips = np.mean(out["dists_fine"] != g["dists_fine"])
    assert flips < 0.5
    assert_close(out["dists_fine"], g["dists_fine"], 1e-4, 1e-4, "dists_fine")


def test_nerf_render_rays():
    g = golden("nerf_render_rays.npz")
    kw = dict(embed_pos_rank=10, embed_dir_rank=4, layer_count=8, layer_width=256, activation_type="ReLU",
              density_activation_type="ReLU", skips=[4], lowpass_alpha_offset=10)
    nc = orc.NeRFOracle(synth.nerf_state(seed=21), **kw)
    nf = orc.NeRFOracle(synth.nerf_state(seed=22), **kw)
    out = orc.render_rays(nc, nf, g["uv"], g["R"], g["T"], g["calib"], g["u_coarse"], g["u_fine"], 2.0, 6.0, 6.0,
                          "point")
    for k in ("color", "depth", "transmittance", "color_coarse", "depth_coarse", "transmittance_coarse"):
        assert_close(out[k], g["out_" + k], 1e-4, 1e-5, k)


@pytest.mark.parametrize("name", ["neus_relu", "neus_tanhexp", "neus_w128_384", "neus_w320_64"])
def test_neus_synth(name):
    """NeuS (neus.py:101-162): forward-mode normals of the oracle vs the reference's autograd normals."""
    g = golden(name + ".npz")
    kw = json.loads(str(g["config"]))
    sd = synth.neus_state(kw["embed_pos_rank"], kw["embed_dir_rank"], kw["sdf_layer_count"], kw["sdf_layer_width"],
                          kw["col_layer_count"], kw["col_layer_width"], tuple(kw["skips"]), kw["init_variance"], seed=13)
    net = orc.NeuSOracle(sd, **kw)
    o = net.forward(g["pos"], g["dir"])
    for k in ("sdf", "density", "color"):
        assert_close(o[k], g["eval_" + k], 1e-4, 1e-5, "%s %s" % (name, k))


# ----------------------------------------------------------------- configs[4]: NDC rays, bf16 operands (not in the reference)
def test_ndc_rays_projective_identity():
    """The reference has no NDC code, so the restatement is pinned on what it is derived from (NeRF paper, appendix C):
    a world point o_n + t d (o_n = the ray's intersection with the near plane) maps to o' + t' d' with
    t' = 1 - o_n,z / (o_n,z + t d_z) under the perspective map (x, y, z) -> (-fx/(W/2) x/z, -fy/(H/2) y/z, 1 + 2 near/z)."""
    rng = np.random.default_rng(0)
    B, W, H, fx, fy, near = 64, 1008, 756, 815.1, 809.3, 1.0
    d = rng.standard_normal((B, 3)).astype(np.float32)
    d[:, 2] = -np.abs(d[:, 2]) - 0.5
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = rng.uniform(-0.2, 0.2, (B, 3)).astype(np.float32)
    nd, no = orc.rays_to_ndc(d, o, W, H, fx, fy, near)
    d64, o64 = d.astype(np.float64), o.astype(np.float64)
    on = o64 + (-(near + o64[:, 2]) / d64[:, 2])[:, None] * d64
    for t in (0.0, 0.25, 2.0, 40.0, 1e4):
        p = on + t * d64
        want = np.stack([-fx / (W / 2) * p[:, 0] / p[:, 2], -fy / (H / 2) * p[:, 1] / p[:, 2], 1 + 2 * near / p[:, 2]], 1)
        tp = 1 - on[:, 2] / (on[:, 2] + t * d64[:, 2])
        assert np.abs(no + tp[:, None] * nd - want).max() < 2e-6
    assert np.abs(no[:, 2] + 1).max() < 1e-6            # the NDC origin sits on the near face z' = -1
    assert np.abs(no[:, 2] + nd[:, 2] - 1).max() < 1e-6  # and t' = 1 is the far face z' = +1


def test_bf16_rounding_and_emulation(bunny_weights):
    """bf16 emulation of configs[4]: nearest-even rounding, and how far it moves the field outputs from fp32."""
    x = np.array([1.0, 1.00390625, 1.005859375, 1.01171875, -2.5e-3, 65535.0, 0.0], np.float32)
    want = np.array([1.0, 1.0, 1.0078125, 1.015625, -2.50244140625e-3, 65536.0, 0.0], np.float32)
    assert np.array_equal(orc.bf16_round(x), want)      # ties go to the even mantissa (1.00390625 -> 1.0, 1.01171875 -> 1.015625)
    pos, d, var = synth.random_sampling(4, 16, seed=5)
    a = orc.NeDDFOracle(bunny_weights, **BUNNY_CFG).forward(pos, d, var)
    b = orc.NeDDFOracle(bunny_weights, bf16=True, **BUNNY_CFG).forward(pos, d, var)
    assert np.abs(a["distance"] - b["distance"]).max() < 1e-2 and np.abs(a["color"] - b["color"]).max() < 2e-2
    assert np.abs(a["distance"] - b["distance"]).max() > 1e-5          # the mode is actually on
    c = orc.NeDDFOracle(bunny_weights, **BUNNY_CFG).forward(pos, d, var)
    assert all(np.array_equal(a[k], c[k]) for k in a)                  # ... and off again afterwards


def _negbias():
    g = golden("neddf_negbias.npz")
    kw = json.loads(str(g["config"]))
    sd = synth.neddf_state_negbias(kw["embed_pos_rank"], kw["embed_dir_rank"], kw["ddf_layer_count"], kw["ddf_layer_width"],
                                   kw["col_layer_count"], kw["col_layer_width"], tuple(kw["skips"]), seed=7)
    return g, kw, sd


def test_neddf_negative_bias_regime():
    """The fixture that enters the regime where the fused kernels' reduced-cost tanhExp / sincos / exp differ from the
    reference's arithmetic (tests/golden/gen_goldens.py::gen_negbias): 80 % of the pre-activations below -1 (median -6),
    D down to 0.014, |pos| up to 6 under a rank-10 encoding.  The oracle -- branch-exact -- must hold the reference's fp32
    outputs at the north-star gate here too, and the fixture must really be in that regime."""
    g, kw, sd = _negbias()
    assert float(g["preact_frac_below_m1"]) > 0.75 and float(g["preact_quantiles"][3]) < -5.0
    assert float(np.abs(g["pos"]).max()) > 5.9 and float(g["eval_distance"].min()) < 0.02
    net = orc.NeDDFOracle(sd, **kw)
    for it, tag in ((-1, "eval"), (2500, "it2500")):
        net.set_iter(it)
        o = net.forward(g["pos"], g["dir"], g["var"])
        exact = g["%s_density_fp64" % tag]
        e_ref = float(np.abs(g["%s_density" % tag].astype(np.float64) - exact).max())
        for k in ("distance", "aux_grad", "color", "density", "fields_penalty"):
            assert_close(o[k], g["%s_%s" % (tag, k)], 1e-4, 1e-5, "negbias %s %s" % (tag, k))
        assert float(np.abs(o["density"].astype(np.float64) - exact).max()) <= 2.5 * e_ref + 1e-7


def test_neddf_negative_bias_render_rays():
    g, kw, sd = _negbias()
    r = golden("neddf_negbias_render_rays.npz")
    net = orc.NeDDFOracle(sd, **kw)
    net.set_iter(-1)
    o = orc.render_rays(net, net, r["uv"], r["R"], r["T"], r["calib"], r["u_coarse"], r["u_fine"], 2.0, 6.0, 6.0, "cone")
    for k in ("color", "depth", "transmittance", "color_coarse", "depth_coarse", "weight_coarse", "fields_penalty"):
        assert_close(o[k], r["out_" + k], 1e-4, 1e-5, "negbias render_rays " + k)


@pytest.mark.parametrize("block", [0, 1, 2, 3])
def test_oracle_on_random_architectures(block):
    """Round 4: the oracle against the REFERENCE on the sixty randomly drawn architectures of the rendering sweep (synth.random_arch, seeds
    0 .. 59: field kind, hidden width 8 .. 512, 2 .. 7 layers, up to three skips, any activation, encoding ranks 1 .. 10; goldens:
    tests/golden/gen_goldens.py fields_random, the reference's forward in evaluation mode).  This is what pins the checker of
    tests/test_gpu_parity.py::test_random_architectures_vs_oracle on those architectures, fifteen per block."""
    g = golden("fields_random.npz")
    for seed in range(15 * block, 15 * block + 15):
        pre = "s%d_" % seed
        cfg = json.loads(str(g[pre + "config"]))
        kind, kw = cfg["kind"], cfg["kw"]
        assert (kind, kw) == synth.random_arch(seed)
        rng = np.random.default_rng(1000 + seed)
        rays, samples = int(rng.integers(1, 9)), int(rng.integers(1, 50))
        pos, d, var = synth.random_sampling(rays, samples, seed=2000 + seed)
        sd = synth.arch_state(kind, kw, 3000 + seed)
        if kind == "neddf":
            o = orc.NeDDFOracle(sd, **kw).forward(pos, d, var)
        elif kind == "nerf":
            o = orc.NeRFOracle(sd, **kw).forward(pos, d, var)
        else:
            o = orc.NeuSOracle(sd, **kw).forward(pos, d)
        keys = [k[len(pre) + 4:] for k in g.files if k.startswith(pre + "out_")]
        assert keys
        for k in keys:
            assert_close(o[k], g[pre + "out_" + k], 1e-4, 2e-5, "seed %d %s %s %s" % (seed, kind, json.dumps(kw), k))


def test_oracle_render_rays_on_random_configurations():
    """Round 4: the oracle's `render_rays` against the REFERENCE's on the random configurations of the rendering sweep
    (synth.random_render_config; goldens: gen_goldens.py render_random -- the twenty of twenty-four the reference can run)."""
    g = golden("render_random.npz")
    checked = 0
    for seed in range(24):
        pre = "s%d_" % seed
        if pre + "reference_error" in g.files:
            continue
        c = synth.random_render_config(seed)
        kind, kw = c["kind"], c["kw"]
        oc = {"neddf": orc.NeDDFOracle, "nerf": orc.NeRFOracle, "neus": orc.NeuSOracle}[kind]
        sd_f = synth.arch_state(kind, kw, 500 + seed)
        sd_c = synth.arch_state(kind, kw, 600 + seed) if c["two"] else sd_f
        o = orc.render_rays(oc(sd_c, **kw), oc(sd_f, **kw), c["uv"], g[pre + "R"], g[pre + "T"], c["calib"].astype(np.float32), g[pre + "u_coarse"],
                            g[pre + "u_fine"], c["near"], c["far"], c["max_dist"], "cone" if c["cone"] else "point")
        keys = [k[len(pre) + 4:] for k in g.files if k.startswith(pre + "out_")]
        assert len(keys) >= 8
        for k in keys:
            assert_close(o[k], g[pre + "out_" + k], 1e-4, 1e-5, "seed %d %s %s" % (seed, kind, k))
        checked += 1
    assert checked == 20


def test_oracle_stages_on_random_inputs():
    """Round 4: `sample_pdf` and `integrate_volume_render` of the oracle against the REFERENCE on sixteen random shapes with hostile
    inputs (gen_goldens.py stages_random: 3 .. 130 knots, 1 .. 200 samples, with / without the coarse knots; zero, negative, NaN,
    denormal-small weights, repeated knots; 2 .. 300 compositing samples with negative and saturating densities).  Samples and the
    sanitised weights bit for bit; compositing at the gates of the fixed-shape tests."""
    g = golden("stages_random.npz")
    for seed in range(16):
        pre = "sp%d_" % seed
        w = g[pre + "w"].copy()
        out, ids, fb = orc.sample_pdf(g[pre + "dists"], w, g[pre + "u"], bool(g[pre + "cat"]))
        assert np.array_equal(out, g[pre + "out"], equal_nan=True), seed
        assert np.array_equal(w, g[pre + "wafter"], equal_nan=True), seed
        pre = "iv%d_" % seed
        o = orc.integrate(g[pre + "dists"], g[pre + "dens"], g[pre + "col"], float(g["max_dist"]))
        assert_close(o["weight"], g[pre + "weight"], 2e-5, 3e-7, "seed %d weight" % seed)
        assert_close(o["color"], g[pre + "color"], 2e-5, 2e-6, "seed %d color" % seed)
        assert_close(o["depth"], g[pre + "depth"], 2e-5, 2e-6, "seed %d depth" % seed)
        assert_close(o["transmittance"], g[pre + "trans"], 2e-5, 1e-9, "seed %d transmittance" % seed)


def test_oracle_rays_on_random_cameras():
    """Round 4: `create_rays` and the cone / point sampling of the oracle against the REFERENCE for twelve random pinhole cameras, poses,
    pixel dtypes (int64 / int16 / int32 / float) and distance sets with random cone radii (gen_goldens.py rays_random), at the gates
    of the fixed-shape tests above."""
    g = golden("rays_random.npz")
    for seed in range(12):
        pre = "s%d_" % seed
        rd, ro = orc.create_rays(g[pre + "uv"], g[pre + "R"], g[pre + "T"], g[pre + "calib"])
        assert_close(rd, g[pre + "ray_dir"], 1e-6, 1e-7, "seed %d ray_dir" % seed)
        assert np.array_equal(ro, g[pre + "ray_orig"]), seed
        pos, d, var = orc.sampling(g[pre + "ray_dir"], g[pre + "ray_orig"], g[pre + "dists"], float(g[pre + "radius"]))
        assert_close(pos, g[pre + "cone_pos"], 1e-6, 1e-7, "seed %d cone pos" % seed)
        assert_close(var, g[pre + "cone_var"], 1e-4, 1e-12, "seed %d cone var" % seed)
        assert np.array_equal(d, np.broadcast_to(g[pre + "ray_dir"][:, None, :], d.shape)), seed
        pos, d, var = orc.sampling(g[pre + "ray_dir"], g[pre + "ray_orig"], g[pre + "dists"], None)
        assert_close(pos, g[pre + "point_pos"], 1e-6, 1e-7, "seed %d point pos" % seed)
        assert float(np.abs(var).max()) == 0.0


def _uint8_gate(got, want, what):
    """<= 1 count on <= 0.5 % of the values: a 1e-4 difference on the float side of `astype(uint8)` can move a value across an
    integer boundary; a wrong constant, channel order or depth scale moves (nearly) all of them."""
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert got.shape == want.shape and got.dtype == want.dtype == np.uint8, (what, got.shape, want.shape)
    assert d.max() <= 1 and np.mean(d > 0) <= 0.005, "%s: max |diff| %d counts, %.3f %% of values differ" % (what, d.max(), 100 * np.mean(d > 0))


def test_eval_harness_outputs_vs_reference(bunny_weights):
    """SURVEY 8 A1 / 8f item 1 against the REFERENCE'S OWN harness: tests/golden/eval_harness.npz holds the uint8 arrays the
    reference's `NeRFTrainer.render_test` (base_trainer.py:123-174) handed to `cv2.imwrite` for tests/golden/bunny_mini's two test
    views (72 x 56, chunk 1024: a short last chunk), and the PSNR it printed.  Here: the oracle renders view 0 with the uniforms in
    the reference's chunk order and the conversion constants are applied -- colour `clamp(c * 255)`, depth
    `clamp((d - 2) / 4 * 50000 / 256)`, B,G,R order, ground truth = premultiplied colour -- and the PSNR of the stored arrays is
    recomputed by the product's `metrics.peak_signal_noise_ratio`."""
    import torch
    from neddf_amd.dataset import NeRFSyntheticDataset
    from neddf_amd.metrics import peak_signal_noise_ratio
    from scipy.spatial.transform import Rotation
    import os
    from conftest import GOLDEN
    g = golden("eval_harness.npz")
    ds = NeRFSyntheticDataset(os.path.join(GOLDEN, "bunny_mini"), "test", use_mask=True)
    for cam in (0, 1):
        assert np.array_equal(ds[cam]["rgb_images"].astype(np.uint8), g["cam%d_rgb_gt" % cam])
        assert abs(peak_signal_noise_ratio(g["cam%d_rgb" % cam], g["cam%d_rgb_gt" % cam]) - float(g["cam%d_psnr" % cam])) < 1e-9
        assert str(g["cam%d_printout" % cam]).startswith("psnr: %s, ssim: " % repr(float(g["cam%d_psnr" % cam])))
    h, w = g["cam0_rgb"].shape[:2]
    assert (w, h) == (72, 56) and g["cam0_depth"].shape == (h, w, 1)
    item = ds[0]
    p = np.asarray(item["camera_params"], np.float64)
    R = Rotation.from_rotvec(p[:3]).as_matrix().astype(np.float32)
    Tr = p[3:6].astype(np.float32)
    calib = np.asarray(item["camera_calib_params"], np.float32)
    us, vs = np.meshgrid(np.arange(w), np.arange(h))
    uv = np.stack([us.reshape(-1), vs.reshape(-1)], 1).astype(np.int64)          # nerf_render.py:222-231: idx = v * w + u
    net = orc.NeDDFOracle(bunny_weights, **BUNNY_CFG)
    net.set_iter(-1)
    chunk = int(g["chunk"])
    torch.manual_seed(int(g["seed"]))
    col, dep = [], []
    for lo in range(0, uv.shape[0], chunk):
        n = min(chunk, uv.shape[0] - lo)
        uc, uf = torch.rand(n, 65).numpy(), torch.rand(n, 129).numpy()           # nerf_render.py:137, base_neural_render.py:75
        o = orc.render_rays(net, net, uv[lo:lo + n], R, Tr, calib, uc, uf, 2.0, 6.0, 6.0, "cone")
        col.append(o["color"]); dep.append(o["depth"])
    col = np.concatenate(col).reshape(h, w, 3)
    dep = np.concatenate(dep).reshape(h, w, 1)
    f32 = np.float32
    rgb = np.clip(col * f32(255), 0, 255).astype(np.uint8)
    dpt = np.clip((dep - f32(2.0)) / f32(4.0) * f32(50000) / f32(256), 0, 255).astype(np.uint8)
    _uint8_gate(rgb, g["cam0_rgb"], "colour image")
    _uint8_gate(dpt, g["cam0_depth"], "depth image")
    assert len(np.unique(g["cam0_rgb"])) > 50 and len(np.unique(g["cam0_depth"])) > 20      # the frame is not flat
    assert abs(peak_signal_noise_ratio(rgb, g["cam0_rgb_gt"]) - float(g["cam0_psnr"])) < 0.01
