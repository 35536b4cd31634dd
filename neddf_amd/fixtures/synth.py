"""Deterministic synthetic weights shared by the golden generator and the tests.

The golden generator (gen_goldens.py, runs only where /root/reference exists)
loads these arrays into the *reference* networks and records their outputs;
the tests regenerate the very same arrays with the same numpy bit-generator
and load them into the HIP path / the C oracle.  Only inputs and expected
outputs are stored in the fixtures, never the weights.

numpy's PCG64 + `standard_normal` stream is stable across numpy versions.
"""
from collections import OrderedDict

import numpy as np


def _layer(rng, fan_in, fan_out, transposed):
    """xavier-normal weight, small non-zero bias (exercises the bias path)."""
    std = np.sqrt(2.0 / (fan_in + fan_out))
    w = (rng.standard_normal((fan_in, fan_out)) * std).astype(np.float32)
    b = (rng.standard_normal((fan_out,)) * 0.05).astype(np.float32)
    if transposed:  # nn.Linear layout [out, in]
        w = np.ascontiguousarray(w.T)
    return w, b


def neddf_state(embed_pos_rank=10, embed_dir_rank=4, ddf_layer_count=8,
                ddf_layer_width=256, col_layer_count=4, col_layer_width=256,
                skips=(4,), seed=7):
    """State dict (numpy) with the key names/shapes of the reference NeDDF
    (neddf/network/neddf.py:129-145): LinearGradLayer weights are [in, out]."""
    rng = np.random.default_rng(seed)
    in_ddf = embed_pos_rank * 6
    in_col = (embed_pos_rank + embed_dir_rank) * 6 + 3 + ddf_layer_width
    sd = OrderedDict()
    dims = [(in_ddf, ddf_layer_width)]
    for layer_id in range(ddf_layer_count - 2):
        if layer_id in skips:
            dims.append((ddf_layer_width + in_ddf, ddf_layer_width))
        else:
            dims.append((ddf_layer_width, ddf_layer_width))
    for i, (a, b) in enumerate(dims):
        w, bias = _layer(rng, a, b, False)
        sd["layers_ddf.%d.weight" % i] = w
        sd["layers_ddf.%d.bias" % i] = bias
    dims = [(in_col, col_layer_width)] + [(col_layer_width, col_layer_width)] * (col_layer_count - 2)
    for i, (a, b) in enumerate(dims):
        w, bias = _layer(rng, a, b, False)
        sd["layers_col.%d.weight" % i] = w
        sd["layers_col.%d.bias" % i] = bias
    for name, n in (("layer_ddf_out", 1), ("layer_aux_out", 1), ("layer_col_out", 3)):
        w, bias = _layer(rng, ddf_layer_width, n, False)
        sd[name + ".weight"] = w
        sd[name + ".bias"] = bias
    return sd


def nerf_state(embed_pos_rank=10, embed_dir_rank=4, layer_count=8, layer_width=256,
               skips=(4,), seed=11):
    """State dict (numpy) with the key names/shapes of the reference NeRF
    (neddf/network/nerf.py:88-103): nn.Linear weights are [out, in]."""
    rng = np.random.default_rng(seed)
    in_pos = embed_pos_rank * 6
    in_dir = embed_dir_rank * 6
    sd = OrderedDict()
    dims = [(in_pos, layer_width)]
    for layer_id in range(layer_count - 1):
        if layer_id in skips:
            dims.append((layer_width + in_pos, layer_width))
        else:
            dims.append((layer_width, layer_width))
    for i, (a, b) in enumerate(dims):
        w, bias = _layer(rng, a, b, True)
        sd["layers.%d.weight" % i] = w
        sd["layers.%d.bias" % i] = bias
    w, bias = _layer(rng, layer_width, 1, True)
    sd["outL_density.weight"], sd["outL_density.bias"] = w, bias
    w, bias = _layer(rng, layer_width + in_dir, layer_width // 2, True)
    sd["outL_color.0.weight"], sd["outL_color.0.bias"] = w, bias
    w, bias = _layer(rng, layer_width // 2, 3, True)
    sd["outL_color.2.weight"], sd["outL_color.2.bias"] = w, bias
    return sd


def random_sampling(n_rays, n_samples, seed, cone=True):
    """Seeded Sampling-like inputs (pos, unit dir, diag variance)."""
    rng = np.random.default_rng(seed)
    pos = rng.uniform(-1.2, 1.2, (n_rays, n_samples, 3)).astype(np.float32)
    d = rng.standard_normal((n_rays, 1, 3))
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    d = np.broadcast_to(d, (n_rays, n_samples, 3)).astype(np.float32).copy()
    if cone:
        var = (rng.uniform(0.0, 1.0, (n_rays, n_samples, 3)) ** 4 * 2e-3).astype(np.float32)
    else:
        var = np.zeros((n_rays, n_samples, 3), np.float32)
    return pos, d, var


def neus_state(embed_pos_rank=6, embed_dir_rank=4, sdf_layer_count=8, sdf_layer_width=256, col_layer_count=8,
               col_layer_width=256, skips=(4,), init_variance=0.3, seed=13):
    """State dict (numpy) with the key names/shapes of the reference NeuS
    (neddf/network/neus.py:80-99): nn.Linear weights [out, in] + scalar `variance`."""
    rng = np.random.default_rng(seed)
    in_sdf = embed_pos_rank * 6
    in_col = 6 + embed_dir_rank * 6 + sdf_layer_width
    sd = OrderedDict()
    dims = [(in_sdf, sdf_layer_width)]
    for layer_id in range(sdf_layer_count - 1):
        dims.append((sdf_layer_width + (in_sdf if layer_id in skips else 0), sdf_layer_width))
    for i, (a, b) in enumerate(dims):
        sd["layers_sdf.%d.weight" % i], sd["layers_sdf.%d.bias" % i] = _layer(rng, a, b, True)
    dims = [(in_col, col_layer_width)] + [(col_layer_width, col_layer_width)] * (col_layer_count - 1) + [(col_layer_width, 3)]
    for i, (a, b) in enumerate(dims):
        sd["layers_col.%d.weight" % i], sd["layers_col.%d.bias" % i] = _layer(rng, a, b, True)
    sd["variance"] = np.float32(init_variance)
    return sd


def neddf_state_negbias(embed_pos_rank=10, embed_dir_rank=4, ddf_layer_count=8, ddf_layer_width=256, col_layer_count=4,
                        col_layer_width=256, skips=(4,), seed=7, frac=0.8, lo=-15.0, hi=-1.0, wscale=3.0, head_bias=-3.0):
    """`neddf_state` pushed into the regime where a tanhExp evaluated as 1 - 2 / (e^(2 e^x) + 1) loses RELATIVE accuracy:
    a fraction `frac` of the hidden units of both trunks get a bias drawn from [lo, hi], so their pre-activations sit at
    e^x << 0.3; the hidden weights are scaled by `wscale` so that the remaining units still carry a position-dependent
    signal, and the distance head's bias makes D = softplus(z) + d_near small (the 1/D of the density, neddf.py:239)."""
    sd = neddf_state(embed_pos_rank, embed_dir_rank, ddf_layer_count, ddf_layer_width, col_layer_count, col_layer_width, skips, seed)
    rng = np.random.default_rng(seed + 1000)
    for k in list(sd):
        if not k.startswith(("layers_ddf.", "layers_col.")):
            continue
        if k.endswith(".weight"):
            sd[k] = (sd[k] * np.float32(wscale)).astype(np.float32)
        else:
            b = sd[k].copy()
            m = rng.uniform(size=b.shape) < frac
            b[m] = rng.uniform(lo, hi, size=int(m.sum())).astype(np.float32)
            sd[k] = b
    sd["layer_ddf_out.bias"] = np.full_like(sd["layer_ddf_out.bias"], head_bias)
    return sd


def wide_sampling(n_rays, n_samples, seed, reach=6.0):
    """`random_sampling` with half of the rays reaching out to |pos| <= reach (2^9 * 6 rad at the highest frequency of a
    rank-10 encoding) and half of the rays at zero variance (point sampling: the high frequencies keep their weight)."""
    pos, d, var = random_sampling(n_rays, n_samples, seed, cone=True)
    rng = np.random.default_rng(seed + 77)
    far = rng.uniform(-reach, reach, pos.shape).astype(np.float32)
    pos[n_rays // 2:] = far[n_rays // 2:]
    var[::2] = 0.0
    return pos, d, var


def random_arch(seed, train=False):
    """(kind, constructor keywords) of one architecture the reference's constructors accept (neddf.py:52-66, nerf.py:34-44,
    neus.py:30-41), drawn from `seed`: kinds cycle NeDDF / NeRF / NeuS; hidden width 8 .. 512, 2 .. 7 layers, up to three skip
    connections, any activation on trunk and density head, encoding ranks 1 .. 10.  (`train` is kept for the callers' readability:
    every draw trains since round 4.)"""
    rng = np.random.default_rng(seed)
    kind = ("neddf", "nerf", "neus")[seed % 3]
    widths = [8, 24, 40, 64, 72, 100, 128, 160, 200, 256, 264, 320, 384, 448, 512]
    width = int(rng.choice(widths))
    n = int(rng.integers(2, 8))
    # a skip index names a hidden layer that is followed by another: NeRF / NeuS have n of them, NeDDF n - 1 (its `ddf_layer_count`
    # counts the output layer, neddf.py:128-136) -- the reference's own forward fails beyond that
    n_hidden = n - 1 if kind == "neddf" else n
    skips = (sorted(int(x) for x in rng.choice(np.arange(0, n_hidden - 1), size=int(rng.integers(0, min(3, n_hidden - 1) + 1)), replace=False))
             if n_hidden > 1 else [])
    act = str(rng.choice(["ReLU", "LeakyReLU", "tanhExp"]))
    dact = str(rng.choice(["ReLU", "LeakyReLU", "tanhExp"]))
    E, Ed = int(rng.integers(1, 11)), int(rng.integers(1, 11))
    if kind == "neddf":
        return kind, dict(embed_pos_rank=E, embed_dir_rank=Ed, ddf_layer_count=n, ddf_layer_width=width, col_layer_count=int(rng.integers(2, 6)),
                          col_layer_width=width, d_near=0.01, activation_type=act, density_activation_type=dact, skips=skips, lowpass_alpha_offset=10)
    if kind == "nerf":
        return kind, dict(embed_pos_rank=E, embed_dir_rank=Ed, layer_count=n, layer_width=width + (width & 1), activation_type=act,
                          density_activation_type=dact, skips=skips, lowpass_alpha_offset=10)
    wc = int(rng.choice([16, 64, 128, 256, 320, 512]))
    return kind, dict(embed_pos_rank=E, embed_dir_rank=Ed, sdf_layer_count=n, sdf_layer_width=width, col_layer_count=int(rng.integers(1, 5)),
                      col_layer_width=wc, init_variance=float(rng.uniform(0.1, 0.6)), activation_type=str(rng.choice(["ReLU", "tanhExp"])), skips=skips)


def arch_state(kind, kw, seed):
    """The seeded state dict of random_arch's (kind, kw)."""
    if kind == "neddf":
        return neddf_state(kw["embed_pos_rank"], kw["embed_dir_rank"], kw["ddf_layer_count"], kw["ddf_layer_width"], kw["col_layer_count"],
                           kw["col_layer_width"], tuple(kw["skips"]), seed=seed)
    if kind == "nerf":
        return nerf_state(kw["embed_pos_rank"], kw["embed_dir_rank"], kw["layer_count"], kw["layer_width"], tuple(kw["skips"]), seed=seed)
    return neus_state(kw["embed_pos_rank"], kw["embed_dir_rank"], kw["sdf_layer_count"], kw["sdf_layer_width"], kw["col_layer_count"],
                      kw["col_layer_width"], tuple(kw["skips"]), kw["init_variance"], seed=seed)


def random_render_config(seed):
    """One `render_rays` configuration of the rendering sweep (tests/test_gpu_parity.py test_random_render_configs, goldens:
    gen_goldens.py render_random): field kind and a small architecture, one network or a coarse / fine pair, cone or point sampling,
    sample counts, near / far / max_dist, a pinhole camera with a random pose, integer or float pixel coordinates, a ragged number of rays."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(400 + seed)
    kind, kw = random_arch(900 + seed)
    wkey = {"neddf": "ddf_layer_width", "nerf": "layer_width", "neus": "sdf_layer_width"}[kind]
    kw[wkey] = int(rng.choice([8, 24, 40, 64, 72]))
    if "col_layer_width" in kw:
        kw["col_layer_width"] = kw[wkey] if kind == "neddf" else int(rng.choice([16, 64]))
    # Position encodings up to rank 5 here (the field-level sweep goes to 10): the importance samples of two implementations differ by
    # one ulp of the distance on a few rays (their coarse weights differ in the last bits), and a randomly initialised network under a
    # rank-10 encoding turns 1e-6 of position into 1e-4 of colour -- measured: the fields agree to 7e-6 at identical points while the
    # pixel moves by 9e-3.  That is the conditioning of the random network, not of the renderer.
    kw["embed_pos_rank"] = min(kw["embed_pos_rank"], 5)
    c = dict(kind=kind, kw=kw)
    c["two"] = bool(rng.integers(0, 2))
    c["cone"] = kind != "neus" and bool(rng.integers(0, 2))
    c["n_c"], c["n_f"] = int(rng.integers(1, 81)), int(rng.integers(1, 121))
    c["near"] = float(rng.uniform(0.5, 2.5))
    c["far"] = c["near"] + float(rng.uniform(1.0, 5.0))
    c["max_dist"] = c["far"] + float(rng.choice([0.0, 1.0]))
    n = c["n"] = int(rng.integers(1, 71))
    W, H = int(rng.integers(16, 400)), int(rng.integers(16, 400))
    c["calib"] = np.array([rng.uniform(0.6, 2.0) * W, rng.uniform(0.6, 2.0) * W, 0.5 * W + rng.uniform(-3, 3), 0.5 * H + rng.uniform(-3, 3)])
    uv = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], 1)
    c["uv"] = uv.astype(np.float32) + rng.uniform(0, 1, (n, 2)).astype(np.float32) if rng.integers(0, 2) else uv.astype(np.int64)
    c["rotvec"] = Rotation.random(random_state=int(rng.integers(0, 1 << 30))).as_rotvec().astype(np.float32)
    c["t"] = rng.uniform(-1.0, 1.0, 3).astype(np.float32)
    c["target"] = {"neddf": "neddf.network.NeDDF", "nerf": "neddf.network.NeRF", "neus": "neddf.network.NeuS"}[kind]
    return c


def random_train_render_config(seed):
    """random_render_config restricted to what the REFERENCE can take a training step on: NeDDF over cone samples (over point samples
    its forward dies, neddf.py:210), ReLU density (with negative coarse weights sample_pdf's in-place sanitisation invalidates the
    reference's own autograd graph, base_neural_render.py:52-55), and small sample counts (NeuS is a double backward)."""
    c = random_render_config(200 + seed)
    if c["kind"] == "neddf":
        c["cone"] = True
    if "density_activation_type" in c["kw"]:
        c["kw"]["density_activation_type"] = "ReLU"
    c["n"] = min(c["n"], 24)
    c["uv"] = c["uv"][:c["n"]]
    c["n_c"], c["n_f"] = 2 + c["n_c"] % 38, 1 + c["n_f"] % 60
    return c
