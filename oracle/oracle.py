"""ctypes front-end of the CPU oracle (oracle/neddf_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of neddf_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package (neddf_amd) never does.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
ACT = {"ReLU": 0, "LeakyReLU": 1, "tanhExp": 2}
PENALTY_KEYS = ("constraints_aux_grad", "constraints_dDdt", "range_distance", "range_aux_grad",
                "range_color", "constraints_color")   # dict insertion order, neddf.py:260-291
MAXL = 16

_fp = C.POINTER(C.c_float)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libneddf_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libneddf_oracle.so")
        src = os.path.join(_HERE, "neddf_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            build()
        _LIB = C.CDLL(path)
        assert _LIB.orc_struct_sizes(0) == C.sizeof(_NeDDF), "struct layout mismatch"
        assert _LIB.orc_struct_sizes(1) == C.sizeof(_NeRF), "struct layout mismatch"
        assert _LIB.orc_struct_sizes(2) == C.sizeof(_NeuS), "struct layout mismatch"
    return _LIB


class _NeDDF(C.Structure):
    _fields_ = [("embed_pos_rank", C.c_int), ("embed_dir_rank", C.c_int), ("n_ddf", C.c_int),
                ("ddf_width", C.c_int), ("n_col", C.c_int), ("col_width", C.c_int), ("n_skips", C.c_int),
                ("skips", C.c_int * 8), ("activation", C.c_int), ("density_activation", C.c_int),
                ("d_near", C.c_float), ("aux_grad_scale", C.c_float), ("distance_range_max", C.c_float),
                ("penalty_weight", C.c_float * 6), ("penalty_has", C.c_int * 6), ("lowpass", _fp),
                ("ddf_w", _fp * MAXL), ("ddf_b", _fp * MAXL), ("col_w", _fp * MAXL), ("col_b", _fp * MAXL),
                ("ddf_out_w", _fp), ("ddf_out_b", _fp), ("aux_out_w", _fp), ("aux_out_b", _fp),
                ("col_out_w", _fp), ("col_out_b", _fp)]


class _NeRF(C.Structure):
    _fields_ = [("embed_pos_rank", C.c_int), ("embed_dir_rank", C.c_int), ("n_layers", C.c_int),
                ("width", C.c_int), ("n_skips", C.c_int), ("skips", C.c_int * 8), ("activation", C.c_int),
                ("density_activation", C.c_int), ("lowpass", _fp), ("w", _fp * MAXL), ("b", _fp * MAXL),
                ("dens_w", _fp), ("dens_b", _fp), ("col0_w", _fp), ("col0_b", _fp), ("col1_w", _fp),
                ("col1_b", _fp)]


class _NeuS(C.Structure):
    _fields_ = [("embed_pos_rank", C.c_int), ("embed_dir_rank", C.c_int), ("n_sdf", C.c_int), ("width", C.c_int),
                ("col_width", C.c_int), ("n_col", C.c_int), ("n_skips", C.c_int), ("skips", C.c_int * 8), ("activation", C.c_int),
                ("variance", C.c_float), ("sdf_w", _fp * MAXL), ("sdf_b", _fp * MAXL), ("col_w", _fp * MAXL),
                ("col_b", _fp * MAXL)]


def _f32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(_fp)


def lowpass_scale(alpha, embed_dim):
    """get_lowpass_scale, with_grad/positional_encoding.py:137-157 (per frequency)."""
    s = np.ones(embed_dim, np.float32)
    if alpha >= embed_dim:
        return s
    k = int(alpha)
    s[k] = 0.5 * (1 - math.cos(math.pi * (alpha - k))) + 1e-7
    s[k + 1:] = 1e-7
    return s


# ---------------------------------------------------------------------- stages
def create_rays(uv, R, T, calib):
    uv = _f32(uv); R = _f32(R); T = _f32(T); calib = _f32(calib)
    B = uv.shape[0]
    d = np.empty((B, 3), np.float32); o = np.empty((B, 3), np.float32)
    lib().orc_create_rays(_p(uv), B, _p(R), _p(T), _p(calib), _p(d), _p(o))
    return d, o


def sample_coarse(U, near, far):
    U = _f32(U)
    out = np.empty_like(U)
    lib().orc_sample_coarse(_p(U), U.shape[0], U.shape[1], C.c_float(near), C.c_float(far), _p(out))
    return out


def sampling(ray_dir, ray_orig, dists, ray_radius=None):
    """cone sampling if ray_radius is given, else point sampling."""
    ray_dir = _f32(ray_dir); ray_orig = _f32(ray_orig); dists = _f32(dists)
    B, S = dists.shape
    pos = np.empty((B, S, 3), np.float32); d = np.empty_like(pos); var = np.empty_like(pos)
    if ray_radius is None:
        lib().orc_sampling_points(_p(ray_dir), _p(ray_orig), _p(dists), B, S, _p(pos), _p(d), _p(var))
    else:
        lib().orc_sampling_cones(_p(ray_dir), _p(ray_orig), _p(dists), B, S, C.c_double(ray_radius),
                                 _p(pos), _p(d), _p(var))
    return pos, d, var


def integrate(dists, dens, col, max_dist):
    dists = _f32(dists); dens = _f32(dens); col = _f32(col)
    B, S = dists.shape
    w = np.empty((B, S - 1), np.float32); depth = np.empty(B, np.float32)
    color = np.empty((B, 3), np.float32); trans = np.empty(B, np.float32)
    nan = lib().orc_integrate(_p(dists), _p(dens), _p(col), B, S, C.c_float(max_dist), _p(w), _p(depth),
                              _p(color), _p(trans))
    return dict(weight=w, depth=depth, color=color, transmittance=trans, nan=bool(nan))


def integrate_penalty(dists, pen):
    dists = _f32(dists); pen = _f32(pen)
    out = np.empty(dists.shape[0], np.float32)
    lib().orc_integrate_penalty(_p(dists), _p(pen), dists.shape[0], dists.shape[1], _p(out))
    return out


def sample_pdf(dists, weights, U, cat_coarse=True):
    """weights is mutated in place (as the reference does).  Returns (samples, ids, fallback)."""
    dists = _f32(dists); U = _f32(U)
    assert weights.dtype == np.float32 and weights.flags.c_contiguous
    B, n = dists.shape
    nf = U.shape[1]
    no = nf + n if cat_coarse else nf
    out = np.empty((B, no), np.float32)
    ids = np.empty((B, nf), np.int64)
    fb = lib().orc_sample_pdf(_p(dists), _p(weights), _p(U), B, n, nf, int(cat_coarse), _p(out),
                              ids.ctypes.data_as(C.POINTER(C.c_int64)))
    return out, ids, bool(fb)


# ------------------------------------------------------------------------- ops
def activation_grad(kind, x, J):
    x = _f32(x); J = _f32(J)
    N, Cc = x.shape
    y = np.empty_like(x); G = np.empty_like(J)
    if kind == "softplus":
        lib().orc_softplus_grad_op(_p(x), _p(J), N, Cc, _p(y), _p(G))
    elif kind == "sigmoid":
        assert Cc == 1
        lib().orc_sigmoid_grad_op(_p(x), _p(J), N, _p(y), _p(G))
    else:
        lib().orc_activation_grad(ACT[kind], _p(x), _p(J), N, Cc, _p(y), _p(G))
    return y, G


def linear_grad(x, J, W, b):
    x = _f32(x); J = _f32(J); W = _f32(W); b = _f32(b)
    N, Cin = x.shape
    Cout = W.shape[1]
    y = np.empty((N, Cout), np.float32); G = np.empty((N, 3, Cout), np.float32)
    lib().orc_linear_grad(_p(x), _p(J), _p(W), _p(b), N, Cin, Cout, _p(y), _p(G))
    return y, G


def pe_weights(var, E):
    var = _f32(var).reshape(-1, 3)
    w = np.empty((var.shape[0], 3 * E), np.float32)
    lib().orc_pe_weights(_p(var), var.shape[0], E, _p(w))
    return w


def pe_grad(x, J, scale, E):
    x = _f32(x); J = _f32(J)
    N = x.shape[0]
    y = np.empty((N, 6 * E), np.float32); G = np.empty((N, 3, 6 * E), np.float32)
    sc = None
    if scale is not None:
        sc = _f32(np.broadcast_to(scale, (N, 3 * E)))
    lib().orc_pe_grad(_p(x), _p(J), _p(sc) if sc is not None else None, N, E, _p(y), _p(G))
    return y, G


def pe(x, scale, E):
    x = _f32(x)
    N = x.shape[0]
    y = np.empty((N, 6 * E), np.float32)
    sc = None
    if scale is not None:
        sc = _f32(np.broadcast_to(scale, (N, 3 * E)))
    lib().orc_pe(_p(x), _p(sc) if sc is not None else None, N, E, _p(y))
    return y


# ---------------------------------------------------------------------- fields
def bf16_round(a):
    """fp32 -> nearest-even bfloat16 -> fp32 (numpy), the rounding of v_cvt_pk_bf16_f32."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(a))


class NeDDFOracle:
    """Mirrors NeDDF(...) ctor keywords (neddf.py:52-66) + a numpy state dict."""

    def __init__(self, state, embed_pos_rank=10, embed_dir_rank=4, ddf_layer_count=8, ddf_layer_width=256,
                 col_layer_count=8, col_layer_width=256, activation_type="tanhExp",
                 density_activation_type="ReLU", d_near=0.01, lowpass_alpha_offset=10.0, skips=None,
                 penalty_weight=None, bf16=False):
        """bf16=True emulates the bf16-operand kernels (BASELINE.json configs[4], not a reference code path): the
        weights of the 256-wide layers and every matrix-unit A operand are rounded to bfloat16, arithmetic stays fp32."""
        self.cfg = dict(E=embed_pos_rank, Ed=embed_dir_rank)
        self.bf16 = bool(bf16)
        self.lowpass_alpha_offset = lowpass_alpha_offset
        if skips is None:
            skips = [4]
        if penalty_weight is None:      # neddf.py:152-159 defaults
            penalty_weight = {"constraints_aux_grad": 0.05, "constraints_dDdt": 0.05,
                              "constraints_color": 0.01, "range_distance": 1.0, "range_aux_grad": 1.0}
        self._keep = {k: _f32(v) for k, v in state.items()}
        if self.bf16:
            for k in list(self._keep):
                if k.startswith(("layers_ddf.", "layers_col.")) and k.endswith(".weight"):
                    self._keep[k] = bf16_round(self._keep[k])
        s = _NeDDF()
        s.embed_pos_rank, s.embed_dir_rank = embed_pos_rank, embed_dir_rank
        s.n_ddf, s.ddf_width = ddf_layer_count - 1, ddf_layer_width
        s.n_col, s.col_width = col_layer_count - 1, col_layer_width
        s.n_skips = len(skips)
        for i, k in enumerate(skips):
            s.skips[i] = k
        s.activation, s.density_activation = ACT[activation_type], ACT[density_activation_type]
        s.d_near = d_near
        for i, k in enumerate(PENALTY_KEYS):
            s.penalty_has[i] = int(k in penalty_weight)
            s.penalty_weight[i] = penalty_weight.get(k, 1.0)
        for i in range(s.n_ddf):
            s.ddf_w[i] = _p(self._keep["layers_ddf.%d.weight" % i]); s.ddf_b[i] = _p(self._keep["layers_ddf.%d.bias" % i])
        for i in range(s.n_col):
            s.col_w[i] = _p(self._keep["layers_col.%d.weight" % i]); s.col_b[i] = _p(self._keep["layers_col.%d.bias" % i])
        s.ddf_out_w, s.ddf_out_b = _p(self._keep["layer_ddf_out.weight"]), _p(self._keep["layer_ddf_out.bias"])
        s.aux_out_w, s.aux_out_b = _p(self._keep["layer_aux_out.weight"]), _p(self._keep["layer_aux_out.bias"])
        s.col_out_w, s.col_out_b = _p(self._keep["layer_col_out.weight"]), _p(self._keep["layer_col_out.bias"])
        self.s = s
        self.set_iter(-1)

    def set_iter(self, it):     # neddf.py:311-326
        E = self.cfg["E"]
        if it == -1:
            self.s.aux_grad_scale, self.s.distance_range_max, alpha = 1.1, 2.0, E
        else:
            self.s.aux_grad_scale = min(1.1, max(0.01, 0.0001 * it))
            self.s.distance_range_max = min(2.0, 2.0 + 0.0001 * it)
            alpha = self.lowpass_alpha_offset + 0.001 * it
        self._lp = lowpass_scale(alpha, E)
        self.s.lowpass = _p(self._lp)

    def forward(self, pos, dir, var):
        pos = _f32(pos); dir = _f32(dir); var = _f32(var)
        shp = pos.shape[:-1]
        N = int(np.prod(shp))
        o = dict(distance=np.empty(N, np.float32), density=np.empty(N, np.float32),
                 color=np.empty((N, 3), np.float32), fields_penalty=np.empty(N, np.float32),
                 aux_grad=np.empty(N, np.float32))
        lib().orc_set_bf16(int(self.bf16))
        try:
            lib().orc_neddf_forward(C.byref(self.s), _p(pos), _p(dir), _p(var), N, _p(o["distance"]),
                                    _p(o["density"]), _p(o["color"]), _p(o["fields_penalty"]), _p(o["aux_grad"]))
        finally:
            lib().orc_set_bf16(0)
        return {k: v.reshape(shp + ((3,) if k == "color" else ())) for k, v in o.items()}

    def forward_fast(self, pos, dir, var):
        """The same field values (no fields_penalty) by bench.py's CPU BASELINE implementation (oracle/neddf_cpu_fast.c):
        eval-minimal like the HIP path -- value rows forward, reverse-mode distance gradient, colour trunk on value rows -- on
        blocked GEMM micro-kernels.  Not the parity checker (summation order differs); held to the 1e-4 gates against forward()."""
        assert not self.bf16
        pos = _f32(pos); dir = _f32(dir); var = _f32(var)
        shp = pos.shape[:-1]
        N = int(np.prod(shp))
        o = dict(distance=np.empty(N, np.float32), density=np.empty(N, np.float32), color=np.empty((N, 3), np.float32),
                 aux_grad=np.empty(N, np.float32))
        lib().fast_neddf_forward(C.byref(self.s), _p(pos), _p(dir), _p(var), N, _p(o["distance"]), _p(o["density"]), _p(o["color"]),
                                 _p(o["aux_grad"]))
        return {k: v.reshape(shp + ((3,) if k == "color" else ())) for k, v in o.items()}


class NeRFOracle:
    """Mirrors NeRF(...) ctor keywords (nerf.py:34-44) + a numpy state dict."""

    def __init__(self, state, embed_pos_rank=10, embed_dir_rank=4, layer_count=8, layer_width=256,
                 activation_type="ReLU", density_activation_type="ReLU", skips=None, lowpass_alpha_offset=10.0):
        if skips is None:
            skips = [4]
        self.E = embed_pos_rank
        self.lowpass_alpha_offset = lowpass_alpha_offset
        self._keep = {k: _f32(v) for k, v in state.items()}
        s = _NeRF()
        s.embed_pos_rank, s.embed_dir_rank, s.n_layers, s.width = embed_pos_rank, embed_dir_rank, layer_count, layer_width
        s.n_skips = len(skips)
        for i, k in enumerate(skips):
            s.skips[i] = k
        s.activation, s.density_activation = ACT[activation_type], ACT[density_activation_type]
        for i in range(layer_count):
            s.w[i] = _p(self._keep["layers.%d.weight" % i]); s.b[i] = _p(self._keep["layers.%d.bias" % i])
        s.dens_w, s.dens_b = _p(self._keep["outL_density.weight"]), _p(self._keep["outL_density.bias"])
        s.col0_w, s.col0_b = _p(self._keep["outL_color.0.weight"]), _p(self._keep["outL_color.0.bias"])
        s.col1_w, s.col1_b = _p(self._keep["outL_color.2.weight"]), _p(self._keep["outL_color.2.bias"])
        self.s = s
        self.set_iter(-1)

    def set_iter(self, it):     # nerf.py:167-178
        alpha = self.E if it == -1 else self.lowpass_alpha_offset + 0.001 * it
        self._lp = lowpass_scale(alpha, self.E)
        self.s.lowpass = _p(self._lp)

    def forward(self, pos, dir, var):
        pos = _f32(pos); dir = _f32(dir); var = _f32(var)
        shp = pos.shape[:-1]
        N = int(np.prod(shp))
        dens = np.empty(N, np.float32); col = np.empty((N, 3), np.float32)
        lib().orc_nerf_forward(C.byref(self.s), _p(pos), _p(dir), _p(var), N, _p(dens), _p(col))
        return dict(density=dens.reshape(shp), color=col.reshape(shp + (3,)))


class NeuSOracle:
    """Mirrors NeuS(...) ctor keywords (neus.py:30-41) + a numpy state dict."""

    def __init__(self, state, embed_pos_rank=6, embed_dir_rank=4, sdf_layer_count=8, sdf_layer_width=256,
                 col_layer_count=8, col_layer_width=256, activation_type="ReLU", init_variance=0.3, skips=None):
        if skips is None:
            skips = [4]
        self._keep = {k: _f32(v) for k, v in state.items()}
        s = _NeuS()
        s.embed_pos_rank, s.embed_dir_rank, s.n_sdf, s.width, s.n_col = embed_pos_rank, embed_dir_rank, sdf_layer_count, sdf_layer_width, col_layer_count
        s.col_width = col_layer_width
        s.n_skips = len(skips)
        for i, k in enumerate(skips):
            s.skips[i] = k
        s.activation = ACT[activation_type]
        s.variance = float(np.asarray(self._keep["variance"]).reshape(-1)[0])
        for i in range(sdf_layer_count):
            s.sdf_w[i] = _p(self._keep["layers_sdf.%d.weight" % i]); s.sdf_b[i] = _p(self._keep["layers_sdf.%d.bias" % i])
        for i in range(col_layer_count + 1):
            s.col_w[i] = _p(self._keep["layers_col.%d.weight" % i]); s.col_b[i] = _p(self._keep["layers_col.%d.bias" % i])
        self.s = s

    def set_iter(self, it):
        pass

    def forward(self, pos, dir, var=None):
        pos = _f32(pos); dir = _f32(dir)
        shp = pos.shape[:-1]
        N = int(np.prod(shp))
        sdf = np.empty(N, np.float32); dens = np.empty(N, np.float32); col = np.empty((N, 3), np.float32)
        lib().orc_neus_forward(C.byref(self.s), _p(pos), _p(dir), N, _p(sdf), _p(dens), _p(col))
        return dict(sdf=sdf.reshape(shp), density=dens.reshape(shp), color=col.reshape(shp + (3,)))


def rays_to_ndc(ray_dir, ray_orig, width, height, fx, fy, near):
    """World rays -> NDC rays (not a reference function; see orc_rays_to_ndc)."""
    ray_dir, ray_orig = _f32(ray_dir), _f32(ray_orig)
    nd, no = np.empty_like(ray_dir), np.empty_like(ray_orig)
    lib().orc_rays_to_ndc(_p(ray_dir), _p(ray_orig), ray_dir.shape[0], int(width), int(height), C.c_float(fx), C.c_float(fy),
                          C.c_float(near), _p(nd), _p(no))
    return nd, no


def render_rays(field_coarse, field_fine, uv, R, T, calib, u_coarse, u_fine, dist_near, dist_far, max_dist,
                sampling_type="cone", ndc=None):
    """NeRFRender.render_rays nerf_render.py:109-188 with the uniforms given explicitly.  ndc = (width, height, near)
    samples along NDC rays while the fields keep the world-space viewing direction (not a reference mode)."""
    ray_radius = 1.0 / 1111 / math.sqrt(12) if sampling_type == "cone" else None
    rd, ro = create_rays(uv, R, T, calib)
    if ndc is not None:
        view = rd
        rd, ro = rays_to_ndc(rd, ro, ndc[0], ndc[1], calib[0], calib[1], ndc[2])

        def sample_fn(rd_, ro_, d_, rr_):     # positions along the NDC ray, direction = world-space view direction
            pos, _, var = sampling(rd_, ro_, d_, rr_)
            return pos, np.broadcast_to(view[:, None, :], pos.shape).astype(np.float32).copy(), var
    else:
        sample_fn = sampling
    dc = sample_coarse(u_coarse, dist_near, dist_far)
    vc = field_coarse.forward(*sample_fn(rd, ro, dc, ray_radius))
    ic = integrate(dc, vc["density"], vc["color"], max_dist)
    if "fields_penalty" in vc:
        ic["fields_penalty"] = integrate_penalty(dc, vc["fields_penalty"])
    df, _, _ = sample_pdf(dc, ic["weight"], u_fine, True)
    vf = field_fine.forward(*sample_fn(rd, ro, df, ray_radius))
    out = integrate(df, vf["density"], vf["color"], max_dist)
    if "fields_penalty" in vf:
        out["fields_penalty"] = integrate_penalty(df, vf["fields_penalty"])
    out.pop("nan"); ic.pop("nan")
    for k, v in ic.items():
        out[k + "_coarse"] = v
    out["dists_fine"] = df
    out["dists_coarse"] = dc
    return out
