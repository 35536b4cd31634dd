"""BASELINE.json configs[4]: forward-facing NDC rays + bf16 MLP operands on the bf16 MFMA.

The reference has neither (fp32 only, no NDC/LLFF code), so parity here is pinned differently ("parity
unpinned" items): the NDC construction against the CPU restatement of the published formula (itself checked on the
projective identity in tests/test_oracle.py), and the bf16 kernels against the oracle's bf16 emulation, which rounds
weights and matrix-unit inputs to bfloat16 at the same places and keeps fp32 arithmetic elsewhere.  The distance to
the fp32 result is reported and bounded loosely (north_star states 1e-4 for fp32 only)."""
import numpy as np
import pytest
import torch
from conftest import BUNNY_CFG, assert_close, golden

import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _inference():
    with torch.no_grad():
        yield


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ctx(dev):
    from neddf_amd import Context
    return Context.get(dev)


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def _rays(n, seed):
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    d[:, 2] = -np.abs(d[:, 2]) - 0.5
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = rng.uniform(-0.3, 0.3, (n, 3)).astype(np.float32)
    return d, o


def test_rays_to_ndc_bit_exact(ctx, dev, orc):
    d, o = _rays(1000, 1)
    nd, no = ctx.rays_to_ndc(T(d, dev), T(o, dev), 1008, 756, 815.1, 809.3, 1.0)
    rd, ro = orc.rays_to_ndc(d, o, 1008, 756, 815.1, 809.3, 1.0)
    assert np.array_equal(N(nd), rd) and np.array_equal(N(no), ro)       # +,-,*,/ only
    e = ctx.rays_to_ndc(T(d[:0], dev), T(o[:0], dev), 8, 8, 1.0, 1.0, 1.0)
    assert e[0].shape == (0, 3)


def test_sampling_with_view_direction(ctx, dev, orc):
    d, o = _rays(33, 2)
    view, _ = _rays(33, 3)
    dists = np.sort(np.random.default_rng(4).uniform(0, 1, (33, 17)).astype(np.float32), axis=1)
    pos, sd, var = ctx.sampling(T(d, dev), T(o, dev), T(dists, dev), None, T(view, dev))
    rpos, _, rvar = orc.sampling(d, o, dists, None)
    assert np.array_equal(N(pos), rpos) and np.array_equal(N(var), rvar)
    assert np.array_equal(N(sd), np.broadcast_to(view[:, None, :], (33, 17, 3)))


def _net(dev, weights, dtype, mode="full", cfg=BUNNY_CFG):
    import neddf_amd
    net = neddf_amd.NeDDF(**cfg)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    net.to(dev)
    net.set_iter(-1)
    net.weight_dtype = dtype
    net.output_mode = mode
    return net


@pytest.mark.parametrize("mode", ["full", "minimal"])
def test_bf16_field_against_bf16_emulation(dev, orc, bunny_weights, mode):
    """bf16 kernels vs the oracle with the same roundings: what remains is fp32 summation order inside the MFMA plus
    the occasional activation that lands on the other side of a bf16 rounding boundary."""
    from neddf_amd import Sampling
    pos, d, var = synth.random_sampling(40, 33, seed=21)         # 1320 points: ragged last tile
    net = _net(dev, bunny_weights, "bf16", mode)
    o = net(Sampling(T(pos, dev), T(d, dev), T(var, dev)))
    emu = orc.NeDDFOracle(bunny_weights, bf16=True, **BUNNY_CFG).forward(pos, d, var)
    f32 = orc.NeDDFOracle(bunny_weights, **BUNNY_CFG).forward(pos, d, var)
    keys = ("distance", "density", "color", "aux_grad") + (("fields_penalty",) if mode == "full" else ())
    for k in keys:
        a = N(o[k])
        scale = np.abs(f32[k]).max()
        err_emu = np.abs(a - emu[k]).max() / scale
        err_f32 = np.abs(a - f32[k]).max() / scale
        print("%-14s vs bf16 emulation %.2e, vs fp32 %.2e (of max |value| %.3g)" % (k, err_emu, err_f32, scale))
        # rounding to bf16 makes the network discontinuous: a 1e-7 difference in summation order that pushes one
        # activation across a rounding boundary is amplified layer by layer up to the bf16 noise floor, so a deep
        # network is tracked by the emulation only to a fraction of that floor (the shallow test below is the tight one)
        assert err_emu < {"fields_penalty": 5e-2, "color": 2e-2}.get(k, 5e-3), (k, err_emu)
        assert err_f32 < (0.2 if k == "fields_penalty" else 3e-2), (k, err_f32)
    # and the fp32 path is untouched by the dtype switch on the same module
    net.weight_dtype = "fp32"
    o32 = net(Sampling(T(pos, dev), T(d, dev), T(var, dev)))
    assert_close(N(o32["distance"]), f32["distance"], 1e-4, 1e-5, "fp32 after bf16")


def test_bf16_single_layer_tight(dev, orc):
    """One trunk layer and one colour layer: no room for rounding differences to cascade, so the kernel must match the
    emulation closely: what differs is the reduced-cost elementwise math of the bf16 path (hardware sine/cosine with
    |err| ~ 4e-5 on the encodings, tanhExp without the small-argument polynomial), which moves about 1 % of the values
    to the neighbouring bf16, and single activations rounding the other way."""
    from neddf_amd import Sampling
    for act in ("ReLU", "tanhExp"):
        cfg = dict(embed_pos_rank=10, embed_dir_rank=4, ddf_layer_count=2, ddf_layer_width=256, col_layer_count=2,
                   col_layer_width=256, d_near=0.01, activation_type=act, density_activation_type="ReLU",
                   lowpass_alpha_offset=10, skips=[], penalty_weight={"constraints_dDdt": 0.5})
        w = synth.neddf_state(ddf_layer_count=2, col_layer_count=2, skips=(), seed=5)
        pos, d, var = synth.random_sampling(16, 24, seed=9)
        net = _net(dev, w, "bf16", "full", cfg)
        o = net(Sampling(T(pos, dev), T(d, dev), T(var, dev)))
        emu = orc.NeDDFOracle(w, bf16=True, **cfg).forward(pos, d, var)
        f32 = orc.NeDDFOracle(w, **cfg).forward(pos, d, var)
        for k in ("distance", "density", "color", "aux_grad"):
            scale = np.abs(emu[k]).max()
            err = np.abs(N(o[k]) - emu[k]).max() / scale
            gap = np.abs(f32[k] - emu[k]).max() / scale
            print("%s %-9s kernel vs emulation %.2e (emulation vs fp32 %.2e)" % (act, k, err, gap))
            assert err < (4e-3 if k == "density" else 2e-3), (act, k, err)
            assert gap > 2 * err or gap < 1e-6, (act, k, err, gap)      # the test can tell bf16 from fp32


def test_bf16_other_architectures(dev, orc):
    """ReLU activations, two skip connections, small encodings (column padding differs between the operand types)."""
    from neddf_amd import Sampling
    cfg = dict(embed_pos_rank=5, embed_dir_rank=2, ddf_layer_count=6, ddf_layer_width=256, col_layer_count=3, col_layer_width=256,
               d_near=0.01, activation_type="ReLU", density_activation_type="ReLU", lowpass_alpha_offset=10, skips=[1, 3],
               penalty_weight={"constraints_dDdt": 0.5})
    w = synth.neddf_state(embed_pos_rank=5, embed_dir_rank=2, ddf_layer_count=6, col_layer_count=3, skips=(1, 3), seed=77)
    pos, d, var = synth.random_sampling(9, 14, seed=8)
    net = _net(dev, w, "bf16", "full", cfg)
    o = net(Sampling(T(pos, dev), T(d, dev), T(var, dev)))
    emu = orc.NeDDFOracle(w, bf16=True, **cfg).forward(pos, d, var)
    for k in ("distance", "density", "color", "aux_grad"):
        scale = np.abs(emu[k]).max()
        assert np.abs(N(o[k]) - emu[k]).max() / scale < 1e-2, k


def test_bf16_neus_close_to_fp32(dev):
    """NeuS shares the trunk kernels; bf16 is checked against its own fp32 result."""
    import neddf_amd
    from neddf_amd import Sampling
    w = synth.neus_state()
    net = neddf_amd.NeuS()
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()})
    net.to(dev)
    pos, d, var = synth.random_sampling(6, 30, seed=5, cone=False)
    s = Sampling(T(pos, dev), T(d, dev), T(var, dev))
    a = net(s)
    net.weight_dtype = "bf16"
    b = net(s)
    for k in a:
        scale = float(a[k].abs().max())
        # random-weight NeuS colours are small sums of large cancelling terms: loose bound there
        assert float((a[k] - b[k]).abs().max()) / scale < (0.25 if k == "color" else 5e-2), k
        assert not torch.equal(a[k], b[k])


@pytest.mark.parametrize("act,dact", [("ReLU", "ReLU"), ("tanhExp", "LeakyReLU")])
def test_nerf_field_operand_policies(dev, orc, act, dact):
    """The plain NeRF field kernel under the three operand policies: split-bf16 meets the fp32 gate, bf16 stays within 1e-2."""
    import neddf_amd
    from neddf_amd import Sampling
    w = synth.nerf_state(seed=11)
    net = neddf_amd.NeRF(activation_type=act, density_activation_type=dact)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    net.to(dev)
    net.set_iter(-1)
    pos, d, var = synth.random_sampling(7, 50, seed=3)
    s = Sampling(T(pos, dev), T(d, dev), T(var, dev))
    ref = orc.NeRFOracle(w, activation_type=act, density_activation_type=dact).forward(pos, d, var)
    for dtype, rtol, atol in (("fp32", 1e-4, 2e-5), ("f16_split", 1e-4, 2e-5), ("bf16", 3e-2, 3e-2)):
        net.weight_dtype = dtype
        o = net(s)
        for k in ("density", "color"):
            scale = np.abs(ref[k]).max()
            assert_close(N(o[k]) / scale, ref[k] / scale, rtol, atol, "%s %s" % (dtype, k))


def test_render_rays_ndc_bf16_end_to_end(dev, orc, bunny_weights):
    """configs[4] end to end on 48 rays: NDC rays + point sampling + bf16 fields through neddf_render_rays vs the
    oracle's restatement with the same uniforms (bf16 emulation) -- and the fp32 variant of the same NDC render at 1e-4."""
    import neddf_amd
    g = golden("bunny_stages.npz")
    W, H, near = 400, 400, 1.0
    rng = np.random.default_rng(12)
    uv = rng.integers(40, 360, (48, 2)).astype(np.int64)
    # a forward-facing pose: camera at the origin region looking down -z (identity rotation)
    R, Tr = np.eye(3, dtype=np.float32), np.array([0.05, -0.02, 0.1], np.float32)
    calib = g["calib"].astype(np.float32)
    u_c, u_f = rng.uniform(0, 1, (48, 65)).astype(np.float32), rng.uniform(0, 1, (48, 129)).astype(np.float32)
    cfg = dict(BUNNY_CFG, _target_="neddf.network.NeDDF")
    r = neddf_amd.NeRFRender(cfg, sample_coarse=64, sample_fine=128, dist_near=0.0, dist_far=1.0, max_dist=1.0,
                             use_coarse_network=False, sampling_type="point")
    r.network_fine.load_state_dict({k: torch.from_numpy(v) for k, v in bunny_weights.items()})
    r.to(dev)
    r.set_iter(-1)
    r.ray_space, r.ndc_width, r.ndc_height, r.ndc_near = "ndc", W, H, near
    cam = neddf_amd.Camera(neddf_amd.PinholeCalib(calib.astype(np.float64)), None).to(dev)
    cam.R, cam.T = T(R, dev), T(Tr, dev)
    for dtype, tol in (("fp32", 1e-4), ("bf16", 2e-2)):
        r.network_fine.weight_dtype = dtype
        o = r._render(r._ctx(dev), T(uv, dev), cam, T(u_c, dev), T(u_f, dev), full=True)
        assert int(o["_nan"].item()) == 0
        net = orc.NeDDFOracle(bunny_weights, bf16=(dtype == "bf16"), **BUNNY_CFG)
        ref = orc.render_rays(net, net, uv, R, Tr, calib, u_c, u_f, 0.0, 1.0, 1.0, "point", ndc=(W, H, near))
        for k in ("color", "depth", "transmittance", "color_coarse", "depth_coarse", "transmittance_coarse"):
            assert_close(N(o[k]), ref[k], tol, tol * 0.1 + 1e-5, "%s %s" % (dtype, k))


def test_c5_full_frame_properties(dev, bunny_weights):
    """BASELINE.json configs[4] at full size: all 762 048 rays of a 1008x756 forward-facing view, NDC rays, point samples,
    hierarchical 65 + 194, bf16 operands (what `bench.py --workload c5` times).  Size-independent properties: fine distances sorted
    inside [0, 1] with every coarse knot present, no NaN flag, finite pixels; the frame is bit-identical under another batch split
    (tile tails and the 128-row bf16 shapes included); rendering is deterministic; and the bf16 frame stays within the policy's
    error of the fp32 frame of the same rays and uniforms (PSNR, median and 99th-percentile colour error)."""
    import neddf_amd
    W, H, near = 1008, 756, 1.0
    cfg = dict(BUNNY_CFG, _target_="neddf.network.NeDDF")
    r = neddf_amd.NeRFRender(cfg, sample_coarse=64, sample_fine=128, dist_near=0.0, dist_far=1.0, max_dist=1.0,
                             use_coarse_network=False, sampling_type="point")
    r.network_fine.load_state_dict({k: torch.from_numpy(v) for k, v in bunny_weights.items()})
    r.to(dev)
    r.set_iter(-1)
    r.ray_space, r.ndc_width, r.ndc_height, r.ndc_near = "ndc", W, H, near
    cam = neddf_amd.Camera(neddf_amd.PinholeCalib(np.array([815.13, 815.13, W / 2.0, H / 2.0])), None).to(dev)
    cam.R, cam.T = T(np.eye(3, dtype=np.float32), dev), T(np.array([0.05, -0.02, 0.1], np.float32), dev)
    desc = cam.descriptor()
    n = W * H
    gen = torch.Generator(device=dev).manual_seed(7)
    U_c = torch.rand(n, 65, device=dev, generator=gen)
    U_f = torch.rand(n, 129, device=dev, generator=gen)
    idx = torch.arange(n, device=dev)
    uv = torch.stack([idx % W, idx // W], 1)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)

    def frame(dtype, batch, check=False):
        r.network_fine.weight_dtype = dtype
        ctx = r._ctx(dev)          # (re-packs the slot's weights under the operand policy just selected)
        color, depth, trans = torch.empty(n, 3, device=dev), torch.empty(n, device=dev), torch.empty(n, device=dev)
        dc, df = torch.empty(batch, 65, device=dev), torch.empty(batch, 194, device=dev)
        for lo in range(0, n, batch):
            hi = min(n, lo + batch)
            b = hi - lo
            ctx.render_rays(uv[lo:hi], desc, r._params(), U_c[lo:hi], U_f[lo:hi],
                            dict(color=color[lo:hi], depth=depth[lo:hi], transmittance=trans[lo:hi], dists_coarse=dc[:b], dists_fine=df[:b], nan_flag=flag))
            if check:
                assert bool((df[:b, 1:] >= df[:b, :-1]).all()), lo
                merged = torch.sort(torch.cat([df[:b], dc[:b]], 1), dim=1)[0]
                assert bool((merged[:, 1:] == merged[:, :-1]).sum(1).ge(65).all()), lo
                assert float(df[:b].min()) >= 0.0 and bool((df[:b].max(1)[0] <= dc[:b].max(1)[0] + 1e-6).all())
        return color, depth, trans

    c16, d16, t16 = frame("bf16", 1 << 16, check=True)
    assert int(flag.item()) == 0
    for v in (c16, d16, t16):
        assert bool(torch.isfinite(v).all())
    c16b, d16b, t16b = frame("bf16", 50001)           # another batch split (odd size: every tile tail)
    assert torch.equal(c16, c16b) and torch.equal(d16, d16b) and torch.equal(t16, t16b)
    c32, d32, t32 = frame("fp32", 1 << 16)
    assert int(flag.item()) == 0
    err = (c16 - c32).abs()
    assert float(err.max()) > 0.0, "the bf16 frame is bit-identical to the fp32 one: the operand policy was not applied"
    mse = float(((c16 - c32).double() ** 2).mean())
    psnr = 10 * np.log10(1.0 / max(mse, 1e-20))
    print("\nc5 full frame: bf16 vs fp32 colour PSNR %.1f dB, median |err| %.2e, 99th percentile %.2e, max %.2e; depth max |err| %.2e"
          % (psnr, float(err.median()), float(torch.quantile(err.flatten()[::7], 0.99)), float(err.max()), float((d16 - d32).abs().max())))
    assert psnr > 45.0 and float(err.median()) < 2e-3 and float(torch.quantile(err.flatten()[::7], 0.99)) < 3e-2


def test_llff_ndc_training_and_eval_flow(dev, tmp_path, monkeypatch, capsys):
    """configs[4] as a workflow: LLFF-layout dataset -> scripts/run.py with NDC rays and a NeRF network pair (one epoch) ->
    scripts/run_eval.py on the result."""
    from PIL import Image
    from scipy.spatial.transform import Rotation
    from neddf_amd.scripts import run, run_eval
    root = tmp_path / "data" / "fern"
    (root / "images_4").mkdir(parents=True)
    rng = np.random.default_rng(4)
    rows = []
    for i in range(9):
        Rm = Rotation.from_euler("xyz", rng.normal(0, 0.05, 3)).as_matrix()
        t = rng.normal(0, 0.3, 3) + np.array([0.0, 0.0, 3.0])
        llff = np.concatenate([-Rm[:, 1:2], Rm[:, 0:1], Rm[:, 2:3], t[:, None], np.array([[64.0], [80.0], [260.0]])], 1)
        rows.append(np.concatenate([llff.reshape(-1), [2.0, 20.0]]))
        Image.fromarray(rng.integers(0, 256, (16, 20, 3), dtype=np.uint8)).save(root / "images_4" / ("img_%03d.png" % i))
    np.save(root / "poses_bounds.npy", np.stack(rows))
    monkeypatch.chdir(tmp_path)
    run.seed_everything()
    run.main(["trainer=test", "dataset=llff_fern", "dataset.dataset_dir=data/fern/", "render=llff_render", "network=nerf", "loss=nerf_loss",
              "trainer.batch_size=16", "trainer.epoch_max=0", "trainer.epoch_save_model=1", "render.sample_coarse=16",
              "render.sample_fine=24"])
    rd = next((tmp_path / "outputs").glob("*/*"))
    assert (rd / "models" / "model_00000.pth").is_file() and len(list((rd / "render" / "0000").glob("*_rgb.png"))) == 1
    monkeypatch.chdir(tmp_path)
    with torch.enable_grad():
        pass
    run_eval.main([str(rd), "--epoch", "0"])
    out = capsys.readouterr().out
    assert out.count("psnr:") == 2 and (rd / "eval" / "001_depth.png").is_file()


# ------------------------------------------------------------------ split-fp16 operands (fp32 data on the fp16 matrix instructions)
@pytest.mark.parametrize("dtype", ["f16_split"])
@pytest.mark.parametrize("mode", ["full", "minimal"])
def test_split_operand_fields_meet_the_fp32_gate(dev, orc, bunny_weights, mode, dtype):
    """weight_dtype = "f16_split": every operand split into two fp16 terms, three products per multiply-add, fp32
    accumulation.  Held to the SAME tolerances as the fp32 path (tests/test_gpu_parity.py) against the fp32 oracle."""
    from neddf_amd import Sampling
    pos, d, var = synth.random_sampling(40, 33, seed=21)
    net = _net(dev, bunny_weights, dtype, mode)
    s = Sampling(T(pos, dev), T(d, dev), T(var, dev))
    o = net(s)
    ref = orc.NeDDFOracle(bunny_weights, **BUNNY_CFG).forward(pos, d, var)
    net.weight_dtype = "fp32"
    o32 = net(s)
    tol = {"distance": (1e-4, 1e-6), "aux_grad": (1e-4, 1e-6), "color": (1e-4, 2e-5), "density": (1e-4, 3e-4), "fields_penalty": (1e-4, 1e-5)}
    for k in o:
        assert_close(N(o[k]), ref[k], *tol[k], dtype + " " + k)
        scale = np.abs(ref[k]).max()
        print("%-10s %-14s vs oracle %.2e, fp32 MFMA vs oracle %.2e (of max |value|)" % (
            dtype, k, np.abs(N(o[k]) - ref[k]).max() / scale, np.abs(N(o32[k]) - ref[k]).max() / scale))


@pytest.mark.parametrize("name", ["neddf_w128", "neddf_w192", "neddf_w384", "neddf_leaky", "neddf_relu", "neddf_skips2"])
def test_operand_policies_on_other_widths_and_activations(dev, name):
    """Round 3: the 16-bit operand policies on the engine widths other than 256 (128, 192 -> 256, 384), on ReLU / LeakyReLU fields (whose
    reverse pass carries y' as mask bits) and with two skip connections, both differentiation modes.  Split fp16 is held to the
    reference goldens at the fp32 path's own gate (1e-4 rel + 1e-5 abs; 2e-5 for the colour sums), bf16 against this library's fp32
    result (the configs[4] policy has no reference)."""
    import json
    g = golden(name + ".npz")
    kw = json.loads(str(g["config"]))
    sd = synth.neddf_state(kw["embed_pos_rank"], kw["embed_dir_rank"], kw["ddf_layer_count"], kw["ddf_layer_width"],
                           kw["col_layer_count"], kw["col_layer_width"], tuple(kw["skips"]), seed=7)
    from neddf_amd import Sampling
    s = Sampling(T(g["pos"], dev), T(g["dir"], dev), T(g["var"], dev))
    for mode in ("full", "minimal"):
        net = _net(dev, sd, "fp32", mode, kw)
        o32 = {k: N(v) for k, v in net(s).items()}
        net.weight_dtype = "f16_split"
        o = net(s)
        for k in o:
            assert_close(N(o[k]), g["eval_" + k], 1e-4, 2e-5 if k == "color" else 1e-5, "%s %s f16_split %s" % (name, mode, k))
        net.weight_dtype = "bf16"
        o = net(s)
        # bf16 (8 mantissa bits): the smooth outputs stay within 1 % of their range everywhere; density and colour depend on the
        # normal grad D / |grad D|, which is piecewise constant under ReLU / LeakyReLU and jumps where a rounding flips a unit
        # (measured, tools/bf16_stats.py: median 1e-3, 99th percentile <= 3.3e-2, isolated points up to 0.12 of the range -- the
        # same at width 256 and in both differentiation modes), so those two are gated on the bulk, not on the worst point
        for k in ("distance", "aux_grad"):
            scale = max(float(np.abs(o32[k]).max()), 1e-3)
            assert float(np.abs(N(o[k]) - o32[k]).max()) / scale < 1e-2, "%s %s bf16 %s" % (name, mode, k)
        for k in ("density", "color"):
            e = np.abs(N(o[k]) - o32[k]) / max(float(np.abs(o32[k]).max()), 1e-3)
            assert float(np.percentile(e, 99)) < 5e-2 and float(np.median(e)) < 5e-3 and float(e.max()) < 0.25, "%s %s bf16 %s" % (name, mode, k)


@pytest.mark.parametrize("name", ["nerf_w128", "nerf_w384", "nerf_skips2"])
def test_nerf_operand_policies_on_other_widths(dev, name):
    import json
    import neddf_amd
    from neddf_amd import Sampling
    g = golden(name + ".npz")
    kw = json.loads(str(g["config"]))
    sd = synth.nerf_state(kw["embed_pos_rank"], kw["embed_dir_rank"], kw["layer_count"], kw["layer_width"], tuple(kw["skips"]), seed=11)
    net = neddf_amd.NeRF(**kw)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net.to(dev)
    net.set_iter(-1)
    s = Sampling(T(g["pos"], dev), T(g["dir"], dev), T(g["var"], dev))
    net.weight_dtype = "f16_split"
    o = net(s)
    for k in ("density", "color"):
        assert_close(N(o[k]), g["eval_" + k], 1e-4, 2e-5, "%s f16_split %s" % (name, k))
    net.weight_dtype = "bf16"
    o = net(s)
    for k in ("density", "color"):
        scale = max(float(np.abs(g["eval_" + k]).max()), 1e-3)
        assert float(np.abs(N(o[k]) - g["eval_" + k]).max()) / scale < 3e-2, "%s bf16 %s" % (name, k)


@pytest.mark.parametrize("dtype", ["f16_split"])
def test_split_operand_render_rays_end_to_end(dev, bunny_weights, bunny_stages, dtype):
    """The golden 64-ray render of the reference (tests/test_gpu_parity.py::test_render_rays_end_to_end) with split-bf16 fields."""
    import neddf_amd
    g = bunny_stages
    cfg = dict(BUNNY_CFG, _target_="neddf.network.NeDDF")
    r = neddf_amd.NeRFRender(cfg, sample_coarse=64, sample_fine=128, dist_near=2.0, dist_far=6.0, max_dist=6.0,
                             use_coarse_network=False, sampling_type="cone")
    r.network_fine.load_state_dict({k: torch.from_numpy(v) for k, v in bunny_weights.items()})
    r.to(dev)
    r.set_iter(-1)
    r.network_fine.weight_dtype = dtype
    cam = neddf_amd.Camera(neddf_amd.PinholeCalib(g["calib"].astype(np.float64)), None).to(dev)
    cam.R, cam.T = T(g["R"], dev), T(g["T"], dev)
    o = r._render(r._ctx(dev), T(g["uv"], dev), cam, T(g["u_coarse"], dev), T(g["u_fine"], dev), full=True)
    assert int(o["_nan"].item()) == 0
    for k in ("color", "depth", "transmittance", "color_coarse", "depth_coarse", "transmittance_coarse"):
        assert_close(N(o[k]), g["out_" + k], 1e-4, 1e-5, k)
    mse = float(np.mean((N(o["color"]) - g["out_color"]) ** 2))
    assert 10 * np.log10(1.0 / max(mse, 1e-20)) > 80.0


@pytest.mark.parametrize("dtype", ["f16_split"])
def test_split_operand_neus(dev, dtype):
    import neddf_amd
    from neddf_amd import Sampling
    w = synth.neus_state()
    net = neddf_amd.NeuS()
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()})
    net.to(dev)
    pos, d, var = synth.random_sampling(6, 30, seed=5, cone=False)
    s = Sampling(T(pos, dev), T(d, dev), T(var, dev))
    a = net(s)
    net.weight_dtype = dtype
    b = net(s)
    for k in a:
        assert_close(N(b[k]), N(a[k]), 1e-4, 2e-5, "NeuS " + k)


def test_split_operands_saturate_instead_of_overflowing(dev):
    """fp16 terms top out at 65504: activations beyond that saturate (the result is no longer accurate, but stays finite)."""
    import neddf_amd
    from neddf_amd import Sampling
    w = {k: (v * (40.0 if k.endswith("weight") and k.startswith("layers.") else 1.0)).astype(np.float32)
         for k, v in synth.nerf_state(seed=11).items()}
    net = neddf_amd.NeRF(activation_type="ReLU", density_activation_type="ReLU")
    net.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    net.to(dev)
    pos, d, var = synth.random_sampling(4, 20, seed=3)
    s = Sampling(T(pos, dev), T(d, dev), T(var, dev))
    ref = net(s)
    assert float(ref["density"].abs().max()) > 1e6            # the fp32 path really is far outside fp16's range here
    net.weight_dtype = "f16_split"
    o = net(s)
    assert torch.isfinite(o["density"]).all() and torch.isfinite(o["color"]).all()


