"""Volume renderer -- host-side mirror of neddf/render/{base_neural_render,nerf_render}.py.

Same constructor keywords, methods, dict keys and RNG draw order as the
reference; the work is the fused HIP pipeline behind neddf_render_rays
(raygen -> stratified/cone sampling -> field -> wave-scan compositing ->
inverse-CDF resampling -> field -> compositing) on the current HIP stream.
"""
import math
from typing import Any, Dict, Iterable, List, Optional

import os

import torch
from torch import Tensor, nn

from ._lib import SLOT_COARSE, SLOT_FINE, Context, NeddfError, RenderParams
from .camera import Camera
from .config import instantiate
from .network import BaseNeuralField, NeDDF
from .rng import skip_uniforms

RenderTarget = str          # Literal["color", "depth", "transmittance"]
SamplingType = str          # Literal["point", "cone"]


class BaseNeuralRender(nn.Module):
    def __init__(self) -> None:
        super().__init__()
        self.iteration: int = -1

    def next_iter(self) -> None:
        self.set_iter(self.iteration + 1)

    def set_iter(self, iter: int) -> None:
        self.iteration = iter


class NeRFRender(BaseNeuralRender):
    """nerf_render.py:40-81.  Extra (non-reference) attributes:

    rng            "torch_cpu" (default; the reference draws its uniforms with
                   torch.rand on the CPU generator, nerf_render.py:137 and
                   base_neural_render.py:75 -- same seed => same samples) or
                   "device" (draw on the HIP device; faster, not seed-compatible)
    rays_per_call  rays handed to one neddf_render_rays call by render_image
    ray_space      "world" (the reference) or "ndc": forward-facing captures sampled along normalised-device-coordinate
                   rays of an ndc_width x ndc_height view with near plane ndc_near (original NeRF paper, appendix C;
                   the reference has no such mode).  dist_near / dist_far are then NDC depths (0 and 1) and the fields
                   still receive the world-space viewing direction.  Use sampling_type="point" with it.
    """

    def __init__(self, network_config: Any, sample_coarse: int = 128, sample_fine: int = 128, dist_near: float = 2.0,
                 dist_far: float = 6.0, max_dist: float = 6.0, use_coarse_network: bool = True,
                 sampling_type: SamplingType = "point", ray_space: str = "world", ndc_near: float = 1.0) -> None:
        super().__init__()
        self.use_coarse_network = use_coarse_network
        self.network_fine: BaseNeuralField = instantiate(network_config)
        self.network_coarse: BaseNeuralField = instantiate(network_config) if use_coarse_network else self.network_fine
        # library slots the modules work in when called directly (training step, stage API); render_rays' fused path
        # loads SLOT_COARSE / SLOT_FINE itself
        self.network_fine._slot = SLOT_FINE
        if use_coarse_network:
            self.network_coarse._slot = SLOT_COARSE
        self.sample_coarse, self.sample_fine = sample_coarse, sample_fine
        self.dist_near, self.dist_far, self.max_dist = dist_near, dist_far, max_dist
        self.sampling_type = sampling_type
        self.rng = "torch_cpu"
        self.rays_per_call = int(os.environ.get("NEDDF_RAYS_PER_CALL", 1 << 16))
        self.ray_space = ray_space          # the two keywords after sampling_type are not reference keywords
        self.ndc_width, self.ndc_height, self.ndc_near = 0, 0, ndc_near

    # ------------------------------------------------------------------ helpers
    def get_network(self) -> BaseNeuralField:
        return self.network_fine

    def get_parameters_list(self) -> List[Any]:
        if self.use_coarse_network:
            return list(self.network_coarse.parameters()) + list(self.network_fine.parameters())
        return list(self.network_coarse.parameters())

    def set_iter(self, iter: int) -> None:
        super().set_iter(iter)
        self.network_coarse.set_iter(iter)
        self.network_fine.set_iter(iter)

    def _params(self, nan_group: int = 0, nan_group_offset: int = 0) -> RenderParams:
        if self.sampling_type not in ("point", "cone"):
            raise ValueError("sampling_type must be 'point' or 'cone'")
        p = RenderParams()
        p.sample_coarse, p.sample_fine = self.sample_coarse, self.sample_fine
        p.dist_near, p.dist_far, p.max_dist = self.dist_near, self.dist_far, self.max_dist
        p.cone_sampling = int(self.sampling_type == "cone")
        # rays per sample_pdf NaN-fallback decision (render_image: its chunk) and where in a group this batch starts
        p.nan_group, p.nan_group_offset = int(nan_group), int(nan_group_offset)
        p.ray_radius = 1.0 / 1111 / math.sqrt(12)      # nerf_render.py:144-145
        if self.ray_space not in ("world", "ndc"):
            raise ValueError("ray_space must be 'world' or 'ndc'")
        p.ndc_rays = int(self.ray_space == "ndc")
        if p.ndc_rays:
            if self.ndc_width < 1 or self.ndc_height < 1:
                raise ValueError("ray_space='ndc' needs ndc_width / ndc_height (the full image size)")
            p.ndc_width, p.ndc_height, p.ndc_near = int(self.ndc_width), int(self.ndc_height), float(self.ndc_near)
        return p

    def _ctx(self, device) -> Context:
        ctx = Context.get(device)
        self.network_coarse.upload(ctx, SLOT_COARSE)
        self.network_fine.upload(ctx, SLOT_FINE)
        return ctx

    def _rand(self, rows: int, cols: int, device) -> Tensor:
        if self.rng == "torch_cpu":     # page-locked, so the upload overlaps the draw of the next batch
            return torch.rand(rows, cols, pin_memory=torch.device(device).type == "cuda").to(device, non_blocking=True)
        if self.rng == "device":
            return torch.rand(rows, cols, device=device)
        raise ValueError("rng must be 'torch_cpu' or 'device'")

    def _has_penalty(self) -> bool:
        return isinstance(self.network_fine, NeDDF)

    # --------------------------------------------------------------- stage API
    def integrate_volume_render(self, dists: Tensor, densities: Tensor, colors: Tensor) -> Dict[str, Tensor]:
        """base_neural_render.py:117-172: weight [B,S-1], depth [B], color [B,3], transmittance [B]."""
        ctx = Context.get(dists.device)
        if torch.is_grad_enabled() and (densities.requires_grad or colors.requires_grad):
            from .autograd import CompositeFunction
            w, depth, color, trans, flag = CompositeFunction.apply(ctx, self.max_dist, dists.detach(), densities, colors)
            out = dict(weight=w, depth=depth, color=color, transmittance=trans)
        else:
            out, flag = ctx.composite(dists, densities, colors, self.max_dist)
        assert int(flag.item()) == 0, "NaN weight in integrate_volume_render"      # reference asserts (:155)
        return out

    def sample_pdf(self, dists: Tensor, weights: Tensor, samples_fine: int, cat_coarse: bool = True) -> Tensor:
        """base_neural_render.py:27-115.  `weights` is sanitised in place like the reference."""
        ctx = Context.get(dists.device)
        U = self._rand(dists.shape[0], samples_fine, dists.device).contiguous()
        if weights.dtype == torch.float32 and weights.is_contiguous():
            return ctx.importance_resample(dists, weights, U, cat_coarse)
        w = weights.to(torch.float32).contiguous()
        out = ctx.importance_resample(dists, w, U, cat_coarse)
        weights.copy_(w)
        return out

    # -------------------------------------------------------------- render_rays
    def _render(self, ctx: Context, uv: Tensor, camera: Camera, U_c: Tensor, U_f: Tensor, full: bool, cam_desc=None,
                nan_group: int = 0, nan_group_offset: int = 0) -> Dict[str, Tensor]:
        B = uv.shape[0]
        dev = uv.device
        S2 = self.sample_coarse + self.sample_fine + 2

        def buf(*shape):
            return torch.empty(*shape, device=dev, dtype=torch.float32)

        o = dict(color=buf(B, 3), depth=buf(B), transmittance=buf(B))
        if full:
            o.update(weight=buf(B, S2 - 1), color_coarse=buf(B, 3), depth_coarse=buf(B), transmittance_coarse=buf(B),
                     weight_coarse=buf(B, self.sample_coarse))
            if self._has_penalty():
                o.update(fields_penalty=buf(B), fields_penalty_coarse=buf(B))
        flag = torch.zeros(1, device=dev, dtype=torch.int32)
        ctx.render_rays(uv, camera.descriptor() if cam_desc is None else cam_desc, self._params(nan_group, nan_group_offset), U_c, U_f, dict(o, nan_flag=flag))
        o["_nan"] = flag
        return o

    def render_rays(self, uv: Tensor, camera: Camera) -> Dict[str, Tensor]:
        """nerf_render.py:109-188.  Keys: weight, depth, color, transmittance[, fields_penalty] + *_coarse."""
        uv = uv.to(camera.device)
        B = uv.shape[0]
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._render_rays_with_grad(uv, camera)
        ctx = self._ctx(uv.device)
        U_c = self._rand(B, self.sample_coarse + 1, uv.device)       # draw order is part of the contract
        U_f = self._rand(B, self.sample_fine + 1, uv.device)
        o = self._render(ctx, uv, camera, U_c, U_f, full=True)
        assert int(o.pop("_nan").item()) == 0, "NaN weight in integrate_volume_render"
        order = ["weight", "depth", "color", "transmittance", "fields_penalty", "weight_coarse", "depth_coarse",
                 "color_coarse", "transmittance_coarse", "fields_penalty_coarse"]
        return {k: o[k] for k in order if k in o}

    def _render_rays_with_grad(self, uv: Tensor, camera: Camera) -> Dict[str, Tensor]:
        """render_rays as the training step needs it (nerf_render.py:109-188 under autograd): the samplers run as in
        inference, the two field evaluations and the two volume integrals are autograd nodes (autograd.py)."""
        from .ray import Sampling
        ctx = Context.get(uv.device)
        B = uv.shape[0]
        p = self._params()
        U_c = self._rand(B, self.sample_coarse + 1, uv.device)
        U_f = self._rand(B, self.sample_fine + 1, uv.device)
        radius = p.ray_radius if p.cone_sampling else None
        with torch.no_grad():
            cam = camera.descriptor()
            rd, ro = ctx.raygen(uv, cam)
            view = None
            if p.ndc_rays:
                view = rd
                rd, ro = ctx.rays_to_ndc(rd, ro, p.ndc_width, p.ndc_height, cam.calib[0], cam.calib[1], p.ndc_near)
            dists_c = ctx.sample_coarse(U_c, self.dist_near, self.dist_far)
            smp_c = Sampling(*ctx.sampling(rd, ro, dists_c, radius, view))
        val_c = self.network_coarse(smp_c)
        integ_c = self.integrate_volume_render(dists_c, val_c["density"], val_c["color"])
        for key in val_c:
            if "penalty" in key:
                delta = dists_c[:, 1:] - dists_c[:, :-1]
                integ_c[key] = torch.sum(delta * val_c[key].reshape(B, -1)[:, :-1], dim=1)
        with torch.no_grad():
            # sanitises the coarse weights in place, as the reference does under set_grad_enabled(False)
            dists_f = ctx.importance_resample(dists_c, integ_c["weight"].detach(), U_f, True)
            smp_f = Sampling(*ctx.sampling(rd, ro, dists_f, radius, view))
        val_f = self.network_fine(smp_f)
        integ = self.integrate_volume_render(dists_f, val_f["density"], val_f["color"])
        for key in val_f:
            if "penalty" in key:
                delta = dists_f[:, 1:] - dists_f[:, :-1]
                integ[key] = torch.sum(delta * val_f[key].reshape(B, -1)[:, :-1], dim=1)
        for key in list(integ_c):
            integ["{}_coarse".format(key)] = integ_c[key]
        return integ

    # ------------------------------------------------------------- render_image
    def render_image(self, width: int, height: int, camera: Camera, target_types: Iterable[RenderTarget],
                     downsampling: int = 1, chunk: int = 512, pixel_range=None) -> Dict[str, Tensor]:
        """nerf_render.py:190-249.  Pixels are row-major (idx = v*w + u).  In
        "torch_cpu" RNG mode the uniforms are drawn chunk by chunk in the
        reference's order ([B,Sc+1] then [B,Sf+1] per chunk of `chunk` rays), so a
        given torch seed renders the same samples; rays are then processed in
        batches of `rays_per_call` regardless of `chunk` (rays are independent).
        pixel_range=(lo, hi) (not in the reference) renders only that slab of the
        row-major pixel index and returns flat [hi-lo, C] tensors -- the unit of
        multi-GPU ray sharding (neddf_amd/parallel.py); the slab's uniforms are the
        ones the whole-frame draw would have given it (generator jump-ahead, rng.py),
        so the image does not depend on the sharding."""
        target_types = list(target_types)
        with torch.no_grad():
            dev = camera.device
            w, h = width // downsampling, height // downsampling
            us = torch.arange(w, device=dev).reshape(1, w).expand(h, w).reshape(-1) * downsampling
            vs = torch.arange(h, device=dev).reshape(h, 1).expand(h, w).reshape(-1) * downsampling
            uv = torch.stack([us, vs], 1)
            n = uv.shape[0]
            self.network_coarse.eval()
            self.network_fine.eval()
            ctx = self._ctx(dev)
            parts: Dict[str, List[Tensor]] = {k: [] for k in target_types}
            flags = []
            lo, hi = (0, n) if pixel_range is None else pixel_range

            cam_desc = camera.descriptor()          # three device -> host reads: once per image, not per batch

            def launch(below, above, U_c, U_f):
                # sample_pdf's NaN fallback keeps the reference's per-chunk granularity: batches start on chunk boundaries,
                # except a slab's first batch, which may start inside a chunk another rank shares
                o = self._render(ctx, uv[below:above], camera, U_c, U_f, full=False, cam_desc=cam_desc, nan_group=chunk,
                                 nan_group_offset=below % chunk)
                flags.append(o["_nan"])
                for k in target_types:
                    parts[k].append(o[k])

            if self.rng == "torch_cpu":
                # Draw chunk by chunk in the reference's order, but hand rays_per_call rays at a time to the GPU:
                # launches are asynchronous, so the host draws the next batch while the device renders this one.
                # A slab (pixel_range) starts its draws where the reference's stream would be at its first chunk:
                # the generator JUMPS there (rng.py, milliseconds) instead of drawing and discarding, so the host
                # cost of a rank shrinks with the slab; after the slab it jumps to where the whole frame would have
                # left it, i.e. every rank ends in the reference's generator state.
                per_ray = self.sample_coarse + 1 + self.sample_fine + 1
                first = (lo // chunk) * chunk if hi > lo else n      # chunks before the slab are all full
                skip_uniforms(first * per_ray)
                uc, uf, start, count, below = [], [], first, 0, first
                while below < min(n, hi):
                    b = min(n, below + chunk) - below
                    uc.append(torch.rand(b, self.sample_coarse + 1)); uf.append(torch.rand(b, self.sample_fine + 1))
                    count += b
                    below += b
                    if count >= self.rays_per_call or below >= min(n, hi):
                        a0, a1 = max(start, lo), min(start + count, hi)
                        U_c = torch.cat(uc)[a0 - start:a1 - start].to(dev, non_blocking=True)
                        U_f = torch.cat(uf)[a0 - start:a1 - start].to(dev, non_blocking=True)
                        launch(a0, a1, U_c, U_f)
                        uc, uf, start, count = [], [], start + count, 0
                skip_uniforms((n - below) * per_ray)
            else:
                # batches of whole chunks (sample_pdf's NaN fallback is decided per chunk, never on part of one)
                step = max(chunk, self.rays_per_call // chunk * chunk)
                for below in range(lo, hi, step):
                    above = min(hi, below + step)
                    launch(below, above, self._rand(above - below, self.sample_coarse + 1, dev),
                           self._rand(above - below, self.sample_fine + 1, dev))
            assert not flags or int(torch.stack(flags).sum().item()) == 0, "NaN weight in integrate_volume_render"
            if pixel_range is None:
                images = {k: torch.cat(parts[k], 0).reshape(h, w, -1) for k in target_types}
            elif hi <= lo:          # an empty slab (more ranks than chunks)
                images = {k: torch.empty(0, 3 if k == "color" else 1, device=dev) for k in target_types}
            else:
                images = {k: torch.cat(parts[k], 0).reshape(hi - lo, -1) for k in target_types}
            self.network_coarse.train(True)
            self.network_fine.train(True)
        return images

    def render_image_single_pass(self, width: int, height: int, camera: Camera, samples: int,
                                 U: Optional[Tensor] = None, pixel_range=None) -> Dict[str, Tensor]:
        """One stratified pass of `samples` points per ray through network_fine
        (BASELINE.json configs[1]).  Not a reference method: it is render_rays'
        coarse half (nerf_render.py:128-152) at image scale.  pixel_range=(lo,hi)
        restricts to a slab of the row-major pixel index (multi-GPU sharding).
        Returns flat per-ray tensors color [n,3], depth [n], transmittance [n]."""
        with torch.no_grad():
            dev = camera.device
            lo, hi = (0, width * height) if pixel_range is None else pixel_range
            idx = torch.arange(lo, hi, device=dev)
            uv = torch.stack([idx % width, idx // width], 1)
            n = hi - lo
            ctx = self._ctx(dev)
            out = dict(color=torch.empty(n, 3, device=dev), depth=torch.empty(n, device=dev),
                       transmittance=torch.empty(n, device=dev))
            flag = torch.zeros(1, device=dev, dtype=torch.int32)
            cam_desc = camera.descriptor()
            for below in range(0, n, self.rays_per_call):
                above = min(n, below + self.rays_per_call)
                o = {k: v[below:above] for k, v in out.items()}
                # without U the uniforms are drawn per batch: in "torch_cpu" mode the host draws batch k+1 while batch k renders
                Ub = U[below:above] if U is not None else self._rand(above - below, samples, dev)
                ctx.render_rays(uv[below:above], cam_desc, self._params(), Ub, None,
                                dict(o, nan_flag=flag), single_slot=SLOT_FINE)
            out["_nan"] = flag
        return out

    def render_field_slice(self, slice_t: float = 0.0, render_size: float = 1.1, render_resolution: int = 128,
                           colormap: bool = True):
        """nerf_render.py:263-336: z = slice_t plane of the fine network's fields as uint8 images (debug view).
        Scalar fields are JET colour-mapped (BGR, same control points as cv2.COLORMAP_JET; cv2 itself is not a
        dependency here), colour is scaled by 256.  colormap=False (not a reference keyword) returns the scalar
        fields as the [n,n,1] uint8 images the reference hands to cv2.applyColorMap."""
        import numpy as np
        from .ray import Sampling
        with torch.no_grad():
            device = self.network_fine.device
            n = render_resolution
            lin = torch.linspace(-render_size, render_size, n, device=device)
            xs = lin.reshape(1, n).expand(n, n)
            ys = -lin.reshape(n, 1).expand(n, n)
            zs = torch.zeros(n, n, device=device) + slice_t
            pos = torch.stack([xs, ys, zs], 2).contiguous()
            d = torch.zeros(n, n, 3, device=device)
            d[:, :, 2] = 1.0
            self.network_fine.train(False)
            values = self.network_fine(Sampling(pos, d, torch.zeros_like(pos)))
            self.network_fine.train(True)
            scales = {"distance": 256.0, "density": 12.8, "color": 256.0, "aux_grad": 256.0}
            fields = {}
            for key, val in values.items():
                if key not in scales:
                    continue
                f = (scales[key] * val.reshape(n, n, -1)).cpu().numpy().clip(0, 255).astype(np.uint8)
                fields[key] = jet_bgr(f[:, :, 0]) if (f.shape[2] == 1 and colormap) else f
            return fields


def jet_bgr(gray):
    """uint8 [h,w] -> uint8 [h,w,3] (B,G,R): cv2.applyColorMap(gray, cv2.COLORMAP_JET) restated (nerf_render.py:330; cv2 is in neither
    container, so this is pinned on the published algorithm, not on cv2's output).  OpenCV builds the table from three float arrays
    r, g, b of 256 entries -- the piecewise-linear ramps clamp(1.5 - |4 i/255 - c|, 0, 1) with c = 3, 2, 1, written out as literals --
    and converts them with `convertTo(CV_8U, 255.)`: a float product and cvRound (round half to EVEN).  Every ramp entry times 255 is
    an exact half-integer (382.5 - |4 i - 255 c|), so the rounding mode decides each of them: the ramps step by 4 (blue: 128, 132,
    136, ... at i = 0, 1, 2, ...; entries 0 / 255 are (128, 0, 0) / (0, 0, 128) in B, G, R).  Round 3's `+ 0.5` truncation in double
    differed from this on 149 of the 768 entries by one count."""
    import numpy as np
    x = np.arange(256, dtype=np.float64) / 255.0
    lut = np.stack([np.rint(np.clip(1.5 - np.abs(4 * x - c), 0, 1).astype(np.float32) * np.float32(255.0)) for c in (1, 2, 3)], 1)
    return lut.astype(np.uint8)[gray]
