"""Neural fields -- host-side mirror of neddf/network/{neddf,nerf,base_neuralfield}.py.

The modules own their parameters under the reference's state-dict keys
(`layers_ddf.N.weight` [in,out], ... / `layers.N.weight` [out,in], ...) so that
checkpoints written by the reference load unchanged; `forward(sampling)` keeps
the reference signature and dict keys but the computation is the fused HIP
field kernels (csrc/field_kernels.hip) reached through the C ABI.
"""
import itertools
import math
from abc import ABC, abstractmethod
from typing import Dict, List, Optional

import torch
from torch import Tensor, nn

from ._lib import ACT, DTYPE, FIELD_NEDDF, FIELD_NERF, FIELD_NEUS, OUT_FULL, OUT_MINIMAL, PENALTY_KEYS, SLOT_GENERIC, Context, FieldDesc
from .ray import Sampling


def lowpass_scale(alpha: float, embed_dim: int) -> List[float]:
    """Per-frequency progressive low-pass (with_grad/positional_encoding.py:137-157)."""
    s = [1.0] * embed_dim
    if alpha >= embed_dim:
        return s
    k = int(alpha)
    s[k] = 0.5 * (1 - math.cos(math.pi * (alpha - k))) + 1e-7
    for j in range(k + 1, embed_dim):
        s[j] = 1e-7
    # the reference stores the scale in a float32 tensor
    return torch.tensor(s, dtype=torch.float32).tolist()


class PositionalEncodingInfo(nn.Module):
    """Carries embed_dim / freq like the reference PE layers (their tensors are
    plain attributes, not buffers, so they never enter the state dict)."""

    def __init__(self, embed_dim: int) -> None:
        super().__init__()
        self.embed_dim = embed_dim
        self.freq = torch.tensor([(2.0 ** t) for t in range(embed_dim)])

    def get_grad_scale(self, input_dim: int = 3) -> Tensor:
        return torch.reciprocal(0.5 * self.freq).unsqueeze(1).expand(-1, input_dim).reshape(1, -1)

    def get_lowpass_scale(self, alpha: float = 1.0, input_dim: int = 3) -> Tensor:
        s = torch.tensor(lowpass_scale(alpha, self.embed_dim), dtype=torch.float32)
        return s.unsqueeze(1).expand(-1, input_dim).reshape(1, -1)


class LinearGradLayer(nn.Module):
    """Parameter holder with the reference layout weight [in,out], bias [out]
    (with_grad/linear.py:113-116: xavier-normal weight, zero bias)."""

    def __init__(self, input_ch: int = 128, output_ch: int = 128) -> None:
        super().__init__()
        self.input_ch, self.output_ch = input_ch, output_ch
        self.weight = nn.Parameter(torch.empty(input_ch, output_ch))
        self.bias = nn.Parameter(torch.zeros(output_ch))
        nn.init.xavier_normal_(self.weight)


_module_ids = itertools.count(1)

TRAIN_ENGINE_WIDTH = 256        # hidden width of the training kernels (csrc/train_*.hip)


def _pad_axis(t: Tensor, dim: int, segments) -> Tensor:
    """Zero-pad consecutive segments of `t` along `dim`: segments = [(length, padded_length), ...] covering the axis.  Built
    from narrow / cat, so the result carries the autograd graph back to `t` (the padding receives no gradient that matters:
    it is sliced off on the way back)."""
    if all(a == b for a, b in segments):
        return t
    parts, off = [], 0
    for length, padded in segments:
        piece = t.narrow(dim, off, length)
        if padded > length:
            shape = list(t.shape)
            shape[dim] = padded - length
            piece = torch.cat([piece, t.new_zeros(shape)], dim)
        parts.append(piece)
        off += length
    assert off == t.shape[dim], (off, tuple(t.shape), dim)
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim)


class BaseNeuralField(ABC, nn.Module):
    def __init__(self) -> None:
        super().__init__()
        self._slot = SLOT_GENERIC
        self._uid = next(_module_ids)        # identity in upload signatures (id() values are recycled by the allocator)
        # operand type of the 256-wide dense layers (not a reference keyword): "fp32" = exact fp32 MFMA, the parity path;
        # "bf16" = bf16 weights and activations with fp32 accumulation (BASELINE.json configs[4]); NeDDF / NeuS only
        self.weight_dtype = "fp32"
        self._epoch = 0                      # bumped by invalidate(): part of the upload signature

    def invalidate(self) -> None:
        """Tell the fused inference path that the parameters changed behind autograd's back.  The packed copy in the
        library is refreshed when a parameter's storage or its version counter changes -- optimiser steps, `copy_`,
        `load_state_dict`, `.to()` all do that -- but writes through `p.data` (`p.data.mul_()`, EMA / weight surgery code)
        bump no counter: call this after them, or the renderer keeps the stale packed weights.  (The training kernels
        read the live tensors and need nothing.)"""
        self._epoch += 1

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    @abstractmethod
    def set_iter(self, iter: int) -> None:
        raise NotImplementedError()

    @abstractmethod
    def _descriptor(self) -> FieldDesc:
        raise NotImplementedError()

    @abstractmethod
    def _tensors(self):
        """(weights, biases) in the order neddf_set_field documents."""
        raise NotImplementedError()

    @abstractmethod
    def _iter_state(self):
        """(aux_grad_scale, distance_range_max, lowpass list)"""
        raise NotImplementedError()

    def voxelize(self, field_name: str = "density", cube_range: float = 1.1, cube_resolution: int = 64,
                 chunk: int = 65536):
        """base_neuralfield.py:49-79: `field_name` on a cube_resolution^3 grid (meshing / visualisation)."""
        import numpy as np
        with torch.no_grad():
            ids = np.linspace(-cube_range, cube_range, cube_resolution)
            zs, ys, xs = np.meshgrid(ids, ids, ids)
            dev = self.device
            pos = torch.from_numpy(np.stack([xs, ys, zs], -1).reshape(1, -1, 3).astype(np.float32)).to(dev)
            d = torch.tensor([1.0, 0.0, 0.0], device=dev).expand_as(pos).contiguous()
            val = self.forward(Sampling(pos, d, torch.zeros_like(pos)))     # one call: no need to chunk on 288 GB
            return val[field_name].reshape(cube_resolution, cube_resolution, cube_resolution).cpu().numpy()

    # ---- training at hidden widths other than 256 ---------------------------------------------------------------------
    # The training kernels are built for hidden width 256 (every field kind) and, on their per-layer route in 256 x 256 blocks,
    # 512 (csrc/train_capi.hip train_supported).  Any other width trains ZERO-PADDED to the next of the two: every parameter
    # tensor is padded with differentiable torch ops (narrow / cat), the kernels see the padded network, and autograd slices the
    # parameter gradients back.  Exact: a padded unit has zero weights and bias, so it outputs a(0) = 0 under every activation of
    # the reference, feeds zero weight rows downstream and receives a zero gradient; the real parameters' gradients are what the
    # reference computes (tests/golden/train_widths.npz: 128 / 192 / 384 / 512).  Anything above 512 is
    # refused by the library.
    def _train_layout(self):
        """None, or (weight layouts, bias layouts): per tensor, per axis, the [(length, padded_length)] segments."""
        return None

    def _train_tensors(self, ws, bs):
        """(descriptor, weights, biases) the training kernels are called with."""
        desc = self._descriptor()
        layout = self._train_layout()
        if layout is None:
            return desc, ws, bs
        wl, bl = layout
        pw = [w if lay is None else _pad_axis(_pad_axis(w, 0, lay[0]), 1, lay[1]) if w.dim() == 2 else _pad_axis(w, 0, lay[0])
              for w, lay in zip(ws, wl)]
        pb = [b if lay is None else _pad_axis(b, 0, lay[0]) for b, lay in zip(bs, bl)]
        desc.layer_width = self._train_engine_width()
        if desc.kind != FIELD_NERF:
            desc.col_layer_width = desc.layer_width
        return desc, pw, pb

    def _train_engine_width(self) -> int:
        """Hidden width the training kernels see for this module (the padded one)."""
        return TRAIN_ENGINE_WIDTH

    def upload(self, ctx: Context, slot: int, weights: bool = True, train=None) -> None:
        """Pack + upload the parameters into `slot` if they changed since the last upload.  weights=False (training
        step: the kernels read the live parameter tensors) only makes sure the slot describes this architecture;
        train = (descriptor, weights, biases) of _train_tensors when that is a zero-padded one."""
        ws, bs = self._tensors() if train is None else (train[1], train[2])
        desc = self._descriptor() if train is None else train[0]
        desc.weight_dtype = DTYPE[self.weight_dtype]
        sig = (self._uid, slot, bytes(desc), tuple((t.data_ptr(), t._version) for t in ws + bs), self._epoch)
        have = ctx.slot_owner.get(slot)
        if not weights and have is not None and have[:3] == sig[:3]:
            pass            # same module, same architecture: the training kernels do not read the slot's weights
        elif have != sig:
            hw = [t.detach().to("cpu", torch.float32).contiguous() for t in ws]
            hb = [t.detach().to("cpu", torch.float32).contiguous() for t in bs]
            ctx.set_field(slot, desc, hw, hb, sig)
        ags, drm, lp = self._iter_state()
        ctx.set_iter(slot, ags, drm, lp)

    def _run(self, sampling: Sampling, out_mode: int, want) -> Dict[str, Tensor]:
        pos = sampling.sample_pos
        ctx = Context.get(pos.device)
        self.upload(ctx, self._slot)
        batch_size, sampling_size = pos.shape[0], pos.shape[1]
        o = ctx.field_forward(self._slot, pos, sampling.sample_dir, sampling.diag_variance, out_mode, want)
        return {k: (v.view(batch_size, sampling_size, 3) if k == "color" else v.view(batch_size, sampling_size))
                for k, v in o.items()}


class NeDDF(BaseNeuralField):
    """Density-distance field (neddf.py:52-160 constructor keywords)."""

    def __init__(self, embed_pos_rank: int = 10, embed_dir_rank: int = 4, ddf_layer_count: int = 8,
                 ddf_layer_width: int = 256, col_layer_count: int = 8, col_layer_width: int = 256,
                 activation_type: str = "tanhExp", density_activation_type: str = "ReLU", d_near: float = 0.01,
                 lowpass_alpha_offset: float = 10.0, skips: Optional[List[int]] = None,
                 penalty_weight: Optional[Dict[str, float]] = None) -> None:
        super().__init__()
        in_ddf = embed_pos_rank * 6
        in_col = (embed_pos_rank + embed_dir_rank) * 6 + 3 + ddf_layer_width
        self.skips = [4] if skips is None else list(skips)
        self.activation_type, self.density_activation_type = activation_type, density_activation_type
        self.pe_pos = PositionalEncodingInfo(embed_pos_rank)
        self.pe_dir = PositionalEncodingInfo(embed_dir_rank)
        ddf = [LinearGradLayer(in_ddf, ddf_layer_width)]
        for layer_id in range(ddf_layer_count - 2):
            ddf.append(LinearGradLayer(ddf_layer_width + (in_ddf if layer_id in self.skips else 0), ddf_layer_width))
        col = [LinearGradLayer(in_col, col_layer_width)]
        col += [LinearGradLayer(col_layer_width, col_layer_width) for _ in range(col_layer_count - 2)]
        self.layers_ddf = nn.ModuleList(ddf)
        self.layers_col = nn.ModuleList(col)
        self.layer_ddf_out = LinearGradLayer(ddf_layer_width, 1)
        self.layer_aux_out = LinearGradLayer(ddf_layer_width, 1)
        self.layer_col_out = LinearGradLayer(ddf_layer_width, 3)     # sic: ddf width (neddf.py:145)
        self.ddf_layer_count, self.ddf_layer_width = ddf_layer_count, ddf_layer_width
        self.col_layer_count, self.col_layer_width = col_layer_count, col_layer_width
        self.d_near = d_near
        self.aux_grad_scale = 1.1
        self.distance_range_max = 2.0
        self.lowpass_alpha_offset = lowpass_alpha_offset
        self.lowpass_alpha = lowpass_alpha_offset
        if penalty_weight is None:      # neddf.py:152-159
            penalty_weight = {"constraints_aux_grad": 0.05, "constraints_dDdt": 0.05, "constraints_color": 0.01,
                              "range_distance": 1.0, "range_aux_grad": 1.0}
        self.penalty_weight = dict(penalty_weight)
        # "full" reproduces every key of the reference dict; "minimal" skips the
        # colour-trunk Jacobian + penalties that eval rendering never reads
        self.output_mode = "full"

    def _descriptor(self) -> FieldDesc:
        d = FieldDesc()
        d.kind = FIELD_NEDDF
        d.embed_pos_rank, d.embed_dir_rank = self.pe_pos.embed_dim, self.pe_dir.embed_dim
        d.layer_count, d.layer_width = self.ddf_layer_count, self.ddf_layer_width
        d.col_layer_count, d.col_layer_width = self.col_layer_count, self.col_layer_width
        d.n_skips = len(self.skips)
        for i, s in enumerate(self.skips[:8]):
            d.skips[i] = s
        d.activation, d.density_activation = ACT[self.activation_type], ACT[self.density_activation_type]
        d.d_near = self.d_near
        for i, k in enumerate(PENALTY_KEYS):
            d.penalty_has[i] = int(k in self.penalty_weight)
            d.penalty_weight[i] = float(self.penalty_weight.get(k, 1.0))
        return d

    def _tensors(self):
        mods = list(self.layers_ddf) + list(self.layers_col) + [self.layer_ddf_out, self.layer_aux_out, self.layer_col_out]
        return [m.weight for m in mods], [m.bias for m in mods]

    def _iter_state(self):
        return self.aux_grad_scale, self.distance_range_max, lowpass_scale(self.lowpass_alpha, self.pe_pos.embed_dim)

    def _train_engine_width(self) -> int:
        return TRAIN_ENGINE_WIDTH if self.ddf_layer_width <= TRAIN_ENGINE_WIDTH else 2 * TRAIN_ENGINE_WIDTH

    def _train_layout(self):
        W, E = self.ddf_layer_width, self._train_engine_width()
        if W >= E or self.col_layer_width != W:       # E itself needs no padding; wider than 512: the library refuses
            return None
        cpe, small = 6 * self.pe_pos.embed_dim, 6 * (self.pe_pos.embed_dim + self.pe_dir.embed_dim) + 3
        hid, same = [(W, E)], lambda n: [(n, n)]
        wl, bl = [], []
        for l in range(len(self.layers_ddf)):               # [in, out]; cat([embed_pos_scaled, h]) puts the encoding first
            wide = l > 0 and (l - 1) in self.skips
            wl.append((same(cpe) if l == 0 else (same(cpe) + hid if wide else hid), hid))
            bl.append((hid,))
        for l in range(len(self.layers_col)):
            wl.append((same(small) + hid if l == 0 else hid, hid))
            bl.append((hid,))
        for n in (1, 1, 3):                                  # heads: [W, n]
            wl.append((hid, same(n)))
            bl.append(None)
        return wl, bl

    def _forward_with_grad(self, sampling: Sampling) -> Dict[str, Tensor]:
        """Training-mode forward: one autograd node over the layer-by-layer HIP kernels (csrc/train_*.hip)."""
        from .autograd import FieldFunction
        pos = sampling.sample_pos
        ctx = Context.get(pos.device)
        ws, bs = self._tensors()
        desc, ws, bs = self._train_tensors(ws, bs)
        self.upload(ctx, self._slot, weights=False, train=(desc, ws, bs) if self._train_layout() is not None else None)
        B, S = pos.shape[0], pos.shape[1]
        distance, density, color, penalty, aux = FieldFunction.apply(
            ctx, self._slot, self._iter_state(), len(ws), pos.detach(), sampling.sample_dir.detach(),
            sampling.diag_variance.detach(), *ws, *bs)
        return {"distance": distance.view(B, S), "density": density.view(B, S), "color": color.view(B, S, 3),
                "fields_penalty": penalty.view(B, S), "aux_grad": aux.view(B, S)}

    def forward(self, sampling: Sampling) -> Dict[str, Tensor]:
        """distance, density, fields_penalty, aux_grad [B,S]; color [B,S,3] (neddf.py:302-308).

        With autograd enabled and trainable parameters the outputs carry the graph (like the reference's always do);
        under torch.no_grad() the fused inference kernels run."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._forward_with_grad(sampling)
        if self.output_mode == "full":
            return self._run(sampling, OUT_FULL, ("distance", "density", "color", "fields_penalty", "aux_grad"))
        return self._run(sampling, OUT_MINIMAL, ("distance", "density", "color", "aux_grad"))

    def set_iter(self, iter: int) -> None:
        """Warm-up schedule (neddf.py:311-326); -1 = evaluation."""
        if iter == -1:
            self.aux_grad_scale, self.distance_range_max = 1.1, 2.0
            self.lowpass_alpha = self.pe_pos.embed_dim
        else:
            self.aux_grad_scale = min(1.1, max(0.01, 0.0001 * iter))
            self.distance_range_max = min(2.0, 2.0 + 0.0001 * iter)
            self.lowpass_alpha = self.lowpass_alpha_offset + 0.001 * iter


class NeRF(BaseNeuralField):
    """Plain NeRF field (nerf.py:34-105 constructor keywords)."""

    def __init__(self, embed_pos_rank: int = 10, embed_dir_rank: int = 4, layer_count: int = 8, layer_width: int = 256,
                 activation_type: str = "ReLU", density_activation_type: str = "ReLU", skips: Optional[List[int]] = None,
                 lowpass_alpha_offset: float = 10.0) -> None:
        super().__init__()
        in_pos, in_dir = embed_pos_rank * 6, embed_dir_rank * 6
        self.skips = [4] if skips is None else list(skips)
        self.activation_type, self.density_activation_type = activation_type, density_activation_type
        self.pe_pos = PositionalEncodingInfo(embed_pos_rank)
        self.pe_dir = PositionalEncodingInfo(embed_dir_rank)
        layers = [nn.Linear(in_pos, layer_width)]
        for layer_id in range(layer_count - 1):
            layers.append(nn.Linear(layer_width + (in_pos if layer_id in self.skips else 0), layer_width))
        self.layers = nn.ModuleList(layers)
        self.outL_density = nn.Linear(layer_width, 1)
        self.outL_color = nn.Sequential(nn.Linear(layer_width + in_dir, layer_width // 2), nn.ReLU(),
                                        nn.Linear(layer_width // 2, 3))
        self.layer_count, self.layer_width = layer_count, layer_width
        self.lowpass_alpha_offset = lowpass_alpha_offset
        self.lowpass_alpha = lowpass_alpha_offset

    def _descriptor(self) -> FieldDesc:
        d = FieldDesc()
        d.kind = FIELD_NERF
        d.embed_pos_rank, d.embed_dir_rank = self.pe_pos.embed_dim, self.pe_dir.embed_dim
        d.layer_count, d.layer_width = self.layer_count, self.layer_width
        d.n_skips = len(self.skips)
        for i, s in enumerate(self.skips[:8]):
            d.skips[i] = s
        d.activation, d.density_activation = ACT[self.activation_type], ACT[self.density_activation_type]
        return d

    def _tensors(self):
        mods = list(self.layers) + [self.outL_density, self.outL_color[0], self.outL_color[2]]
        return [m.weight for m in mods], [m.bias for m in mods]

    def _iter_state(self):
        return 1.1, 2.0, lowpass_scale(self.lowpass_alpha, self.pe_pos.embed_dim)

    def _train_engine_width(self) -> int:
        return TRAIN_ENGINE_WIDTH if self.layer_width <= TRAIN_ENGINE_WIDTH else 2 * TRAIN_ENGINE_WIDTH

    def _train_layout(self):
        W, E = self.layer_width, self._train_engine_width()
        if W >= E or W % 2:             # E itself needs no padding; wider than 512: the library refuses
            return None
        cpe, cdir = 6 * self.pe_pos.embed_dim, 6 * self.pe_dir.embed_dim
        hid, half, same = [(W, E)], [(W // 2, E // 2)], lambda n: [(n, n)]
        wl, bl = [], []
        for l in range(len(self.layers)):                   # nn.Linear [out, in]; cat([hx, embed_pos]) puts the hidden state first
            wide = l > 0 and (l - 1) in self.skips
            wl.append((hid, same(cpe) if l == 0 else (hid + same(cpe) if wide else hid)))
            bl.append((hid,))
        wl += [(same(1), hid), (half, hid + same(cdir)), (same(3), half)]
        bl += [None, (half,), None]
        return wl, bl

    def forward(self, sampling: Sampling) -> Dict[str, Tensor]:
        """density [B,S], color [B,S,3] (nerf.py:161-164); with autograd enabled and trainable parameters the outputs
        carry the graph (one node over the HIP forward / backward kernels), under torch.no_grad() the fused kernel runs."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .autograd import RadianceFieldFunction
            pos = sampling.sample_pos
            ctx = Context.get(pos.device)
            ws, bs = self._tensors()
            desc, ws, bs = self._train_tensors(ws, bs)
            self.upload(ctx, self._slot, weights=False, train=(desc, ws, bs) if self._train_layout() is not None else None)
            B, S = pos.shape[0], pos.shape[1]
            density, color = RadianceFieldFunction.apply(ctx, self._slot, self._iter_state(), len(ws), pos.detach(),
                                                         sampling.sample_dir.detach(), sampling.diag_variance.detach(), *ws, *bs)
            return {"density": density.view(B, S), "color": color.view(B, S, 3)}
        return self._run(sampling, OUT_MINIMAL, ("density", "color"))

    def set_iter(self, iter: int) -> None:
        self.lowpass_alpha = self.pe_pos.embed_dim if iter == -1 else self.lowpass_alpha_offset + 0.001 * iter


class NeuS(BaseNeuralField):
    """NeuS SDF field (neus.py:30-99 constructor keywords).  The reference takes the
    surface normal with torch.autograd.grad; here rendering takes it in reverse mode
    through the sdf trunk (the kernel of NeDDF's distance gradient, seeded with the
    unit vector of the sdf feature), training carries forward-mode Jacobian rows."""

    def __init__(self, embed_pos_rank: int = 6, embed_dir_rank: int = 4, sdf_layer_count: int = 8,
                 sdf_layer_width: int = 256, col_layer_count: int = 8, col_layer_width: int = 256,
                 activation_type: str = "ReLU", init_variance: float = 0.3, skips: Optional[List[int]] = None) -> None:
        super().__init__()
        in_sdf = embed_pos_rank * 6
        in_col = 6 + embed_dir_rank * 6 + sdf_layer_width
        self.skips = [4] if skips is None else list(skips)
        self.activation_type = activation_type
        self.activation = activation_type
        self.pe_pos = PositionalEncodingInfo(embed_pos_rank)
        self.pe_dir = PositionalEncodingInfo(embed_dir_rank)
        sdf = [nn.Linear(in_sdf, sdf_layer_width)]
        for layer_id in range(sdf_layer_count - 1):
            sdf.append(nn.Linear(sdf_layer_width + (in_sdf if layer_id in self.skips else 0), sdf_layer_width))
        col = [nn.Linear(in_col, col_layer_width)]
        col += [nn.Linear(col_layer_width, col_layer_width) for _ in range(col_layer_count - 1)]
        col.append(nn.Linear(col_layer_width, 3))
        self.layers_sdf = nn.ModuleList(sdf)
        self.layers_col = nn.ModuleList(col)
        self.variance = nn.Parameter(torch.tensor(init_variance))
        self.sdf_layer_count, self.sdf_layer_width = sdf_layer_count, sdf_layer_width
        self.col_layer_count, self.col_layer_width = col_layer_count, col_layer_width

    def _descriptor(self) -> FieldDesc:
        d = FieldDesc()
        d.kind = FIELD_NEUS
        d.embed_pos_rank, d.embed_dir_rank = self.pe_pos.embed_dim, self.pe_dir.embed_dim
        d.layer_count, d.layer_width = self.sdf_layer_count, self.sdf_layer_width
        d.col_layer_count, d.col_layer_width = self.col_layer_count, self.col_layer_width
        d.n_skips = len(self.skips)
        for i, s in enumerate(self.skips[:8]):
            d.skips[i] = s
        d.activation = d.density_activation = ACT[self.activation_type]
        return d

    def _tensors(self):
        mods = list(self.layers_sdf) + list(self.layers_col)
        dummy = torch.zeros(1)
        return [m.weight for m in mods] + [self.variance.reshape(1)], [m.bias for m in mods] + [dummy]

    def _train_engine_width(self) -> int:
        return TRAIN_ENGINE_WIDTH if max(self.sdf_layer_width, self.col_layer_width) <= TRAIN_ENGINE_WIDTH else 2 * TRAIN_ENGINE_WIDTH

    def _train_layout(self):
        Ws, Wc, E = self.sdf_layer_width, self.col_layer_width, self._train_engine_width()
        if (Ws >= E and Wc >= E) or Ws > E or Wc > E:     # both at E: no padding; wider than 512: the library refuses
            return None
        cpe, small = 6 * self.pe_pos.embed_dim, 6 + 6 * self.pe_dir.embed_dim
        hs, hc, same = [(Ws, E)], [(Wc, E)], lambda n: [(n, n)]
        wl, bl = [], []
        for l in range(len(self.layers_sdf)):               # nn.Linear [out, in]; cat([hx, embed_pos]): hidden state first
            wide = l > 0 and (l - 1) in self.skips
            wl.append((hs, same(cpe) if l == 0 else (hs + same(cpe) if wide else hs)))
            bl.append((hs,))
        n_col = len(self.layers_col)
        for l in range(n_col):                               # colour input [pos, embed_dir, gradients | sdf features] (neus.py:146-149)
            last = l == n_col - 1
            wl.append((same(3) if last else hc, same(small) + hs if l == 0 else hc))
            bl.append(None if last else (hc,))
        wl.append(None)                                      # variance
        bl.append(None)
        return wl, bl

    def upload(self, ctx: Context, slot: int, weights: bool = True, train=None) -> None:
        # `variance.reshape(1)` is a fresh view each call: key the upload on the parameter itself
        ws, bs = self._tensors() if train is None else (train[1], train[2])
        desc = self._descriptor() if train is None else train[0]
        desc.weight_dtype = DTYPE[self.weight_dtype]
        sig = (self._uid, slot, bytes(desc), tuple((t.data_ptr(), t._version) for t in ws[:-1] + bs[:-1]), self.variance.data_ptr(),
               self.variance._version)
        have = ctx.slot_owner.get(slot)
        if not weights and have is not None and have[:3] == sig[:3]:
            return          # training step: the kernels read the live parameter tensors, the slot only describes the architecture
        if have != sig:
            hw = [t.detach().to("cpu", torch.float32).contiguous() for t in ws]
            hb = [t.detach().to("cpu", torch.float32).contiguous() for t in bs]
            ctx.set_field(slot, desc, hw, hb, sig)

    def _iter_state(self):
        return 1.1, 2.0, [1.0] * self.pe_pos.embed_dim

    def forward(self, sampling: Sampling) -> Dict[str, Tensor]:
        """sdf, density [B,S]; color [B,S,3] (neus.py:157-161).  With autograd enabled and trainable parameters the outputs
        carry the graph (one node over the HIP forward / backward kernels); under torch.no_grad() the fused kernels run."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .autograd import SdfFieldFunction
            pos = sampling.sample_pos
            ctx = Context.get(pos.device)
            mods = list(self.layers_sdf) + list(self.layers_col)
            ws = [m.weight for m in mods] + [self.variance.reshape(1)]
            bs = [m.bias for m in mods] + [torch.zeros(1, device=pos.device)]
            desc, ws, bs = self._train_tensors(ws, bs)
            self.upload(ctx, self._slot, weights=False, train=(desc, ws, bs) if self._train_layout() is not None else None)
            B, S = pos.shape[0], pos.shape[1]
            sdf, density, color = SdfFieldFunction.apply(ctx, self._slot, len(ws), pos.detach(), sampling.sample_dir.detach(), *ws, *bs)
            return {"sdf": sdf.view(B, S), "density": density.view(B, S), "color": color.view(B, S, 3)}
        o = self._run(sampling, OUT_MINIMAL, ("distance", "density", "color"))
        return {"sdf": o["distance"], "density": o["density"], "color": o["color"]}

    def set_iter(self, iter: int) -> None:      # base_neuralfield.py:14-22: no warm-up state
        pass


# spellings used by BASELINE.json's north_star
NeDDFField = NeDDF
NeRFField = NeRF
