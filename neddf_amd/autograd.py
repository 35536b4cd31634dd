"""torch.autograd glue of the training step.

The reference trains through torch autograd over its with_grad modules
(neddf/nn_module/with_grad/*.py carry hand-written backward passes for the
(value, Jacobian) pairs) and plain torch ops for the volume integral
(base_neural_render.py:148-171).  Here both are single autograd nodes whose
forward and backward are HIP kernels behind the C ABI
(neddf_train_field_forward / _backward, neddf_composite / _composite_backward);
torch only carries the small per-ray tensors between them and into the losses.
"""
from typing import Tuple

import torch
from torch import Tensor

from ._lib import Context


class FieldFunction(torch.autograd.Function):
    """NeDDF.forward (neddf.py:162-309) on [N,3] samples -> distance, density, color, fields_penalty, aux_grad.

    Gradients flow to the parameters only: sample positions come out of the
    (non-differentiable) sampler, exactly as in the reference's render_rays."""

    @staticmethod
    def forward(ctx, hip: Context, slot: int, iter_state, n_tensors: int, pos: Tensor, dir: Tensor, var: Tensor,
                *params: Tensor) -> Tuple[Tensor, ...]:
        weights = [p.detach() for p in params[:n_tensors]]
        biases = [p.detach() for p in params[n_tensors:]]
        hip.set_iter(slot, *iter_state)
        ws, distance, density, color, penalty, aux = hip.train_field_forward(slot, weights, biases, pos, dir, var)
        ctx.hip, ctx.slot, ctx.iter_state, ctx.n_tensors, ctx.n_points = hip, slot, iter_state, n_tensors, distance.shape[0]
        ctx.save_for_backward(ws, *params)
        return distance, density, color, penalty, aux

    @staticmethod
    def backward(ctx, g_distance, g_density, g_color, g_penalty, g_aux):
        ws, *params = ctx.saved_tensors
        n = ctx.n_tensors
        weights = [p.detach() for p in params[:n]]
        biases = [p.detach() for p in params[n:]]
        ctx.hip.set_iter(ctx.slot, *ctx.iter_state)
        with torch.cuda.device(ws.device):
            gw, gb = ctx.hip.train_field_backward(ctx.slot, weights, biases, ctx.n_points, ws, g_distance, g_density, g_color,
                                                  g_penalty, g_aux)
        return (None,) * 7 + tuple(gw) + tuple(gb)


class RadianceFieldFunction(torch.autograd.Function):
    """NeRF.forward (nerf.py:107-165) on [N,3] samples -> density, color.  The reference differentiates this network
    with plain torch autograd over nn.Linear; here forward and backward are the same HIP building blocks as NeDDF's
    (value rows only), one autograd node for the whole field."""

    @staticmethod
    def forward(ctx, hip: Context, slot: int, iter_state, n_tensors: int, pos: Tensor, dir: Tensor, var: Tensor,
                *params: Tensor) -> Tuple[Tensor, Tensor]:
        weights = [p.detach() for p in params[:n_tensors]]
        biases = [p.detach() for p in params[n_tensors:]]
        hip.set_iter(slot, *iter_state)
        ws, _, density, color, _, _ = hip.train_field_forward(slot, weights, biases, pos, dir, var, radiance_only=True)
        ctx.hip, ctx.slot, ctx.iter_state, ctx.n_tensors, ctx.n_points = hip, slot, iter_state, n_tensors, density.shape[0]
        ctx.save_for_backward(ws, *params)
        return density, color

    @staticmethod
    def backward(ctx, g_density, g_color):
        ws, *params = ctx.saved_tensors
        n = ctx.n_tensors
        weights = [p.detach() for p in params[:n]]
        biases = [p.detach() for p in params[n:]]
        ctx.hip.set_iter(ctx.slot, *ctx.iter_state)
        with torch.cuda.device(ws.device):
            gw, gb = ctx.hip.train_field_backward(ctx.slot, weights, biases, ctx.n_points, ws, None, g_density, g_color, None, None)
        return (None,) * 7 + tuple(gw) + tuple(gb)


class SdfFieldFunction(torch.autograd.Function):
    """NeuS.forward (neus.py:101-162) on [N,3] samples -> sdf, density, color.  The reference takes the normal with
    torch.autograd.grad(create_graph=True), so its parameter gradients are a double backward; here the normal is the
    Jacobian rows of the sdf trunk and the backward is the ordinary reverse pass over (value, Jacobian) rows.  `params` =
    weights (..., variance as a 1-element tensor) followed by biases (..., an unused 1-element tensor)."""

    @staticmethod
    def forward(ctx, hip: Context, slot: int, n_tensors: int, pos: Tensor, dir: Tensor, *params: Tensor) -> Tuple[Tensor, ...]:
        weights = [p.detach() for p in params[:n_tensors]]
        biases = [p.detach() for p in params[n_tensors:]]
        ws, sdf, density, color, _, _ = hip.train_field_forward(slot, weights, biases, pos, dir, torch.zeros_like(pos),
                                                                radiance_only=False, sdf=True)
        ctx.hip, ctx.slot, ctx.n_tensors, ctx.n_points = hip, slot, n_tensors, sdf.shape[0]
        ctx.save_for_backward(ws, *params)
        return sdf, density, color

    @staticmethod
    def backward(ctx, g_sdf, g_density, g_color):
        ws, *params = ctx.saved_tensors
        n = ctx.n_tensors
        weights = [p.detach() for p in params[:n]]
        biases = [p.detach() for p in params[n:]]
        with torch.cuda.device(ws.device):
            gw, gb = ctx.hip.train_field_backward(ctx.slot, weights, biases, ctx.n_points, ws, g_sdf, g_density, g_color, None, None)
        return (None,) * 5 + tuple(gw) + tuple(gb[:-1]) + (None,)


class CompositeFunction(torch.autograd.Function):
    """integrate_volume_render (base_neural_render.py:117-172) -> weight, depth, color, transmittance."""

    @staticmethod
    def forward(ctx, hip: Context, max_dist: float, dists: Tensor, densities: Tensor, colors: Tensor):
        out, flag = hip.composite(dists, densities, colors, max_dist)
        ctx.hip, ctx.max_dist = hip, max_dist
        ctx.save_for_backward(dists, densities, colors)
        ctx.mark_non_differentiable(flag)
        return out["weight"], out["depth"], out["color"], out["transmittance"], flag

    @staticmethod
    def backward(ctx, g_weight, g_depth, g_color, g_trans, _g_flag):
        dists, densities, colors = ctx.saved_tensors
        with torch.cuda.device(dists.device):
            g_density, g_point_color = ctx.hip.composite_backward(dists, densities, colors, ctx.max_dist, g_weight, g_depth,
                                                                  g_color, g_trans)
        return None, None, None, g_density, g_point_color
