"""Rays and sampling containers -- host-side mirror of neddf/ray/{ray,sampling}.py.

The arithmetic (ray.py:88-194) runs in the HIP kernels of
csrc/render_kernels.hip; these classes only carry device tensors.
"""
from typing import Tuple

import torch
from torch import Tensor

from ._lib import Context


class Sampling:
    """Sample centre, ray direction and diagonal covariance per point (sampling.py:5-37)."""

    def __init__(self, sample_pos: Tensor, sample_dir: Tensor, diag_variance: Tensor) -> None:
        self.sample_pos = sample_pos
        self.sample_dir = sample_dir
        self.diag_variance = diag_variance

    @property
    def device(self) -> torch.device:
        return self.sample_pos.device

    def get_pe_weights(self, freq: Tensor) -> Tensor:
        """exp(-0.5 f^2 var) per (frequency, axis) -> [N, len(freq)*3] (sampling.py:44-71).

        Stand-alone op for API compatibility; the fused field kernels evaluate
        the same weights on chip and never call this."""
        embed_dim = int(freq.shape[0])
        expect = torch.tensor([2.0 ** t for t in range(embed_dim)])
        if not torch.equal(freq.detach().cpu().to(torch.float32), expect):
            raise ValueError("get_pe_weights: the HIP kernels assume the reference's frequencies 2**t")
        return Context.get(self.device).op_pe_weights(self.diag_variance, embed_dim)


class Ray:
    """Batch of rays (ray.py:23-50): ray_dir/ray_orig [B,3], uv [B,2]."""

    def __init__(self, ray_dir: Tensor, ray_orig: Tensor, uv: Tensor) -> None:
        self.single = ray_dir.dim() == 1
        assert ray_orig.shape == ray_dir.shape
        if self.single:
            assert uv.shape == (2,)
        else:
            assert uv.shape == (ray_orig.shape[0], 2)
        self.ray_dir = ray_dir
        self.ray_orig = ray_orig
        self.uv = uv

    @property
    def device(self) -> torch.device:
        return self.ray_dir.device

    def __len__(self) -> int:
        return 1 if self.single else self.ray_dir.shape[0]

    def __getitem__(self, item: int) -> Tuple[Tensor, Tensor]:
        if self.single:
            return (self.ray_dir, self.ray_orig)
        return (self.ray_dir[item, :], self.ray_orig[item, :])

    def _sample(self, dists: Tensor, ray_radius) -> Sampling:
        assert dists.shape[0] == self.ray_dir.shape[0]
        ctx = Context.get(self.device)
        pos, d, var = ctx.sampling(self.ray_dir, self.ray_orig, dists, ray_radius)
        return Sampling(pos, d, var)

    def get_sampling_points(self, dists: Tensor) -> Sampling:
        """pos = o + d*t, zero variance (ray.py:88-126)."""
        return self._sample(dists, None)

    def get_sampling_cones(self, dists: Tensor, ray_radius: float = 1e-3) -> Sampling:
        """mip-NeRF conical-frustum Gaussians (ray.py:128-194)."""
        return self._sample(dists, ray_radius)
