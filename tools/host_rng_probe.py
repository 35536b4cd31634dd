#!/usr/bin/env python3
"""Host cost of the reference-compatible ("torch_cpu") uniforms of one 800x800 hierarchical frame, whole and sharded.

The reference draws [b, 65] then [b, 129] floats per chunk of 1024 rays on torch's CPU generator (nerf_render.py:137,
base_neural_render.py:75); a rank that renders a slab jumps the generator to its slab (neddf_amd/rng.py) and draws only its
own rows.  CPU only; prints one JSON object.

    python tools/host_rng_probe.py
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neddf_amd import rng  # noqa: E402
from neddf_amd.parallel import shard_range  # noqa: E402

N, CHUNK, PER_RAY = 800 * 800, 1024, 65 + 129


def draw(lo, hi):
    first = (lo // CHUNK) * CHUNK
    rng.skip_uniforms(first * PER_RAY)
    below = first
    while below < hi:
        b = min(N, below + CHUNK) - below
        torch.rand(b, 65); torch.rand(b, 129)
        below += b
    rng.skip_uniforms((N - below) * PER_RAY)


res = {"frame": "800x800, 65 + 129 uniforms per ray, chunk 1024 (124 160 000 floats)", "threads": torch.get_num_threads()}
torch.manual_seed(0)
t0 = time.perf_counter(); rng._char_poly(); rng._reduction_table(); res["one_time_setup_s"] = time.perf_counter() - t0
for world in (1, 2, 4, 8):
    times = []
    for rank in (0, world - 1):
        lo, hi = shard_range(N, rank, world)
        torch.manual_seed(0)
        draw(lo, hi)                       # first call computes the two jump polynomials of this geometry
        torch.manual_seed(0)
        t0 = time.perf_counter(); draw(lo, hi); times.append(time.perf_counter() - t0)
    res["world_%d_seconds_per_frame_per_rank" % world] = max(times)
torch.manual_seed(0)
t0 = time.perf_counter(); rng.skip_uniforms(N * PER_RAY - 1); res["one_jump_cached_polynomial_s"] = time.perf_counter() - t0
print(json.dumps(res, indent=1))
