"""Ray-parallel rendering over the GPUs of one node.

Rays are independent on the eval path (nerf_render.py:128-188 has no cross-ray
term), so the flat pixel index is cut into contiguous slabs, one per rank, the
packed weights (2.6 MB) are replicated, and the only exchange is one gather of
the rendered pixels: colour(3) + depth(1) + transmittance(1) = 20 B/ray, packed
into ONE [n,5] tensor so a view costs one RCCL all-gather (12.8 MB at 800x800)
over xGMI, issued by the HIP library itself (neddf_gather_pixels: its own
communicator, its own stream -- the pixels of view i travel while view i+1
renders).  No all-reduce anywhere.  The reference has no multi-GPU code; this
module is new.

Training (also new: the reference trains on one device) is data-parallel over
rays: every rank runs the training step on its own random pixel batch with
replicated parameters, and the one exchange per step is a single all-reduce of
the flattened gradients (2.6 MB for the shipped NeDDF) -- one bucket, because
with 26 small tensors the latency of xGMI's point-to-point ring dominates, not
its bandwidth.

One process per GPU (torch.distributed, backend "nccl" = RCCL on ROCm; "gloo"
on CPU for the tests).
"""
from typing import Dict, Iterable, Tuple

import torch
import torch.distributed as dist
from torch import Tensor

CHANNELS = {"color": 3, "depth": 1, "transmittance": 1}


def shard_range(n: int, rank: int, world: int, granule: int = 1) -> Tuple[int, int]:
    """Contiguous slab [lo, hi) of range(n) owned by `rank`, cut on multiples of `granule` (the last granule of the range may
    be short): granule counts differ by at most one (neddf_shard_range_granular of the C ABI).  A sharded frame uses
    render_image's `chunk` as the granule, so that no chunk is split between two ranks: the reference decides sample_pdf's
    NaN fallback per chunk (base_neural_render.py:105-114), and a rank that held only part of one would decide it on
    different rays."""
    units = -(-n // granule)
    base, rem = divmod(units, world)
    ul = rank * base + min(rank, rem)
    uh = ul + base + (1 if rank < rem else 0)
    return min(n, ul * granule), min(n, uh * granule)


def pack_pixels(parts: Dict[str, Tensor], keys: Iterable[str]) -> Tensor:
    # (explicit channel counts: a rank's slab may be empty -- more ranks than chunks -- and reshape(0, -1) is ambiguous)
    return torch.cat([parts[k].reshape(parts[k].shape[0], CHANNELS[k]) for k in keys], dim=1).contiguous()


def unpack_pixels(packed: Tensor, keys: Iterable[str]) -> Dict[str, Tensor]:
    out, c = {}, 0
    for k in keys:
        out[k] = packed[:, c:c + CHANNELS[k]]
        c += CHANNELS[k]
    return out


def native_comm(ctx, group=None):
    """Make sure `ctx` (the HIP library's context of this rank's device) owns an RCCL communicator spanning `group`:
    rank 0 creates the unique id (neddf_comm_unique_id), torch.distributed only carries its 128 bytes to the other
    ranks (bootstrap), every rank joins (neddf_comm_init).  Returns ctx.comm_info()."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    info = ctx.comm_info()
    if info["nranks"] == world and info["rank"] == rank:
        return info
    if info["nranks"]:
        ctx.comm_destroy()
    # the id travels as a byte tensor: on the device for an RCCL process group, on the host for gloo.  Byte 128 says whether
    # rank 0 could make one, and the join ends with a vote, so that a failure raises on EVERY rank (a caller with a second
    # route -- bench.py falls back to torch.distributed's all-gather -- must take it on all ranks or on none)
    on_device = "nccl" in str(dist.get_backend(group))
    side = ctx.device if on_device else "cpu"
    box = torch.zeros(129, dtype=torch.uint8, device=side)
    err = None
    if rank == 0:
        try:
            staged = torch.zeros(129, dtype=torch.uint8)
            staged[:128] = torch.frombuffer(bytearray(ctx.comm_unique_id()), dtype=torch.uint8)
            staged[128] = 1
            box.copy_(staged)
        except Exception as e:              # no RCCL to load, ...: tell the others instead of leaving them in the broadcast
            err = e
    dist.broadcast(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    host = box.cpu()
    if int(host[128]) != 1:
        raise RuntimeError("rank 0 could not create an RCCL unique id%s" % (": %s" % err if err is not None else ""))
    try:
        ctx.comm_init(rank, world, bytes(host[:128].numpy().tobytes()))
    except Exception as e:
        err = e
    vote = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=side)
    dist.all_reduce(vote, op=dist.ReduceOp.MIN, group=group)
    if int(vote.item()) != 1:
        if err is None:
            ctx.comm_destroy()
        raise RuntimeError("neddf_comm_init failed on %s" % ("this rank: %s" % err if err is not None else "another rank"))
    return ctx.comm_info()


class PixelGather:
    """Handle of an all-gather in flight on the library's communication stream (gather_pixels(..., wait=False)).
    .wait() makes the current stream wait for it and returns the [n_total, C] tensor."""

    def __init__(self, ctx, out: Tensor) -> None:
        self.ctx, self.out = ctx, out

    def wait(self) -> Tensor:
        if self.ctx is not None:
            self.ctx.comm_wait()
            self.ctx = None
        return self.out


def gather_pixels(local: Tensor, n_total: int, group=None, force_collective: bool = False, wait: bool = True, out: Tensor = None,
                  granule: int = 1):
    """All-gather per-rank slabs [n_rank, C] (shard_range(..., granule) order) into [n_total, C] on every rank.

    Device tensors go through the HIP library's own RCCL communicator (neddf_gather_pixels): the collective runs on a
    communication stream ordered after the current stream, so with wait=False the caller can keep rendering the next
    view while the pixels travel and collect them later (returns a PixelGather).  Host tensors (the gloo tests) use
    torch.distributed.  force_collective runs the collective even for a single rank."""
    world = dist.get_world_size(group)
    if world == 1 and not force_collective:
        return local if wait else PixelGather(None, local)
    if local.is_cuda and "nccl" not in str(dist.get_backend(group)):
        # a process group without a device backend (gloo: several test ranks sharing one GPU, where RCCL cannot form a
        # communicator): stage the slab through the host
        full = gather_pixels(local.cpu(), n_total, group, force_collective, granule=granule).to(local.device)
        if out is not None:
            out.copy_(full)
            full = out
        return full if wait else PixelGather(None, full)
    if local.is_cuda:
        from ._lib import Context
        ctx = Context.get(local.device)
        native_comm(ctx, group)
        if ctx._gather_refs is not None:       # one gather in flight per context: its buffers are released by the wait
            ctx.comm_wait()
        res = ctx.gather_pixels(local, n_total, out, granule)
        handle = PixelGather(ctx, res)
        return handle.wait() if wait else handle
    sizes = [shard_range(n_total, r, world, granule) for r in range(world)]
    pad = max(hi - lo for lo, hi in sizes)
    ragged = not all(hi - lo == pad for lo, hi in sizes)
    if ragged and _host_route_in_place(group):
        # the library's opt-in route restated for host tensors: rank q's slab straight to its offset of every rank's buffer
        rank = dist.get_rank(group)
        full = local.new_empty(n_total, local.shape[1])
        full[sizes[rank][0]:sizes[rank][1]] = local
        for q, (lo, hi) in enumerate(sizes):
            if hi > lo:                     # more ranks than chunks: that rank contributes nothing
                dist.broadcast(full[lo:hi], src=dist.get_global_rank(group, q) if group is not None else q, group=group)
        return full if wait else PixelGather(None, full)
    buf = local
    if local.shape[0] < pad:
        buf = torch.cat([local, local.new_zeros(pad - local.shape[0], local.shape[1])])
    full = local.new_empty(world * pad, local.shape[1])
    dist.all_gather_into_tensor(full, buf.contiguous(), group=group)
    if ragged:
        full = torch.cat([full[r * pad:r * pad + (hi - lo)] for r, (lo, hi) in enumerate(sizes)])
    return full if wait else PixelGather(None, full)


_HOST_ROUTE = {}


def _host_route_in_place(group=None) -> bool:
    """The host-tensor twin of neddf_comm_init's agreement (comm_capi.hip): the route of a ragged gather is a property of the
    process group, agreed ONCE by every rank -- each contributes its NEDDF_GATHER_INPLACE wish, the in-place route (one broadcast
    per slab) is taken only if all of them asked for it, otherwise all take the padded route.  A per-rank choice would issue
    mismatched collectives."""
    key = id(group) if group is not None else None
    if key not in _HOST_ROUTE:
        import os
        wish = torch.tensor([1 if os.environ.get("NEDDF_GATHER_INPLACE", "0").strip() not in ("", "0") else 0], dtype=torch.int32)
        dist.all_reduce(wish, op=dist.ReduceOp.MIN, group=group)
        _HOST_ROUTE[key] = bool(wish.item())
    return _HOST_ROUTE[key]


def render_image_sharded(render, width: int, height: int, camera, target_types: Iterable[str], downsampling: int = 1,
                         chunk: int = 512, group=None) -> Dict[str, Tensor]:
    """NeRFRender.render_image with the pixel range split over the ranks of
    `group`; every rank returns the full [h, w, C] images.  With the default
    "torch_cpu" RNG every rank must hold the same torch seed: each rank jumps the
    generator to its slab's position in the reference's draw order (rng.py) and
    draws only its own uniforms, which makes the image independent of the world
    size and the host cost proportional to the slab."""
    keys = list(target_types)
    w, h = width // downsampling, height // downsampling
    n = w * h
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(n, rank, world, chunk)          # whole chunks per rank: the image is the single-GPU one bit for bit,
    parts = render.render_image(width, height, camera, keys, downsampling, chunk, pixel_range=(lo, hi))    # NaN fallback included
    full = gather_pixels(pack_pixels(parts, keys), n, group, granule=chunk)
    return {k: v.reshape(h, w, -1) for k, v in unpack_pixels(full, keys).items()}


def average_gradients(params: Iterable[Tensor], group=None, force_collective: bool = False) -> None:
    """Replace every parameter's .grad by its mean over the ranks with ONE all-reduce of the flattened gradients.
    Parameters without a gradient on this rank contribute zeros (all ranks must hold the same parameter list)."""
    if not dist.is_available() or not dist.is_initialized():
        return
    world = dist.get_world_size(group)
    if world == 1 and not force_collective:
        return
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(world)
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n


def sync_parameters(tensors: Iterable[Tensor], group=None, src: int = 0) -> None:
    """Data-parallel start: every rank takes rank `src`'s values (one broadcast of the flattened parameters and buffers),
    so that the replicas the averaged gradients are applied to are the SAME point in parameter space."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    tensors = list(tensors)
    if not tensors:
        return
    with torch.no_grad():
        flat = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in tensors])
        dist.broadcast(flat, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
        off = 0
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t).to(t.dtype))
            off += n


def assert_replicas_identical(tensors: Iterable[Tensor], group=None, what: str = "parameters") -> None:
    """Raise on every rank unless all ranks hold bit-identical values: a (sum, sum of squares, xor of the bit patterns)
    signature per rank, compared by min / max all-reduces."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    tensors = list(tensors)
    if not tensors:
        return
    with torch.no_grad():
        flat = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in tensors])
        bits = flat.view(torch.int32).to(torch.int64)
        # the bit-pattern checksums stay int64 through the collective (a double resolves only ~2^7 at their magnitude of 2^60: a
        # one-ulp divergence of one parameter would be rounded away); two independent weightings so that no single swap cancels
        idx = torch.arange(bits.numel(), device=bits.device)
        isig = torch.stack([(bits * (idx % 8191 + 1)).sum(), (bits * ((idx * 2654435761) % 65521 + 1)).sum(), bits.sum()])
        sig = torch.stack([flat.double().sum(), flat.double().square().sum()])        # for the error message only
        ilo, ihi, lo, hi = isig.clone(), isig.clone(), sig.clone(), sig.clone()
        for t, op in ((ilo, dist.ReduceOp.MIN), (ihi, dist.ReduceOp.MAX), (lo, dist.ReduceOp.MIN), (hi, dist.ReduceOp.MAX)):
            dist.all_reduce(t, op=op, group=group)
    if not torch.equal(ilo, ihi):
        raise RuntimeError("data-parallel replicas hold different %s (bit-pattern checksums differ; spread of sum / sum of squares %s)" % (what, (hi - lo).tolist()))
    return
