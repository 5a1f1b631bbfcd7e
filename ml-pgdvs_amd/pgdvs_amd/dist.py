"""Multi-GPU driver pieces: target views are independent units, so a video shards
embarrassingly across ranks (one process per GPU).  Partitioning follows the reference's
``DistributedSampler(shuffle=False)`` use (pgdvs/engines/trainer_pgdvs.py:290-306): view
``v`` goes to rank ``v % world`` and the tail is padded by wrap-around.  The only data-path
collective is the gather of the final image stack to rank 0 (the reference writes PNGs per
rank instead, pgdvs/engines/evaluator_pgdvs.py:432-440); on ROCm ``backend="nccl"`` is
RCCL over xGMI.  With the ``gloo`` backend the same code runs on CPU tensors (tests)."""
from __future__ import annotations

import math

import torch
import torch.distributed as dist


def shard_indices(n_views: int, rank: int, world: int) -> list[int]:
    """Indices of the views rendered by ``rank`` (DistributedSampler, shuffle=False, drop_last=False)."""
    if n_views <= 0:
        return []
    total = math.ceil(n_views / world) * world
    idx = list(range(n_views))
    pad = total - n_views
    if pad:
        idx += (idx * math.ceil(pad / len(idx)))[:pad]
    return idx[rank:total:world]


def gather_image_stack(local: torch.Tensor, n_views: int, dst: int = 0):
    """local[n_local,3,H,W] on every rank -> on ``dst`` the stack [n_views,3,H,W] in view
    order (wrap-around duplicates dropped), ``None`` elsewhere."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local[:n_views]
    world, rank = dist.get_world_size(), dist.get_rank()
    bufs = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local.contiguous(), bufs, dst=dst)
    if rank != dst:
        return None
    n_local = local.shape[0]
    out = torch.empty((n_local * world,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        out[r::world] = bufs[r]
    return out[:n_views]


def reduce_metrics(values: torch.Tensor, dst: int = 0) -> torch.Tensor:
    """One packed SUM-reduce for all scalar metrics of a step (the reference issues one
    ``torch.distributed.reduce`` per key, pgdvs/engines/evaluator_pgdvs.py:183-186)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(values, dst=dst, op=dist.ReduceOp.SUM)
    return values


class AsyncImageGather:
    """Per-step asynchronous gather of the rendered images to ``dst``: step j's transfer (RCCL
    over xGMI, its own stream) overlaps with the rendering of step j+1, so only the last image
    is exposed.  ``finish()`` waits for all transfers and returns, on ``dst``, the stack
    [n_steps * world, ...] in view order (view = step * world + rank), else None.

    With ``n_steps`` and ``like`` (one image) every buffer is allocated once, up front: each rank
    copies its image into a slice of a local stack (the renderer's own output block -- which the
    image shares with the renderer's other outputs -- is free again right after the step), and
    ``dst`` receives every step directly into the slices of one receive stack.  Nothing is
    allocated or concatenated while steps are in flight (at 1080p and 8 ranks a step brings
    200 MB to ``dst``; a caching-allocator miss there is a hipMalloc in the middle of the pipeline)."""

    def __init__(self, dst: int = 0, n_steps: int | None = None, like: torch.Tensor | None = None):
        self.dst = dst
        self.works, self.bufs, self.keep = [], [], []
        self.on = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.local = self.stack = None
        if n_steps and like is not None:
            self.local = torch.empty((n_steps,) + tuple(like.shape), dtype=like.dtype, device=like.device)
            if self.on and dist.get_rank() == dst:
                self.stack = torch.empty((n_steps, dist.get_world_size()) + tuple(like.shape), dtype=like.dtype, device=like.device)

    def slot(self, j: int | None = None):
        """The preallocated local buffer of step ``j`` (default: the next one), or None.  A renderer that
        writes its image straight into it (``data["_combined_rgb_out"]``) makes ``submit`` copy-free."""
        j = len(self.keep) if j is None else j
        return self.local[j] if (self.local is not None and j < self.local.shape[0]) else None

    def submit(self, img: torch.Tensor) -> None:
        j = len(self.keep)
        if self.local is not None and j < self.local.shape[0] and tuple(img.shape) == tuple(self.local.shape[1:]):
            if img.data_ptr() != self.local[j].data_ptr():  # not rendered in place
                self.local[j].copy_(img)  # on the caller's current stream, like the gather below
            img = self.local[j]
        else:
            img = img.contiguous()
        self.keep.append(img)
        if not self.on:
            return
        bufs = None
        if dist.get_rank() == self.dst:
            if self.stack is not None and j < self.stack.shape[0] and tuple(img.shape) == tuple(self.stack.shape[2:]):
                bufs = list(self.stack[j].unbind(0))
            else:
                bufs = [torch.empty_like(img) for _ in range(dist.get_world_size())]
        self.bufs.append(bufs)
        self.works.append(dist.gather(img, bufs, dst=self.dst, async_op=True))

    def finish(self):
        n = len(self.keep)
        if not self.on:
            if not self.keep:
                return None
            if self.local is not None and n <= self.local.shape[0] and all(
                    k.data_ptr() == self.local[j].data_ptr() for j, k in enumerate(self.keep)):
                return self.local[:n].flatten(0, 1)
            return torch.cat(self.keep, 0)
        for w in self.works:
            w.wait()
        if dist.get_rank() != self.dst:
            return None
        if self.stack is not None and n <= self.stack.shape[0] and all(
                b[0].data_ptr() == self.stack[j, 0].data_ptr() for j, b in enumerate(self.bufs)):
            return self.stack[:n].flatten(0, 2)  # [step, rank, 1-or-more images, ...] -> view order
        return torch.cat([torch.cat(b, 0) for b in self.bufs], 0)
