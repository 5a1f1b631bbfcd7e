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
    200 MB to ``dst``; a caching-allocator miss there is a hipMalloc in the middle of the pipeline).

    ``ring=R`` bounds that memory: R local slots per rank and R x world receive slots on ``dst`` instead of
    ``n_steps`` of each (200 steps x 8 ranks x 25 MB = 40 GB on rank 0 otherwise, beside the renderer's
    workspaces).  Step j uses slot j % R; before the slot is handed out again (``slot(j)``, i.e. before step
    j renders into it) step j - R is *retired*: its transfer is waited for and ``consumer(step, images)`` runs
    on the receiving rank -- default: one float64 checksum per image, kept in ``sums[n_steps, world]`` (a real
    caller writes the images out here, as the reference's evaluator does per rank,
    pgdvs/engines/evaluator_pgdvs.py:432-440).  ``finish()`` then returns ``{"sums": ..., "tail": ...}``: the
    checksums of all views and the images of the last min(R, n_steps) steps in view order (``None`` off ``dst``).
    R must cover the steps in flight: lanes + the host's run-ahead (+ a spare)."""

    def __init__(self, dst: int = 0, n_steps: int | None = None, like: torch.Tensor | None = None, ring: int | None = None,
                 consumer=None):
        self.dst = dst
        self.works, self.bufs, self.keep = [], [], []
        self.on = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.world = dist.get_world_size() if self.on else 1
        self.is_dst = (not self.on) or dist.get_rank() == dst
        self.local = self.stack = None
        self.ring = int(ring) if ring else None
        self.n_steps = n_steps
        self.consumer = consumer
        self.events, self.retired, self.sums = [], 0, None
        self.capacity_steps = n_steps
        if self.ring:
            assert n_steps and like is not None, "ring mode needs n_steps and like"
            self.ring = max(1, min(self.ring, n_steps))
            self.local = torch.empty((self.ring,) + tuple(like.shape), dtype=like.dtype, device=like.device)
            if self.is_dst:
                self.stack = (torch.empty((self.ring, self.world) + tuple(like.shape), dtype=like.dtype, device=like.device)
                              if self.on else None)
                self.sums = torch.empty((n_steps, self.world), dtype=torch.float64, device=like.device)  # every used row is written
        elif n_steps and like is not None:
            self.local = torch.empty((n_steps,) + tuple(like.shape), dtype=like.dtype, device=like.device)
            if self.on and dist.get_rank() == dst:
                self.stack = torch.empty((n_steps, dist.get_world_size()) + tuple(like.shape), dtype=like.dtype, device=like.device)

    # -- ring mode -------------------------------------------------------------
    def reset(self, n_steps: int) -> "AsyncImageGather":
        """Start another run of ``n_steps`` steps on the SAME buffers (ring mode; ``n_steps`` at most what the gather
        was created for): nothing is allocated or freed between runs.  (bench.py: with captured HIP graphs alive, a
        fresh allocation between two replay loops was followed by a GPU memory fault on this ROCm.)"""
        assert self.ring and self.retired == len(self.keep), "reset() needs a finished ring-mode gather"
        assert n_steps <= self.capacity_steps, (n_steps, self.capacity_steps)
        self.n_steps = n_steps
        self.works, self.bufs, self.keep, self.events, self.retired = [], [], [], [], 0
        return self

    def _received(self, i: int) -> torch.Tensor:
        """[world, ...] images of step i as they sit in the ring (on ``dst``)"""
        return self.stack[i % self.ring] if self.on else self.local[i % self.ring][None]

    def _retire(self, i: int) -> None:
        """step i's slot is about to be reused: wait for its transfer (and, single rank, for its render) on the
        current stream and consume the images there"""
        if self.on:
            self.works[i].wait()
        elif self.events[i] is not None:
            torch.cuda.current_stream().wait_event(self.events[i])
        if self.is_dst:
            imgs = self._received(i)
            if self.consumer is not None:
                self.consumer(i, imgs)
            else:
                self.sums[i] = self.checksum(imgs)
        self.retired = i + 1

    def checksum(self, imgs: torch.Tensor) -> torch.Tensor:
        """[world] float64: the default consumer's proof of arrival -- head, tail and a strided sample of every image
        (one small kernel; summing all of it would cost rank 0 a 25 MB read per view and rank, 1.6 TB/s at 8 ranks x 1000
        views/s, for a benchmark-side stub: real consumers write the image out)"""
        flat = imgs.reshape(self.world, -1)
        n = flat.shape[1]
        if n <= 3 * 4096:
            return flat.sum(dim=1, dtype=torch.float64)
        return torch.cat([flat[:, :4096], flat[:, -4096:], flat[:, 4096:-4096:4099]], dim=1).sum(dim=1, dtype=torch.float64)

    def slot(self, j: int | None = None):
        """The preallocated local buffer of step ``j`` (default: the next one), or None.  A renderer that
        writes its image straight into it (``data["_combined_rgb_out"]``) makes ``submit`` copy-free.
        (Ring mode: retires the step that held the slot; call it on the stream the step's own stream will wait for.)"""
        j = len(self.keep) if j is None else j
        if self.ring:
            while self.retired <= j - self.ring:
                self._retire(self.retired)
            return self.local[j % self.ring]
        return self.local[j] if (self.local is not None and j < self.local.shape[0]) else None

    def submit(self, img: torch.Tensor) -> None:
        j = len(self.keep)
        if self.ring:
            assert j < self.n_steps, "more steps than announced"
            while self.retired <= j - self.ring:  # (slot() not used by the caller)
                self._retire(self.retired)
            mine = self.local[j % self.ring]
            if img.data_ptr() != mine.data_ptr():
                mine.copy_(img)
            self.keep.append(None)
            if self.on:
                bufs = list(self.stack[j % self.ring].unbind(0)) if self.is_dst else None
                self.works.append(dist.gather(mine, bufs, dst=self.dst, async_op=True))
            else:
                ev = None
                if mine.is_cuda:
                    ev = torch.cuda.Event()
                    ev.record()
                self.events.append(ev)
            return
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

    def finish(self, tail: bool = True):
        """wait for everything; ring mode: ``tail=False`` skips handing back the images still in the ring (a copy of up to R x
        world images: bench.py only wants the checksums inside its timed region)"""
        n = len(self.keep)
        want_tail = tail
        if self.ring:
            first_tail = max(0, n - self.ring)
            tail = None
            if self.is_dst and n and want_tail:
                # the last steps are still in the ring: hand their images back in view order, then retire them
                for i in range(first_tail, n):
                    if self.on:
                        self.works[i].wait()
                    elif self.events[i] is not None:
                        torch.cuda.current_stream().wait_event(self.events[i])
                tail = torch.cat([self._received(i).flatten(0, 1) for i in range(first_tail, n)], 0)
            while self.retired < n:
                self._retire(self.retired)
            # (a copy: reset() reuses the buffer)
            return {"sums": self.sums[:n].clone(), "tail": tail, "tail_first_step": first_tail} if self.is_dst else None
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
