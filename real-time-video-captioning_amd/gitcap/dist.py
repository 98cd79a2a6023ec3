"""Clip sharding across the GPUs of one node (one process per GPU) and the single collective of
the path: a gather of the caption ids.

The reference has no inference-time multi-GPU at all; it loops clips independently
(/root/reference/src/models/model.py:765), which is what makes the path shard embarrassingly:
rank r of R takes clips [r*B/R, (r+1)*B/R), holds a full weight replica, and nothing is exchanged
during encode/decode.  Only the final ids (int64 [B/R, max_len+1], ~2.7 KB per rank at B/R = 16)
are all-gathered, rank-major, so output row i is global clip i.  Backend "nccl" is RCCL on ROCm
(xGMI between the GPUs of a node); "gloo" runs the same code on CPU tensors for tests.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_range(n_clips: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced partition: the first (n_clips % world) ranks take one extra clip."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    q, r = divmod(n_clips, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_captions(ids: torch.Tensor, n_clips: int | None = None, group=None) -> torch.Tensor:
    """All ranks receive the ids of all clips, rows in global clip order.  `ids` is this rank's
    [b_local, L] int64 block; ragged shards (n_clips % world != 0) are padded for the collective
    and trimmed afterwards."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return ids
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    L = ids.shape[1]
    if n_clips is None:
        n_clips = ids.shape[0] * world
    sizes = [shard_range(n_clips, r, world) for r in range(world)]
    bmax = max(hi - lo for lo, hi in sizes)
    if ids.shape[0] != sizes[rank][1] - sizes[rank][0]:
        raise ValueError("local block does not match shard_range")
    pad = ids
    if ids.shape[0] < bmax:
        pad = torch.cat([ids, ids.new_zeros((bmax - ids.shape[0], L))], 0)
    pad = pad.contiguous()
    if dist.get_backend(group) == "gloo":
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
        out = torch.cat(parts, 0)
    else:
        out = torch.empty((world * bmax, L), dtype=ids.dtype, device=ids.device)
        dist.all_gather_into_tensor(out, pad, group=group)
    if all(hi - lo == bmax for lo, hi in sizes):
        return out
    return torch.cat([out[r * bmax: r * bmax + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], 0)


class CaptionGatherRing:
    """Asynchronous form of gather_captions for a stream of batches (what bench.py --gpus N runs every step).

    Every batch's ids are all-gathered with ``async_op=True`` into one of ``nbuf`` output buffers and joined a few
    batches later: issued synchronously the collective would sit, stream-ordered, in front of every later submission
    while its kernel waits for a free CU on a saturated GPU.  ``push`` returns the output buffer (valid after the
    returned work has been waited for, ``join=True`` waits at once); a buffer is reused only after the gather that
    last wrote it has been waited for.  ``fence`` waits for every gather issued so far (then barrier).
    Ragged shards are padded to the largest shard for the collective and trimmed in ``rows``.
    """

    def __init__(self, n_clips: int, L: int, device, nbuf: int = 8, group=None, dtype=torch.int64):
        if not dist.is_initialized():
            raise RuntimeError("CaptionGatherRing needs an initialised process group")
        self.group, self.world, self.rank = group, dist.get_world_size(group), dist.get_rank(group)
        self.sizes = [shard_range(n_clips, r, self.world) for r in range(self.world)]
        self.bmax = max(hi - lo for lo, hi in self.sizes)
        self.n_clips, self.L, self.nbuf = n_clips, L, nbuf
        self.gloo = dist.get_backend(group) == "gloo"
        self.bufs = [torch.empty((self.world * self.bmax, L), dtype=dtype, device=device) for _ in range(nbuf)]
        # one entry per buffer: (work, local ids kept alive) of the gather that last wrote it, or None.  Bounded: a
        # long-running captioner pushes for ever and only the last nbuf gathers can still be in flight.
        self.works = [None] * nbuf
        self.pushed = 0

    def push(self, ids: torch.Tensor, join: bool = False):
        lo, hi = self.sizes[self.rank]
        if tuple(ids.shape) != (hi - lo, self.L):
            raise ValueError(f"local block {tuple(ids.shape)} does not match shard_range {(hi - lo, self.L)}")
        i = self.pushed % self.nbuf
        if self.works[i] is not None:                     # the buffer's previous gather (nbuf batches ago): wait, then drop it
            self.works[i][0].wait()
            self.works[i] = None
        pad = ids
        if ids.shape[0] < self.bmax:
            pad = torch.cat([ids, ids.new_zeros((self.bmax - ids.shape[0], self.L))], 0)
        pad = pad.contiguous()
        buf = self.bufs[i]
        if self.gloo:
            w = dist.all_gather(list(buf.view(self.world, self.bmax, self.L).unbind(0)), pad, group=self.group, async_op=True)
        else:
            w = dist.all_gather_into_tensor(buf, pad, group=self.group, async_op=True)   # rank-major: row i = global clip i
        self.works[i] = (w, pad)
        self.pushed += 1
        if join:
            w.wait()
        return w, buf

    def rows(self, buf: torch.Tensor) -> torch.Tensor:
        """Global clip order [n_clips, L] (drops the padding rows of ragged shards)."""
        if all(hi - lo == self.bmax for lo, hi in self.sizes):
            return buf
        return torch.cat([buf[r * self.bmax: r * self.bmax + (hi - lo)] for r, (lo, hi) in enumerate(self.sizes)], 0)

    def fence(self):
        for k in range(self.nbuf):                        # oldest first
            i = (self.pushed + k) % self.nbuf
            if self.works[i] is not None:
                self.works[i][0].wait()
                self.works[i] = None                      # completed: release the work handle and the padded ids
        dist.barrier(group=self.group)

    def in_flight(self) -> int:
        """Entries the ring still holds (work handle + padded ids): at most nbuf, however many batches were pushed."""
        return sum(e is not None for e in self.works)
