"""N > 1 path on CPU: two gloo ranks shard a clip batch and gather the caption ids (the only
collective of the path); row i of the result must be global clip i on every rank."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gitcap.dist import CaptionGatherRing, gather_captions, shard_range


def _worker(rank, world, port, n_clips, L, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(n_clips, rank, world)
        # stand-in for this rank's greedy_decode output: the row of global clip c is [101, c+1, c+2, ...]
        ids = torch.zeros((hi - lo, L), dtype=torch.long)
        for i, c in enumerate(range(lo, hi)):
            ids[i] = torch.arange(c, c + L)
            ids[i, 0] = 101
        out = gather_captions(ids, n_clips)
        q.put((rank, out.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [32, 5])
def test_two_rank_gather_is_rank_major(n_clips):
    world, L = 2, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + n_clips
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_clips, L, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = [[101] + list(range(c + 1, c + L)) for c in range(n_clips)]
    for r in range(world):
        assert got[r] == expect


def _ring_worker(rank, world, port, n_clips, L, nbatch, nbuf, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(n_clips, rank, world)
        ring = CaptionGatherRing(n_clips, L, "cpu", nbuf=nbuf)
        outs = []
        for b in range(nbatch):                           # batch b: the row of global clip c is [101, 1000*b + c + 1, ...]
            ids = torch.zeros((hi - lo, L), dtype=torch.long)
            for i, c in enumerate(range(lo, hi)):
                ids[i] = torch.arange(1000 * b + c, 1000 * b + c + L)
                ids[i, 0] = 101
            w, buf = ring.push(ids, join=(b % 3 == 0))    # some joined at once, most a few batches later
            outs.append((w, buf))
            # the ring is bounded: a long-running captioner must not accumulate work handles / padded ids
            assert ring.in_flight() <= nbuf and len(ring.works) == nbuf, (ring.in_flight(), nbuf)
            if len(outs) > 2:                             # consume two batches behind, like bench.py's pipeline
                w0, buf0 = outs[-3]
                w0.wait()
                q.put((rank, b - 2, ring.rows(buf0).tolist()))
                outs[-3] = None                           # (the test itself keeps nothing alive either)
        ring.fence()
        assert ring.in_flight() == 0
        for b in (nbatch - 2, nbatch - 1):
            q.put((rank, b, ring.rows(outs[b][1]).tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [32, 5])
def test_two_rank_gather_ring(n_clips):
    """The asynchronous gather ring bench.py uses for N > 1 (>= 2 x nbuf batches, ragged last shard when
    n_clips = 5): every batch arrives complete and in global clip order on every rank."""
    world, L, nbuf = 2, 6, 3
    nbatch = 12 * nbuf + 2                                # many times the ring size: the ring must stay at nbuf entries
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30100 + (os.getpid() % 500) + n_clips
    procs = [ctx.Process(target=_ring_worker, args=(r, world, port, n_clips, L, nbatch, nbuf, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world * nbatch):
        rank, b, rows = q.get(timeout=120)
        got[(rank, b)] = rows
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for b in range(nbatch):
        expect = [[101] + list(range(1000 * b + c + 1, 1000 * b + c + L)) for c in range(n_clips)]
        for r in range(world):
            assert got[(r, b)] == expect, (r, b)


@pytest.mark.gpu
def test_gather_ring_on_rccl_single_rank():
    """The same ring over RCCL ("nccl" backend) with device tensors: one rank on the GPU box (the multi-rank runs are
    the driver's); every batch must come back intact after more pushes than ring buffers."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    port = 31200 + (os.getpid() % 500)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        ring = CaptionGatherRing(16, 21, torch.device("cuda", 0), nbuf=3)
        outs = []
        for b in range(8):
            ids = (torch.arange(16 * 21, device="cuda").view(16, 21) + 1000 * b).long()
            w, buf = ring.push(ids, join=(b % 4 == 0))
            outs.append((w, buf, ids))
            if len(outs) > 2:
                w0, buf0, ids0 = outs[-3]
                w0.wait()
                torch.cuda.synchronize()
                assert torch.equal(ring.rows(buf0), ids0)
        ring.fence()
        torch.cuda.synchronize()
        assert torch.equal(ring.rows(outs[-1][1]), outs[-1][2])
        assert torch.equal(gather_captions(outs[-1][2], 16), outs[-1][2])
    finally:
        dist.destroy_process_group()
