"""N > 1 path on CPU: two gloo ranks shard a clip batch and gather the caption ids (the only
collective of the path); row i of the result must be global clip i on every rank."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gitcap.dist import gather_captions, shard_range


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
