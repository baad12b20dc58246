"""CPU, world_size 2 over gloo: the N>1 path of bench.py -- shard bounds and the
single statistics all-reduce -- with the same helpers the GPU run uses."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_items, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from abismal_amd import dist as ad
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = ad.shard_bounds(n_items, rank, world)
    # every rank "maps" its shard: counters are simple functions of the item ids
    ids = torch.arange(lo, hi, dtype=torch.int64)
    stats = torch.stack([torch.tensor(hi - lo), (ids % 3 == 0).sum(), (ids % 7 == 0).sum(), (ids % 11 == 0).sum(),
                         ids.sum(), (ids * 100).sum()]).to(torch.int64)
    elapsed = torch.tensor([0.5 + rank], dtype=torch.float64)
    ad.reduce_stats(stats, elapsed)
    q.put((rank, lo, hi, stats.tolist(), float(elapsed)))
    dist.destroy_process_group()


def test_two_rank_shards_and_stats_reduce():
    world, n_items = 2, 100003
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, lo0, hi0, s0, e0), (r1, lo1, hi1, s1, e1) = out
    assert (lo0, hi1) == (0, n_items) and hi0 == lo1 and abs((hi0 - lo0) - (hi1 - lo1)) <= 1
    ids = torch.arange(n_items)
    want = [n_items, int((ids % 3 == 0).sum()), int((ids % 7 == 0).sum()), int((ids % 11 == 0).sum()),
            int(ids.sum()), int((ids * 100).sum())]
    assert s0 == want and s1 == want
    assert e0 == e1 == 1.5


def test_shard_bounds_cover_everything():
    sys.path.insert(0, ROOT)
    from abismal_amd import dist as ad
    for n in (0, 1, 7, 8, 1000001):
        for w in (1, 2, 3, 8):
            spans = [ad.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
