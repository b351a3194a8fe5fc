"""CPU, world_size 2, gloo: the N > 1 path (scatter of uint8 frame batches, per-rank compute, gather of the
float32 flow, max-over-ranks timing).  The per-rank compute is a stand-in (no GPU here); the plumbing under
test is exactly what bench.py / a multi-GPU caller uses with farneback_batch as the compute."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _fake_flow(p, q):
    pf, qf = p.to(torch.float32), q.to(torch.float32)
    return torch.stack([pf - qf, pf + 2 * qf], -1)


def _worker(rank, world, port, n_total, ret):
    for p_ in (ROOT, PKG):
        if p_ not in sys.path:
            sys.path.insert(0, p_)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from nsof import dist as nd
    r, w = nd.init_from_env("gloo")
    assert (r, w) == (rank, world)
    h, wd = 12, 20
    prev = nxt = None
    if rank == 0:
        g = torch.Generator().manual_seed(0)
        prev = torch.randint(0, 256, (n_total, h, wd), dtype=torch.uint8, generator=g)
        nxt = torch.randint(0, 256, (n_total, h, wd), dtype=torch.uint8, generator=g)
    lo, hi = nd.shard_range(n_total)
    seen = {}

    def compute(p, q):
        seen["n"] = p.shape[0]
        return _fake_flow(p, q)

    out = nd.run_sharded(prev, nxt, n_total, (h, wd), "cpu", compute)
    assert seen["n"] == hi - lo
    t = nd.max_over_ranks(1.0 + rank)
    assert t == float(world)
    if rank == 0:
        ok = torch.equal(out, _fake_flow(prev, nxt)) and out.shape == (n_total, h, wd, 2)
        ret.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [7, 2, 1])
def test_scatter_compute_gather_world2(n_total):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29600 + n_total + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=5) is True
