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
    # the chunked, point-to-point, overlapped variant gives the same field (chunks of 2 pairs: several rounds)
    out2 = nd.run_sharded_overlapped(prev, nxt, n_total, (h, wd), "cpu", _fake_flow, chunk=2)
    if rank == 0:
        assert torch.equal(out2, out)
    else:
        assert out2 is None
    # preallocated staging + compute_into; the sink rank with a smaller share / no share of the pairs; the stats record
    for share in (1.0, 0.5, 0.0):
        st = {}
        # stage_frames: rank 0 uploads every chunk through its two preallocated buffers per destination (what it does
        # when the frames are host memory and the ranks compute on GPUs)
        out3 = nd.run_sharded_overlapped(prev, nxt, n_total, (h, wd), "cpu", None, chunk=3, src_share=share, stats=st,
                                         compute_into=lambda p_, q_, o_: o_.copy_(_fake_flow(p_, q_)),
                                         stage_frames=share != 1.0)
        assert st["backend"] == "gloo" and st["world_size"] == world and sum(st["pairs_per_rank"]) == n_total
        if share == 0.0 and world > 1:
            assert st["pairs_per_rank"][0] == 0 and st["pairs_moved"] == n_total
        if rank == 0:
            assert torch.equal(out3, out), share
        else:
            assert out3 is None
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


def _band_worker(rank, world, port, ret):
    for p_ in (ROOT, PKG):
        if p_ not in sys.path:
            sys.path.insert(0, p_)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from nsof import dist as nd
    from nsof import synth
    from oracle import oracle
    nd.init_from_env("gloo")
    H, W = 47, 64                                       # odd height: bands of 24 and 23 rows
    x, y, p, t = synth.make_events(11, W, H, 3000, 40_000, box=(12, 9))

    def band(xb, yb, pb, tb, idx, hw, slice_times):     # stand-in for the per-rank GPU accumulator: the CPU oracle
        w = np.full(hw, 0.5, np.float32)
        for s in range(len(idx) - 1):
            V = np.zeros(hw, np.float32)
            V[yb[idx[s]:idx[s + 1]], xb[idx[s]:idx[s + 1]]] = -6.0
            w = oracle.accum_update_state(w, V)
        return w

    out = nd.simulate_banded(x, y, p, t, (H, W), 1000, band)
    if rank == 0:
        full = oracle.accum_simulate(x, y, p, t, H, W, 1, "split", 1000, -6.0, 0.0)["w_final"]
        ret.put(bool(np.array_equal(out.numpy(), full)))
    dist.barrier()
    dist.destroy_process_group()


def test_accumulator_row_bands_world2():
    """Row-band sharding of the accumulator (scheme 1): two ranks, each its band on the global slice grid, gathered
    state == the unsharded run."""
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29850 + (os.getpid() % 100)
    procs = [ctx.Process(target=_band_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert ret.get(timeout=5) is True


def _scheme2_band(xb, yb, pb, tb, idx, hw, slice_times, split, active_v, silent_v, update_state):
    """Scheme 2 (event_mem_sim.py:231-270) of one row band in numpy, the refractory timestamps of every slice taken from
    ``slice_times`` (the whole stream's): the stand-in for ``Accumulator(..., 2, ...)`` + ``set_slice_times`` in the gloo test."""
    t_first, t_last = slice_times
    wa = np.full(hw, 0.5, np.float32)
    wb = np.full(hw, 0.5, np.float32)
    ok = [np.zeros(hw, np.int64), np.zeros(hw, np.int64)]
    for s in range(len(idx) - 1):
        sl = slice(idx[s], idx[s + 1])
        V = [np.full(hw, silent_v, np.float32), np.full(hw, silent_v, np.float32)]   # noqa: N806
        if sl.stop > sl.start:
            groups = ((0, pb[sl] == 1), (1, pb[sl] == 0)) if split else ((0, np.ones(sl.stop - sl.start, bool)),)
            for arr, m in groups:
                ys, xs = yb[sl][m], xb[sl][m]
                good = ok[arr][ys, xs] <= t_first[s]
                V[arr][ys[good], xs[good]] = np.float32(silent_v) + np.float32(active_v)
                ok[arr][ys[good], xs[good]] = t_last[s] + 800
        wa = update_state(wa, V[0])
        if split:
            wb = update_state(wb, V[1])
    return (wa, wb) if split else wa


def _refractory_stream(H, W, seed=5, duration_us=50_000):   # noqa: N803
    """A stream on which scheme 2's refractory rule depends on WHOSE first / last event time a slice is given: a lone
    pixel in the top rows fires every ~950 us -- further apart than the 800 us refractory time, so judged by their own
    band's (sparse) events they would be driven in every slice -- while dense noise in the bottom rows puts the whole
    stream's last event near the end of every slice, which blocks them every other slice."""
    rng = np.random.default_rng(seed)
    xs, ys, ts = [], [], []
    for k in range(1):   # ONE lone pixel: a second one in the same band would already put that band's last event late
        px, py = int(rng.integers(0, W)), int(rng.integers(0, max(H // 4, 1)))
        tt = np.arange(int(rng.integers(0, 400)), duration_us, 930 + 15 * k)
        xs.append(np.full(tt.size, px)); ys.append(np.full(tt.size, py)); ts.append(tt)
    n = 40 * (duration_us // 1000)
    xs.append(rng.integers(0, W, n)); ys.append(rng.integers(H - max(H // 4, 1), H, n)); ts.append(rng.integers(0, duration_us, n))
    x, y, t = np.concatenate(xs), np.concatenate(ys), np.concatenate(ts)
    o = np.argsort(t, kind="stable")
    p = rng.integers(0, 2, t.size)
    return x[o].astype(np.int64), y[o].astype(np.int64), p[o].astype(np.int64), t[o].astype(np.int64)


def _band2_worker(rank, world, port, ret):
    for p_ in (ROOT, PKG):
        if p_ not in sys.path:
            sys.path.insert(0, p_)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from nsof import dist as nd
    from nsof import synth
    from oracle import oracle
    nd.init_from_env("gloo")
    H, W = 41, 56                                       # noqa: N806  (odd height: bands of 21 and 20 rows)
    x, y, p, t = _refractory_stream(H, W)
    oks = []
    for polarity in ("split", "magnitude"):
        split = polarity == "split"
        out = nd.simulate_banded(x, y, p, t, (H, W), 1000,
                                 lambda *a: _scheme2_band(*a, split, -0.3, 0.0, oracle.accum_update_state))   # noqa: B023
        if rank == 0:
            full = oracle.accum_simulate(x, y, p, t, H, W, 2, polarity, 1000, -0.3, 0.0)
            if split:
                oks.append(bool(np.array_equal(out[0].numpy(), full["w_final"]) and np.array_equal(out[1].numpy(), full["w_final_b"])))
            else:
                oks.append(bool(np.array_equal(out.numpy(), full["w_final"])))
            # a band run on its OWN first / last event times must differ somewhere, or the stream proves nothing
            y0, y1 = nd.band_bounds(H, world)[0]
            xb, yb, pb, tb, _ = nd.events_in_band(x, y, p, t, y0, y1)
            idx = nd.band_slice_bounds(t, tb, 1000)
            lo, hi = idx[:-1], idx[1:]
            own = (np.where(hi > lo, tb[np.minimum(lo, tb.size - 1)], 0), np.where(hi > lo, tb[np.maximum(hi - 1, 0)], 0))
            w_own = _scheme2_band(xb, yb, pb, tb, idx, (y1 - y0, W), own, split, -0.3, 0.0, oracle.accum_update_state)
            w_own = w_own[0] if split else w_own
            oks.append(bool(not np.array_equal(w_own, full["w_final"][y0:y1])))
        else:
            assert out is None
    if rank == 0:
        ret.put(oks)
    dist.barrier()
    dist.destroy_process_group()


def test_accumulator_scheme2_row_bands_world2():
    """Row bands for scheme 2 (VERDICT r3 Missing 2): the per-slice first / last event times come from the unfiltered
    stream (nsof.dist.global_slice_times), so the gathered state == the unsharded oracle run for split and magnitude."""
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29900 + (os.getpid() % 50)
    procs = [ctx.Process(target=_band2_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert ret.get(timeout=5) == [True, True, True, True]


def _seq_worker(rank, world, port, ret):
    for p_ in (ROOT, PKG):
        if p_ not in sys.path:
            sys.path.insert(0, p_)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from nsof import dist as nd
    from nsof import synth
    nd.init_from_env("gloo")
    H, W, every = 37, 52, 5   # noqa: N806  (odd height: bands of 19 and 18 rows)
    x, y, p, t = synth.make_events(11, W, H, 6000, 48_000, box=(12, 9))

    def band_frames(xb, yb, pb, tb, idx, hw, ev, n_frames):
        # stand-in for the GPU accumulator: running event count per pixel, one frame every `ev` slices
        cnt = np.zeros(hw, np.int64)
        out = np.zeros((n_frames,) + tuple(hw), np.uint8)
        for s in range(n_frames * ev):
            np.add.at(cnt, (yb[idx[s]:idx[s + 1]], xb[idx[s]:idx[s + 1]]), 1)
            if (s + 1) % ev == 0:
                out[s // ev] = (cnt * 37 % 256).astype(np.uint8)
        return out

    def flow_of_frames(fr):
        return _fake_flow(fr[:-1], fr[1:])

    (lo, hi), frames, flows = nd.events_to_flow_sharded(x, y, p, t, (H, W), 1000, every, band_frames, flow_of_frames)
    from nsof.accumulator import slice_index_array
    idx = slice_index_array(t, 1000)
    n_frames = (len(idx) - 1) // every
    full = torch.as_tensor(band_frames(x, y, p, t, idx, (H, W), every, n_frames))
    want = _fake_flow(full[:-1], full[1:])
    ok = torch.equal(frames, full) and (lo, hi) == nd.shard_bounds(n_frames - 1, world)[rank] and hi > lo \
        and torch.equal(flows, want[lo:hi])
    ret.put((rank, bool(ok), n_frames))
    dist.barrier()
    dist.destroy_process_group()


def test_events_to_flow_sharded_world2():
    """Config 5 over two ranks (third row): accumulator row bands -> all-gather of the surface
    frames -> contiguous shards of the frame pairs.  Frames on every rank == the unsharded frames, each rank's flows
    == its slice of the unsharded sequence."""
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = 29750 + (os.getpid() % 100)
    procs = [ctx.Process(target=_seq_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    got = sorted(ret.get(timeout=5) for _ in range(2))
    assert [g[:2] for g in got] == [(0, True), (1, True)] and got[0][2] >= 4
