"""Multi-GPU plumbing: independent frame pairs shard across ranks (one process per GPU).

The path has no exchange step: every pair is independent, so the only
communication is the optional scatter of uint8 frame batches from one rank and the gather of the float32
flow back -- ``torch.distributed`` over RCCL ("nccl" backend on ROCm) on GPUs, gloo on CPU for tests.
"""
import os

import torch
import torch.distributed as dist


def shard_bounds(n_items, world):
    """Contiguous, balanced chunks: [(lo, hi)] * world (the first n_items % world ranks get one extra)."""
    base, extra = divmod(n_items, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def shard_range(n_items, rank=None, world=None):
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    return shard_bounds(n_items, world)[rank]


def init_from_env(backend=None):
    """Reads RANK / WORLD_SIZE / MASTER_* (torch.distributed.run); 127.0.0.1 rendezvous by default."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        dist.init_process_group(backend)
    return dist.get_rank(), dist.get_world_size()


def scatter_pairs(prev_all, next_all, n_total, shape, device, src=0):
    """Rank ``src`` holds uint8 tensors [n_total, H, W]; every rank returns its own shard
    (prev, next) of shape [hi-lo, H, W] on ``device``.  Chunks are padded to equal length for dist.scatter."""
    rank, world = dist.get_rank(), dist.get_world_size()
    bounds = shard_bounds(n_total, world)
    cap = max(hi - lo for lo, hi in bounds)
    h, w = shape
    out = torch.empty((2, cap, h, w), dtype=torch.uint8, device=device)
    chunks = None
    if rank == src:
        chunks = []
        for lo, hi in bounds:
            c = torch.zeros((2, cap, h, w), dtype=torch.uint8, device=device)
            c[0, :hi - lo] = prev_all[lo:hi].to(device)
            c[1, :hi - lo] = next_all[lo:hi].to(device)
            chunks.append(c)
    dist.scatter(out, chunks, src=src)
    lo, hi = bounds[rank]
    return out[0, :hi - lo].contiguous(), out[1, :hi - lo].contiguous()


def gather_flows(flow_local, n_total, dst=0):
    """Inverse of scatter_pairs: rank ``dst`` returns float32 [n_total, H, W, 2]; other ranks return None."""
    rank, world = dist.get_rank(), dist.get_world_size()
    bounds = shard_bounds(n_total, world)
    cap = max(hi - lo for lo, hi in bounds)
    h, w = flow_local.shape[1:3]
    buf = torch.zeros((cap, h, w, 2), dtype=torch.float32, device=flow_local.device)
    buf[:flow_local.shape[0]] = flow_local
    parts = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, parts, dst=dst)
    if rank != dst:
        return None
    return torch.cat([parts[r][:hi - lo] for r, (lo, hi) in enumerate(bounds)], 0)


def max_over_ranks(value, device="cpu"):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def run_sharded(prev_all, next_all, n_total, shape, device, compute, src=0):
    """scatter -> ``compute(prev_shard, next_shard) -> flow_shard`` on every rank -> gather on ``src``."""
    p, q = scatter_pairs(prev_all, next_all, n_total, shape, device, src)
    flow = compute(p, q)
    return gather_flows(flow, n_total, dst=src)


def run_sharded_overlapped(prev_all, next_all, n_total, shape, device, compute, chunk=32, src=0, compute_into=None,
                           src_share=1.0, stats=None, stage_frames=None):
    """scatter -> compute -> gather as a three-stage pipeline over chunks of every rank's shard, point to point
    (``dist.batch_isend_irecv``: ncclSend/ncclRecv pairs under RCCL, one xGMI link per peer, all peers of a round in one
    group) instead of the padded ``dist.scatter`` / ``dist.gather`` through rank ``src``: in round k the frames of chunk k
    travel to the peers while chunk k-1 computes and the flow of chunk k-2 travels back, so the links and the GPUs are
    busy together.  Staging is allocated ONCE (two frame and two flow buffers of one chunk per peer rank; rank ``src``
    receives straight into the result; when its frames are not on ``device`` -- host memory -- rank ``src`` uploads
    every chunk into two preallocated frame buffers per destination) -- nothing is allocated inside the rounds when
    ``compute_into`` is given.  ``stage_frames``: force (True) or forbid (False) that upload staging; None = by device.

    ``prev_all`` / ``next_all``: uint8 [n_total, H, W] on rank ``src`` (ignored elsewhere).
    ``compute(prev, next) -> float32 [n, H, W, 2]`` on ``device``, or ``compute_into(prev, next, out)`` writing a
    preallocated ``out``.  STREAM ORDER: the result must be ordered on torch's CURRENT stream -- RCCL's ``wait()`` orders
    only that stream, so a compute that runs on a private stream (a default ``nsof.Context`` owns one) must either be
    switched to torch's stream (``ctx.set_stream(torch.cuda.current_stream().cuda_stream)``, what bench.py does) or
    synchronise before it returns (``ctx.synchronize()``).
    ``src_share``: rank ``src`` is also the sink of every flow field (7 peers x 16.6 MB per 1080p pair); a share below 1
    gives it proportionally fewer pairs than a peer (0 = it only distributes and collects).
    ``stats`` (dict): receives backend, world size, rounds and bytes moved.
    Returns float32 [n_total, H, W, 2] on rank ``src``, None elsewhere."""
    rank, world = dist.get_rank(), dist.get_world_size()
    h, w = shape
    if world > 1 and src_share != 1.0:
        # weights: peers 1, src src_share
        wsum = (world - 1) + max(0.0, float(src_share))
        cnt = [int(n_total * ((max(0.0, float(src_share)) if r == src else 1.0) / wsum)) for r in range(world)]
        rest = n_total - sum(cnt)
        order = [r for r in range(world) if r != src] + [src]
        for i in range(rest):
            cnt[order[i % world]] += 1
        bounds, lo = [], 0
        for r in range(world):
            bounds.append((lo, lo + cnt[r]))
            lo += cnt[r]
    else:
        bounds = shard_bounds(n_total, world)
    chunks = [[(c, min(c + chunk, hi)) for c in range(lo, hi, chunk)] for lo, hi in bounds]   # per rank: [(a, b)]
    rounds = max([len(c) for c in chunks] + [0])
    out = torch.empty((n_total, h, w, 2), dtype=torch.float32, device=device) if rank == src else None
    my = chunks[rank]
    frames = flows = None
    if rank != src and my:
        cap = max(b - a for a, b in my)
        frames = [torch.empty((2, cap, h, w), dtype=torch.uint8, device=device) for _ in range(2)]
        flows = [torch.empty((cap, h, w, 2), dtype=torch.float32, device=device) for _ in range(2)]
    # rank src: frames that live elsewhere (host memory) go through two preallocated buffers per destination rank
    # (its own chunks included); a send of round k has been waited for before round k + 1 starts, so two slots suffice
    stage = None
    if rank == src and n_total > 0:
        if stage_frames is None:
            stage_frames = prev_all.device != torch.device(device) or next_all.device != torch.device(device)
        if stage_frames:
            stage = {r: [torch.empty((2, max(b - a for a, b in chunks[r]), h, w), dtype=torch.uint8, device=device)
                         for _ in range(2)] for r in range(world) if chunks[r]}

    def staged(r, k):
        """Frames of chunk k of rank r on ``device`` (rank src only): views of the source, or of the staging slot."""
        a, b = chunks[r][k]
        if stage is None:
            return prev_all[a:b], next_all[a:b]
        buf = stage[r][k & 1]
        buf[0, :b - a].copy_(prev_all[a:b], non_blocking=True)
        buf[1, :b - a].copy_(next_all[a:b], non_blocking=True)
        return buf[0, :b - a], buf[1, :b - a]

    def run(pv, nx, dst):
        if compute_into is not None:
            compute_into(pv, nx, dst)
        else:
            dst.copy_(compute(pv, nx))

    for k in range(rounds + 2):
        ops = []
        # stage A: frames of chunk k leave rank src
        if rank == src:
            for r in range(world):
                if r != src and k < len(chunks[r]):
                    pv, nx = staged(r, k)
                    ops.append(dist.P2POp(dist.isend, pv, r))
                    ops.append(dist.P2POp(dist.isend, nx, r))
        elif k < len(my):
            a, b = my[k]
            buf = frames[k & 1]
            ops.append(dist.P2POp(dist.irecv, buf[0, :b - a], src))
            ops.append(dist.P2POp(dist.irecv, buf[1, :b - a], src))
        # stage C: flow of chunk k-2 returns to rank src (straight into the result)
        if rank == src:
            for r in range(world):
                if r != src and 0 <= k - 2 < len(chunks[r]):
                    a, b = chunks[r][k - 2]
                    ops.append(dist.P2POp(dist.irecv, out[a:b], r))
        elif 0 <= k - 2 < len(my):
            a, b = my[k - 2]
            ops.append(dist.P2POp(dist.isend, flows[k & 1][:b - a], src))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        # stage B: chunk k-1 computes while the copies above are in flight
        if 0 <= k - 1 < len(my):
            a, b = my[k - 1]
            if rank == src:
                pv, nx = staged(src, k - 1)
                run(pv, nx, out[a:b])
            else:
                f = frames[(k - 1) & 1]
                run(f[0, :b - a], f[1, :b - a], flows[(k - 1) & 1][:b - a])
        for q in reqs:
            q.wait()
    if stats is not None:
        moved = sum(b - a for r in range(world) if r != src for a, b in chunks[r])
        stats.update(backend=dist.get_backend(), world_size=world, rank=rank, rounds=rounds, chunk_pairs=chunk,
                     pairs_per_rank=[hi - lo for lo, hi in bounds], pairs_moved=moved,
                     bytes_moved=moved * (2 * h * w + 8 * h * w))
    return out


# ---- accumulator: row bands --------------------------------------------------------------
def band_bounds(height, world):
    """Row band [y0, y1) of every rank: pixels are independent, so the H x W state splits by rows with no halo."""
    return shard_bounds(height, world)


def events_in_band(x, y, p, t, y0, y1):
    """The events of rows [y0, y1) with y shifted to the band, in stream order, plus their indices in the stream."""
    import numpy as np
    y = np.asarray(y)
    sel = np.nonzero((y >= y0) & (y < y1))[0]
    return np.asarray(x)[sel], y[sel] - y0, np.asarray(p)[sel], np.asarray(t)[sel], sel


def band_slice_bounds(t_all, t_band, slice_us):
    """Slice boundaries of a band's events on the GLOBAL slice grid (``slice_indices`` uses the first and last
    timestamp of the whole stream, event_mem_sim.py:78-83)."""
    import numpy as np
    t_all = np.asarray(t_all)
    bounds = np.arange(t_all[0], t_all[-1] + slice_us, slice_us)
    return np.searchsorted(np.asarray(t_band), bounds, side="left").astype(np.int64)


def global_slice_times(t_all, slice_us):
    """-> (t_first, t_last) int64 [n_slices]: the timestamp of the first and of the last event of every slice of the
    WHOLE stream (0 for an empty slice).  Scheme 2 tests a pixel's refractory time against the slice's first event and
    re-arms it from the slice's last one (event_mem_sim.py:243-267): properties of the global stream, which a row band
    cannot recover from its own events -- ``simulate_banded`` hands this table to every band."""
    import numpy as np
    t_all = np.asarray(t_all, np.int64)
    bounds = np.arange(t_all[0], t_all[-1] + slice_us, slice_us)
    idx = np.searchsorted(t_all, bounds, side="left").astype(np.int64)
    lo, hi = idx[:-1], idx[1:]
    some = hi > lo
    t_first = np.where(some, t_all[np.minimum(lo, t_all.size - 1)], 0).astype(np.int64)
    t_last = np.where(some, t_all[np.maximum(hi - 1, 0)], 0).astype(np.int64)
    return t_first, t_last


def collective_device(device=None):
    """Device the collectives of the current process group need their tensors on: RCCL ("nccl") only moves GPU
    memory, gloo only host memory.  ``device`` overrides (e.g. ``cuda:LOCAL_RANK`` chosen by the caller)."""
    if device is not None:
        return torch.device(device)
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", int(os.environ.get("LOCAL_RANK", torch.cuda.current_device())))
    return torch.device("cpu")


def simulate_banded(x, y, p, t, sensor_hw, slice_us, simulate_band, dst=0, device=None):
    """The accumulator over row bands, one band per rank (both schemes): every rank filters the (replicated, tiny) event
    stream to its band, runs ``simulate_band(xb, yb, pb, tb, idx_b, (rows, W), (t_first, t_last)) -> w [rows][W] float32``
    -- or a tuple ``(w, w_b)`` for scheme 2 / split -- and the bands are gathered on ``dst`` (the only collective, issued
    on ``device`` -- default: the GPU of this rank under RCCL, host memory under gloo).  On a GPU the callback is
    ``Accumulator(rows, W, version, ...)``: ``set_events`` -> ``set_slice_times(t_first, t_last)`` -> ``run`` -> ``.w()``.
    ``(t_first, t_last)`` = ``global_slice_times``: scheme 2 reads each slice's first / last event time, which belong to
    the whole stream's slice, not to the band's events (scheme 1 ignores them); with them a band's state equals its rows
    of the unsharded run.  Returns float32 [H][W] (or the tuple of two) on ``dst``, None elsewhere."""
    import numpy as np
    rank, world = dist.get_rank(), dist.get_world_size()
    H, W = sensor_hw  # noqa: N806
    y0, y1 = band_bounds(H, world)[rank]
    xb, yb, pb, tb, _ = events_in_band(x, y, p, t, y0, y1)
    idx = band_slice_bounds(t, tb, slice_us)
    dev = collective_device(device)
    res = simulate_band(xb, yb, pb, tb, idx, (y1 - y0, W), global_slice_times(t, slice_us))
    arrays = list(res) if isinstance(res, (tuple, list)) else [res]
    cap = max(hi - lo for lo, hi in band_bounds(H, world))
    outs = []
    for w_band in arrays:
        if not torch.is_tensor(w_band):
            w_band = torch.as_tensor(np.ascontiguousarray(w_band, np.float32))
        buf = torch.zeros((cap, W), dtype=torch.float32, device=dev)
        if y1 > y0:
            buf[:y1 - y0] = w_band.to(dev).reshape(y1 - y0, W)
        parts = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
        dist.gather(buf, parts, dst=dst)
        if rank == dst:
            outs.append(torch.cat([parts[r][:hi - lo] for r, (lo, hi) in enumerate(band_bounds(H, world))], 0).cpu())
    if rank != dst:
        return None
    return tuple(outs) if isinstance(res, (tuple, list)) else outs[0]


# ---- sequence end to end: accumulator bands -> surface frames on every rank -> pairs sharded ----
def events_to_flow_sharded(x, y, p, t, sensor_hw, slice_us, snapshot_every, band_frames, flow_of_frames, device=None,
                           stats=None):
    """BASELINE config 5 over several ranks.  Pixels are independent, so the accumulator state is split into row bands
    (no halo): every rank filters the (replicated, small) event stream to its band and runs
    ``band_frames(xb, yb, pb, tb, idx_b, (rows, W), snapshot_every, n_frames) -> uint8 [n_frames][rows][W]`` -- the band
    of every surface frame, on the global slice grid.  The frames are the one real exchange of the pipeline: an
    all-gather of the bands (``n_frames * H * W`` bytes in total; RCCL under "nccl", host memory under gloo) leaves
    every rank with the full frames.  Pairs of consecutive frames are independent from then on and are sharded in
    contiguous chunks: ``flow_of_frames(frames[lo : hi + 1]) -> float32 [hi - lo][H][W][2]`` (on a GPU:
    ``nsof.farneback_sequence``).  Returns ``((lo, hi), frames, flows_local)``; without an initialised process group
    it is the single-rank pipeline.  ``stats`` (dict) receives the wall time of the three stages on this rank, the bytes
    the all-gather moved and what backend moved them."""
    import time

    import numpy as np
    have = dist.is_available() and dist.is_initialized()
    rank, world = (dist.get_rank(), dist.get_world_size()) if have else (0, 1)
    H, W = sensor_hw  # noqa: N806
    t = np.asarray(t)
    bounds = np.arange(t[0], t[-1] + slice_us, slice_us)
    n_frames = (len(bounds) - 1) // snapshot_every
    if n_frames < 2:
        raise ValueError("the stream is shorter than two snapshots")
    y0, y1 = band_bounds(H, world)[rank]
    xb, yb, pb, tb, _ = events_in_band(x, y, p, t, y0, y1)
    idx = band_slice_bounds(t, tb, slice_us)
    dev = collective_device(device)

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    t0 = time.perf_counter()
    fb = band_frames(xb, yb, pb, tb, idx, (y1 - y0, W), snapshot_every, n_frames)
    if not torch.is_tensor(fb):
        fb = torch.as_tensor(np.ascontiguousarray(fb, np.uint8))
    sync()
    t1 = time.perf_counter()
    bands = band_bounds(H, world)
    if world == 1:
        frames = fb.to(dev).reshape(n_frames, H, W)
    else:
        cap = max(hi - lo for lo, hi in bands)
        buf = torch.zeros((n_frames, cap, W), dtype=torch.uint8, device=dev)
        if y1 > y0:
            buf[:, :y1 - y0] = fb.to(dev).reshape(n_frames, y1 - y0, W)
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
        frames = torch.cat([parts[r][:, :hi - lo] for r, (lo, hi) in enumerate(bands)], 1).contiguous()
    sync()
    t2 = time.perf_counter()
    lo, hi = shard_bounds(n_frames - 1, world)[rank]
    flows = flow_of_frames(frames[lo:hi + 1]) if hi > lo else None
    sync()
    t3 = time.perf_counter()
    if stats is not None:
        cap = max(b - a for a, b in bands)
        stats.update(rank=rank, world_size=world, backend=dist.get_backend() if have else None, band_rows=y1 - y0,
                     band_events=int(len(tb)), frames=n_frames, pairs=(lo, hi), bands_s=t1 - t0, allgather_s=t2 - t1,
                     flow_s=t3 - t2, allgather_bytes_received=(world - 1) * n_frames * cap * W if world > 1 else 0)
    return (lo, hi), frames, flows
