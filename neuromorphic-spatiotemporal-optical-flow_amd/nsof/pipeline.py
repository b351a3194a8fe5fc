"""Events -> temporal-prior surface -> ROI -> flow: the two stages chained (BASELINE.json configs 3 and 5).

The reference ships no code that connects the accumulator's output to the gating input (the
script that builds ``constructed3DMatrix`` is absent), so the glue here is build-defined and kept minimal:
    surface   resistance map R of the accumulator (``resistance_exp(w)``)
    current   I = V_ds / R with V_ds = 1 V (simulation/simulationcode_v4_transistor_uav.m:36)
    blocks    max of I over MEMSIZE x MEMSIZE pixel blocks  -> the coarse "memristor" map
    gating    ``current_to_gray`` -> threshold -> 4-connected components -> ROI  (nsof.gating, reference logic)
    flow      ``calcOpticalFlowFarneback`` on the ROI crop(s) of the frame pair
"""
import numpy as np

from . import gating
from .accumulator import Accumulator, slice_index_array


def surface_to_block_current(resistance, memsize, v_ds=1.0):
    """Block-max of the device current over memsize x memsize pixel blocks (float64, [H//ms][W//ms])."""
    r = np.asarray(resistance, np.float64)
    hb, wb = r.shape[0] // memsize, r.shape[1] // memsize
    cur = v_ds / r[:hb * memsize, :wb * memsize]
    return cur.reshape(hb, memsize, wb, memsize).max(axis=(1, 3))


def events_to_rois(x, y, p, t, sensor_hw, cfg, version=1, polarity="split", slice_us=1000, active_v=-6.0,
                   silent_v=0.0, snapshot_every=33, ctx=None, max_rects=32):
    """Run the accumulator over the stream and return, for every snapshot, the gating map and its ROI rectangles
    (x0, y0, x1, y1) in sensor pixels.  Everything between the event upload and the result stays in HBM: the block maxima of
    the device current of every snapshot (``Accumulator.block_current_dev``), then ONE launch of the gating kernel over
    all snapshots (``gating.roi_from_surface_dev``: gray map, threshold, connected components, rectangles -- one
    wavefront per snapshot); the rectangle table and the tiny gray maps come back in one copy at the end."""
    import torch

    from .context import default_context
    ctx = ctx or default_context()
    H, W = sensor_hw  # noqa: N806
    idx = slice_index_array(t, slice_us)
    rows, cols = H // cfg.MEMSIZE, W // cfg.MEMSIZE
    if rows > 64 or cols > 64:
        # the gating kernel holds a map in one wavefront (64 x 64 cells at most: the reference's maps are 13 x 24 and
        # smaller); larger maps -- e.g. a 1280 x 720 sensor with MEMSIZE 8 -- take the host mirror of the same arithmetic
        out = events_to_rois_host(x, y, p, t, sensor_hw, cfg, version, polarity, slice_us, active_v, silent_v,
                                  snapshot_every, ctx=ctx)
        if cfg.FLAG == 2:   # the union box, as the device kernel returns it
            out = [(g, [(min(r[0] for r in rs), min(r[1] for r in rs), max(r[2] for r in rs), max(r[3] for r in rs))] if rs else [])
                   for g, rs in out]
        return out
    acc = Accumulator(H, W, version, polarity, active_v, silent_v, ctx=ctx)
    try:
        acc.step(x, y, p, t, idx, snap_every=snapshot_every)
        n = acc.snapshot_count()
        if n == 0:
            return []
        cur = torch.empty((n, rows, cols), dtype=torch.float64, device=torch.device("cuda", ctx.device))
        for k in range(n):
            acc.block_current_dev(cfg.MEMSIZE, cur[k], snapshot=k)
        counts, rects, gray = gating.roi_from_surface_dev(cur, n, (rows, cols), (H, W), cfg, max_rects=max_rects, ctx=ctx,
                                                          want_gray=True)
        ctx.synchronize()
        most = int(counts.max().item())
        if most > max_rects:   # a map with more components than the table holds: one more (tiny) launch with room for all
            counts, rects, gray = gating.roi_from_surface_dev(cur, n, (rows, cols), (H, W), cfg, max_rects=most, ctx=ctx,
                                                              want_gray=True)
        lists = gating.rects_to_host(counts, rects, ctx=ctx)
        g = gray.cpu().numpy()
    finally:
        acc.close()
    return [(g[k], lists[k]) for k in range(n)]


def events_to_rois_host(x, y, p, t, sensor_hw, cfg, version=1, polarity="split", slice_us=1000, active_v=-6.0,
                        silent_v=0.0, snapshot_every=33, ctx=None):
    """The same through the host-side mirror of the reference's gating (``gating.connectedComponentsWithStats`` on the
    downloaded block currents): kept as the independent path the device kernel is tested against."""
    H, W = sensor_hw  # noqa: N806
    idx = slice_index_array(t, slice_us)
    acc = Accumulator(H, W, version, polarity, active_v, silent_v, ctx=ctx)
    try:
        acc.step(x, y, p, t, idx, snap_every=snapshot_every)
        blocks = [acc.block_current(cfg.MEMSIZE, snapshot=k) for k in range(acc.snapshot_count())]
    finally:
        acc.close()
    out = []
    for cur in blocks:
        g = gating.current_to_gray(cur)
        tp = np.zeros((H // cfg.MEMSIZE, W // cfg.MEMSIZE))
        tp = gating.update_transition_pic(g, tp, cfg.THRES).astype(np.uint8)
        n, _, stats, _ = gating.connectedComponentsWithStats(tp, cfg.CONNECT)
        rects = [gating._roi(*[int(v) for v in stats[i, :4]], W, H, cfg.MEMSIZE, cfg.MEMSIZE, cfg) for i in range(1, n)]
        out.append((g, rects))
    return out


def events_to_roi_flows(x, y, p, t, sensor_hw, cfg, slice_us=1000, active_v=-6.0, silent_v=0.0, snapshot_every=33,
                        surface_mode="state", ctx=None, max_rects=32, timings=None):
    """BASELINE config 3 as one pipeline on the device: event stream -> leaky-integrate surface (scheme 1) -> every
    ``snapshot_every`` slices an 8-bit surface frame AND the gating map of the same state -> ROI rectangles on the device
    (``gating.roi_from_surface_dev``) -> Farneback flow of every ROI crop between consecutive surface frames, all crops of
    all pairs as ONE work list (``farneback_roi_sequence_dev``), written into frame-sized zero canvases exactly as
    ``opticalFlow3D`` pastes them (optical_flow_seg.py:129-164, 186-204; with FLAG 1 overlapping component boxes are pasted
    in label order, later ones winning).  Which map gates pair (k, k+1) follows ``cfg.bug_compatible`` like the host
    harness (``gating.gating_maps``): True (the default) = the map of frame k, as the shipped scripts do
    (``memimg2 := memimg1``, optical_flow_seg.py:435), False = the map of frame k+1, as ``opticalFlow3D`` is written.
    Events are uploaded once; frames, maps, rectangles and flow stay in HBM -- the only thing that crosses PCIe before the
    result is the rectangle table (16 bytes per ROI, one copy): the work list's shapes are needed on the host.
    Returns ``(frames uint8 [n][H][W], rects [[(x0, y0, x1, y1), ...] per frame], flows float32 [n-1][H][W][2])`` -- torch
    CUDA tensors and the host-side rectangle lists; ``flows[k]`` is zero outside the ROIs of the gating frame."""
    import time

    import torch

    from .context import default_context
    from .farneback import farneback_roi_sequence_dev
    ctx = ctx or default_context()
    H, W = sensor_hw  # noqa: N806
    dev = torch.device("cuda", ctx.device)
    idx = slice_index_array(t, slice_us)
    n_frames = (len(idx) - 1) // snapshot_every
    if n_frames < 2:
        raise ValueError("the stream is shorter than two snapshots")
    rows, cols = H // cfg.MEMSIZE, W // cfg.MEMSIZE
    if rows > 64 or cols > 64:
        raise ValueError(f"gating map {rows}x{cols}: the device gating kernel takes maps up to 64x64 cells (use events_to_rois_host "
                         "+ farneback_pairs for finer grids)")
    frames = torch.empty((n_frames, H, W), dtype=torch.uint8, device=dev)
    cur = torch.empty((n_frames, rows, cols), dtype=torch.float64, device=dev)
    flows = torch.empty((n_frames - 1, H, W, 2), dtype=torch.float32, device=dev)   # zero-filled by the flow call
    torch.cuda.synchronize(dev)
    acc = Accumulator(H, W, 1, "split", active_v, silent_v, ctx=ctx)
    try:
        acc.set_events(x, y, p, t, idx)
        t0 = time.perf_counter()
        for k in range(n_frames):
            acc.run_surface(k * snapshot_every, snapshot_every, frames[k], mode=surface_mode)
            acc.block_current_dev(cfg.MEMSIZE, cur[k])
        counts, rtab = gating.roi_from_surface_dev(cur, n_frames, (rows, cols), (H, W), cfg, max_rects=max_rects, ctx=ctx)
        ctx.synchronize()
        most = int(counts.max().item())
        if most > max_rects:   # more components than the table holds: one more (tiny) gating launch with room for all
            counts, rtab = gating.roi_from_surface_dev(cur, n_frames, (rows, cols), (H, W), cfg, max_rects=most, ctx=ctx)
            ctx.synchronize()
        t1 = time.perf_counter()
        # crop -> flow -> paste of every ROI of every pair: one native call builds the work list from the rectangle table
        n_calls, n_pixels = farneback_roi_sequence_dev(frames, counts, rtab, flows, cfg.farneback_params,
                                                       gate_frame=0 if cfg.bug_compatible else 1, ctx=ctx)
        ctx.synchronize()
        t2 = time.perf_counter()
        rects = gating.rects_to_host(counts, rtab, ctx=ctx)
    finally:
        acc.close()
    if timings is not None:
        timings.update(surface_and_gating_s=t1 - t0, flow_s=t2 - t1, frames=n_frames, roi_calls=int(n_calls), roi_pixels=int(n_pixels))
    return frames, rects, flows


def events_to_flow_sequence(x, y, p, t, sensor_hw, params=None, slice_us=1000, active_v=-6.0, silent_v=0.0,
                            snapshot_every=33, dense=None, ctx=None, timings=None):
    """BASELINE config 5 as one device-resident pipeline: event stream -> dense scheme-1 accumulator update of every
    slice -> every ``snapshot_every`` slices the surface as an 8-bit frame (``Accumulator.surface_u8``, mode "state":
    uint8(255 * w) -- the reference's current -> gray map saturates for the simulator's w >= 0.5 and the reference has
    no surface -> frame step of its own) -> Farneback flow between consecutive surface frames
    (``farneback_sequence``: every frame's pyramid and expansion computed once).  Events are uploaded once; frames and
    flow never leave HBM.  Returns ``(frames uint8 [n][H][W], flows float32 [n-1][H][W][2])`` as torch CUDA tensors.
    ``dense``: None (default) = the accumulator picks (with ``silent_v`` in the dead zone: frames as copy + patch of the
    previous one, ``nsof_accum_run_frames``), True = the every-pixel pass per interval (the roofline run), False = the
    event-pixel update.  Same frames either way.  ``timings`` (a dict) receives the wall time of the two stages."""
    import time

    import torch

    from .context import default_context
    from .farneback import PARAMS_A, farneback_sequence
    ctx = ctx or default_context()
    params = params or PARAMS_A
    H, W = sensor_hw  # noqa: N806
    dev = torch.device("cuda", ctx.device)
    idx = slice_index_array(t, slice_us)
    n_slices = len(idx) - 1
    n_frames = n_slices // snapshot_every
    if n_frames < 2:
        raise ValueError("the stream is shorter than two snapshots")
    frames = torch.empty((n_frames, H, W), dtype=torch.uint8, device=dev)
    flows = torch.empty((n_frames - 1, H, W, 2), dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)
    acc = Accumulator(H, W, 1, "split", active_v, silent_v, ctx=ctx, dense=dense)
    try:
        acc.set_events(x, y, p, t, idx)
        t0 = time.perf_counter()
        acc.run_frames(0, n_frames, snapshot_every, frames)   # the frame of every interval from its last update pass
        ctx.synchronize()
        t1 = time.perf_counter()
        farneback_sequence(frames, flows, n_frames, H, W, params, ctx=ctx)
        ctx.synchronize()
        t2 = time.perf_counter()
    finally:
        acc.close()
    if timings is not None:
        timings.update(accumulator_s=t1 - t0, flow_s=t2 - t1, slices=n_frames * snapshot_every, frames=n_frames)
    return frames, flows


def events_to_flow_sequence_sharded(x, y, p, t, sensor_hw, params=None, slice_us=1000, active_v=-6.0, silent_v=0.0,
                                    snapshot_every=33, dense=None, ctx=None, stats=None):
    """``events_to_flow_sequence`` over the ranks of the current process group (one GPU each): accumulator row bands,
    all-gather of the 8-bit surface frames, contiguous shards of the frame pairs (``nsof.dist.events_to_flow_sharded``
    with the GPU accumulator and ``farneback_sequence`` as the two stages).  Returns ``((lo, hi), frames, flows_local)``
    -- torch CUDA tensors; ``flows_local`` are pairs ``lo .. hi-1`` of the sequence (None for a rank without pairs)."""
    import torch

    from . import dist as nd
    from .context import default_context
    from .farneback import PARAMS_A, farneback_sequence
    ctx = ctx or default_context()
    params = params or PARAMS_A
    H, W = sensor_hw  # noqa: N806
    dev = torch.device("cuda", ctx.device)

    def band_frames(xb, yb, pb, tb, idx, hw, every, n_frames):
        rows, w = hw
        out = torch.empty((n_frames, rows, w), dtype=torch.uint8, device=dev)
        if rows == 0:
            return out
        acc = Accumulator(rows, w, 1, "split", active_v, silent_v, ctx=ctx, dense=dense)
        try:
            acc.set_events(xb, yb, pb, tb, idx)
            acc.run_frames(0, n_frames, every, out)
            ctx.synchronize()
        finally:
            acc.close()
        return out

    def flow_of_frames(fr):
        fr = fr.to(dev).contiguous()
        flows = torch.empty((fr.shape[0] - 1, H, W, 2), dtype=torch.float32, device=dev)
        torch.cuda.synchronize(dev)
        farneback_sequence(fr, flows, fr.shape[0], H, W, params, ctx=ctx)
        ctx.synchronize()
        return flows

    return nd.events_to_flow_sharded(x, y, p, t, (H, W), slice_us, snapshot_every, band_frames, flow_of_frames, device=dev
                                     if torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl"
                                     else None, stats=stats)


def gated_flow(gray_map, prev, nxt, cfg, flow_fn=None):
    """Flow of a frame pair restricted to the ROI(s) the gating map selects (``opticalFlow3D`` of the reference)."""
    kw = {} if flow_fn is None else {"flow_fn": flow_fn}
    return gating.opticalFlow3D(gray_map, gray_map, prev, nxt, cfg.MEMSIZE, cfg.MEMSIZE, cfg, **kw)


# ---- the segmentation experiment of optical_flow_seg.py (__main__, :399-632) as a function --------------------------
SEG_CSV_COLUMNS = ["Frame_Pair", "Original_Flow_Time", "Mem_Flow_Time", "Flow_Time_Improvement",
                   "Flow_Time_Improvement_Percent", "Original_Seg_Time", "Mem_Seg_Time", "Combination_Time",
                   "Original_PA", "Mem_PA", "Region_Percent", "Cal_Times", "Velocity_Times"]   # seg.py:365-379


def calculate_pixel_accuracy(image1, image2):
    """seg.py:383-387: share of identical pixels, in percent."""
    return float(np.sum(image1 == image2)) / image1.size * 100


def run_segmentation(frames_bgr, gt_masks_bgr, mem_state, cfg, names=None, csv_path=None, seg_th=1, merge_flag=False,
                     flow_fn=None, mask_fn=None):
    """The main loop of optical_flow_seg.py for a sequence held in memory: for every pair (i, i+1), i < n-2, the gated
    flow + segmentation ("Mem") and the full-frame flow + segmentation ("Original"), their times, their pixel
    accuracies against the ground-truth mask of frame i+1, and the CSV row the script writes (same 13 columns, same
    formatting).  ``frames_bgr`` / ``gt_masks_bgr``: uint8 [H][W][3] as ``cv2.imread`` returns them; ``mem_state``: the
    ``constructed3DMatrix`` stack.  ``flow_fn`` / ``mask_fn`` default to the GPU path (``calcOpticalFlowFarneback``,
    ``segment.motion_mask``); tests inject the CPU oracle.  Returns ``(rows, mean_mem_accuracy, mean_original_accuracy)``."""
    import csv
    import time

    from . import segment
    from .farneback import calcOpticalFlowFarneback
    flow_fn = flow_fn or calcOpticalFlowFarneback
    mask_fn = mask_fn or (lambda f: segment.motion_mask(f, seg_th))
    names = names or [f"{i + 1}.jpg" for i in range(len(frames_bgr))]
    rows, acc_mem, acc_orig = [], 0.0, 0.0
    if csv_path:
        with open(csv_path, "w", newline="") as fh:
            csv.writer(fh).writerow(SEG_CSV_COLUMNS)
    for i in range(len(frames_bgr) - 2):
        cfg.mem_opticalflow_times.clear(); cfg.mem_cal_times.clear(); cfg.mem_velocity_times.clear()
        memimg1, memimg2 = gating.gating_maps(mem_state, i, cfg)
        prev_gray = gating.frame_to_gray(frames_bgr[i], "RGB2GRAY")
        next_gray = gating.frame_to_gray(frames_bgr[i + 1], "RGB2GRAY")
        gt = np.where(gating.frame_to_gray(gt_masks_bgr[i + 1], "BGR2GRAY") > 127, np.uint8(255), np.uint8(0))
        h, w = next_gray.shape
        out = gating.opticalFlow3D(memimg1, memimg2, prev_gray, next_gray, cfg.MEMSIZE, cfg.MEMSIZE, cfg,
                                   flow_fn=flow_fn)
        flow, region_list = -out[0], out[3]
        if cfg.FLAG == 1:
            num_labels, regions = out[4], out[5]
        else:
            regions = out[4]
            num_labels = 2 if tuple(regions) != (0, 0, 0, 0) else 1
        # Mem segmentation (task_results, seg.py:253-320)
        t0 = time.time()
        motion = np.zeros((h, w), np.uint8)
        t_comb = 0.0
        if num_labels > 1:
            if cfg.FLAG == 1 and merge_flag:
                pad = 20
                boxes = [(max(0, min(r[0] for r in regions) - pad), max(0, min(r[1] for r in regions) - pad),
                          min(w, max(r[2] for r in regions) + pad), min(h, max(r[3] for r in regions) + pad))]
            else:
                boxes = list(regions) if cfg.FLAG == 1 else [tuple(regions)]
            t_comb = time.time() - t0
            for x0, y0, x1, y1 in boxes:
                if x1 > x0 and y1 > y0:
                    motion[y0:y1, x0:x1] = mask_fn(np.ascontiguousarray(flow[y0:y1, x0:x1], np.float32))
        t_mem_seg = time.time() - t0
        # Original: full-frame flow and segmentation (seg.py:493-537)
        t0 = time.time()
        flow1 = -flow_fn(prev_gray, next_gray, None, **cfg.farneback_params.as_kwargs())
        t_orig_flow = time.time() - t0
        t0 = time.time()
        motion1 = mask_fn(np.ascontiguousarray(flow1, np.float32))
        t_orig_seg = time.time() - t0
        a_mem, a_orig = calculate_pixel_accuracy(motion, gt), calculate_pixel_accuracy(motion1, gt)
        acc_mem += a_mem
        acc_orig += a_orig
        t_mem_flow = cfg.mem_opticalflow_times[0]
        imp = t_orig_flow - t_mem_flow
        row = [f"{names[i + 1]}-{names[i]}", f"{t_orig_flow:.4f}", f"{t_mem_flow:.4f}", f"{imp:.4f}",
               f"{imp / t_orig_flow * 100:.2f}", f"{t_orig_seg:.4f}", f"{t_mem_seg:.4f}", f"{t_comb:.4f}",
               f"{a_orig:.4f}", f"{a_mem:.4f}", region_list, ";".join(f"{t:.4f}" for t in cfg.mem_cal_times),
               ";".join(f"{t:.4f}" for t in cfg.mem_velocity_times)]
        rows.append(row)
        if csv_path:
            with open(csv_path, "a", newline="") as fh:
                csv.writer(fh).writerow(row)
    n = max(len(rows), 1)
    return rows, acc_mem / n, acc_orig / n
