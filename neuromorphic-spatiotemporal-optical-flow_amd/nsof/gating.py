"""ROI gating + flow dispatch: host-side mirror of the reference's ``opticalFlow3D`` family
(/root/reference/optical_flow_seg.py:115-252, 426-435; duplicated verbatim in optical_flow_ob.py,
optical_flow_prediction.py, optical_flow_yolo.py).  SURVEY.md section 8f row 1.

The gating maps are tiny (4x4 .. 24x13): thresholding, 4-connected components and bounding boxes stay on
the host; the flow of every ROI crop runs on the GPU through ``calcOpticalFlowFarneback`` (strided views are
passed straight to the C ABI).  Same names, argument order and return tuples as the reference; the module-level
constants of the reference scripts (MEMSIZE, THRES, EXTEND_*, FLAG) become a ``GatingConfig``.
"""
import time
from dataclasses import dataclass, field

import numpy as np

from .farneback import PARAMS_A, PARAMS_B, PARAMS_C, FarnebackParams, calcOpticalFlowFarneback

# cv2.CC_STAT_* column indices of the stats array
CC_STAT_LEFT, CC_STAT_TOP, CC_STAT_WIDTH, CC_STAT_HEIGHT, CC_STAT_AREA = 0, 1, 2, 3, 4


@dataclass
class GatingConfig:
    """Per-dataset constants (data/*/Parameters.txt; optical_flow_seg.py:36-49,73-81)."""
    MEMSIZE: int = 80
    OFFSET: int = 0
    EXTEND_HEIGHT_UPPER: int = 20
    EXTEND_HEIGHT_LOWER: int = 20
    EXTEND_WIDTH_LEFT: int = 20
    EXTEND_WIDTH_RIGHT: int = 20
    THRES: int = 250
    CONNECT: int = 4
    FLAG: int = 2                     # 1 = one flow call per component, 2 = one call on the union box
    farneback_params: FarnebackParams = PARAMS_A
    bug_compatible: bool = True       # the scripts gate on slice OFFSET+i (memimg2 := memimg1, seg.py:435)
    # dtype of the frame-sized flow canvas: float64 as in the reference (np.zeros((h, w, 2)), seg.py:213); float32
    # holds the same values (the flow IS float32) and halves the host-side zeroing / negation cost per pair
    canvas_dtype: type = np.float64
    # FLAG 1 issues one flow call per component; with a farneback.StreamPool here the calls of a pair overlap on
    # the GPU (same results: the crops are independent).  None = one after the other, like the reference.
    stream_pool: object = None
    # timing lists the reference keeps at module level (optical_flow_seg.py:51-59)
    mem_opticalflow_times: list = field(default_factory=list)
    mem_cal_times: list = field(default_factory=list)
    mem_velocity_times: list = field(default_factory=list)


DATASETS = {
    "grasp": dict(MEMSIZE=80, OFFSET=0, EXTEND_HEIGHT_UPPER=20, EXTEND_HEIGHT_LOWER=20, EXTEND_WIDTH_LEFT=20,
                  EXTEND_WIDTH_RIGHT=20, THRES=250, FLAG=2, farneback_params=PARAMS_A),
    "autodriving": dict(MEMSIZE=200, OFFSET=15, EXTEND_HEIGHT_UPPER=60, EXTEND_HEIGHT_LOWER=60, EXTEND_WIDTH_LEFT=60,
                        EXTEND_WIDTH_RIGHT=60, THRES=114, FLAG=1, farneback_params=PARAMS_B),
    "uav": dict(MEMSIZE=40, OFFSET=15, EXTEND_HEIGHT_UPPER=30, EXTEND_HEIGHT_LOWER=30, EXTEND_WIDTH_LEFT=30,
                EXTEND_WIDTH_RIGHT=30, THRES=114, FLAG=1, farneback_params=PARAMS_B),
    "uavnew2": dict(MEMSIZE=40, OFFSET=0, EXTEND_HEIGHT_UPPER=60, EXTEND_HEIGHT_LOWER=60, EXTEND_WIDTH_LEFT=60,
                    EXTEND_WIDTH_RIGHT=60, THRES=245, FLAG=1, farneback_params=PARAMS_A),
    "tabletennis": dict(MEMSIZE=10, OFFSET=0, EXTEND_HEIGHT_UPPER=20, EXTEND_HEIGHT_LOWER=20, EXTEND_WIDTH_LEFT=20,
                        EXTEND_WIDTH_RIGHT=20, THRES=245, FLAG=2, farneback_params=PARAMS_C),
}


def dataset_config(name, **overrides):
    kw = dict(DATASETS[name])
    kw.update(overrides)
    return GatingConfig(**kw)


def current_to_gray(mem_state):
    """Device current (A) -> 8-bit gating map: ``uint8(clip(-3366/log10(I) - 306, 0, 255))``
    (optical_flow_seg.py:426-431)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        g = -3366 / np.log10(np.asarray(mem_state, np.double)) - 306
    return np.clip(g, 0, 255).astype(np.uint8)


def frame_to_gray(frame, code="RGB2GRAY"):
    """8-bit luma of a 3-channel uint8 frame, as ``cv2.cvtColor(frame, cv2.COLOR_RGB2GRAY)`` computes it
    (optical_flow_seg.py:442-443).  cv2's 8-bit path is fixed point: ``(R*9798 + G*19235 + B*3735 + 2^14) >> 15``.
    The scripts apply RGB2GRAY to ``cv2.imread`` output, which is B,G,R ordered, so channel 0 (blue) gets the red
    weight; ``code="BGR2GRAY"`` gives the conventional weighting (seg.py:447)."""
    f = np.asarray(frame)
    if f.ndim != 3 or f.shape[2] != 3 or f.dtype != np.uint8:
        raise ValueError("frame must be uint8 [H,W,3]")
    wts = {"RGB2GRAY": (9798, 19235, 3735), "BGR2GRAY": (3735, 19235, 9798)}[code]
    acc = np.full(f.shape[:2], 1 << 14, np.int32)
    for c in range(3):
        acc += f[..., c].astype(np.int32) * wts[c]
    return (acc >> 15).astype(np.uint8)


def load_gating_stack(mat_path):
    """The ``constructed3DMatrix`` stack (rows x cols x slices, device currents in ampere) of a dataset's
    ``constructed_3D_matrix.mat`` (optical_flow_seg.py:412-413).  Needs scipy; the file is parsed, nothing in it runs."""
    import scipy.io
    return scipy.io.loadmat(mat_path)["constructed3DMatrix"]


def gating_maps(mem_state, i, cfg):
    """(memimg1, memimg2) for frame pair i from the ``constructed3DMatrix`` stack (seg.py:416-437).
    With ``cfg.bug_compatible`` memimg2 is a copy of memimg1, as in the shipped scripts."""
    m1 = current_to_gray(mem_state[:, :, cfg.OFFSET + i])
    m2 = m1.astype(np.uint8) if cfg.bug_compatible else current_to_gray(mem_state[:, :, cfg.OFFSET + i + 1])
    return m1, m2


def update_transition_pic(prev_memristor, transition_pic, thres):
    """optical_flow_seg.py:115-121 (the reference jit-compiles this double loop with numba)."""
    h = min(prev_memristor.shape[0], transition_pic.shape[0])
    w = min(prev_memristor.shape[1], transition_pic.shape[1])
    if prev_memristor.shape[0] > transition_pic.shape[0] or prev_memristor.shape[1] > transition_pic.shape[1]:
        raise IndexError("gating map larger than the transition picture (the reference would write out of bounds)")
    transition_pic[:h, :w][prev_memristor[:h, :w] >= thres] = 255
    return transition_pic


def connectedComponentsWithStats(image, connectivity=4):  # noqa: N802
    """Subset of ``cv2.connectedComponentsWithStats`` used by the reference (seg.py:223): labels in raster order of
    each component's first pixel, label 0 = background, stats columns LEFT, TOP, WIDTH, HEIGHT, AREA."""
    img = np.asarray(image) != 0
    h, w = img.shape
    labels = np.zeros((h, w), np.int32)
    nbrs = [(-1, 0), (1, 0), (0, -1), (0, 1)]
    if connectivity == 8:
        nbrs += [(-1, -1), (-1, 1), (1, -1), (1, 1)]
    elif connectivity != 4:
        raise ValueError("connectivity must be 4 or 8")
    n = 0
    for y in range(h):
        for x in range(w):
            if img[y, x] and labels[y, x] == 0:
                n += 1
                labels[y, x] = n
                stack = [(y, x)]
                while stack:
                    cy, cx = stack.pop()
                    for dy, dx in nbrs:
                        ny, nx = cy + dy, cx + dx
                        if 0 <= ny < h and 0 <= nx < w and img[ny, nx] and labels[ny, nx] == 0:
                            labels[ny, nx] = n
                            stack.append((ny, nx))
    stats = np.zeros((n + 1, 5), np.int32)
    cents = np.zeros((n + 1, 2), np.float64)
    for lab in range(n + 1):
        ys, xs = np.nonzero(labels == lab)
        if ys.size == 0:
            continue
        stats[lab] = (xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1, ys.size)
        cents[lab] = (xs.mean(), ys.mean())
    return n + 1, labels, stats, cents


def _roi(x, y, a, b, w, h, pixel_width, pixel_height, cfg):
    x_start = max(x * pixel_width - cfg.EXTEND_WIDTH_LEFT, 0)
    y_start = max(y * pixel_height - cfg.EXTEND_HEIGHT_UPPER, 0)
    x_end = min((x + a) * pixel_width + cfg.EXTEND_WIDTH_RIGHT, w)
    y_end = min((y + b) * pixel_height + cfg.EXTEND_HEIGHT_LOWER, h)
    return int(x_start), int(y_start), int(x_end), int(y_end)


def process_separate_regions(stats, rgbimg1, rgbimg2, flow, pixel_width, pixel_height, cfg, flow_fn):
    """FLAG == 1 (optical_flow_seg.py:123-166): one flow call per connected component."""
    region_list, regions_info, jobs = [], [], []
    h, w = rgbimg1.shape[:2]
    for i in range(1, len(stats)):
        t0 = time.time()
        x, y, a, b = (int(stats[i, k]) for k in (CC_STAT_LEFT, CC_STAT_TOP, CC_STAT_WIDTH, CC_STAT_HEIGHT))
        x_start, y_start, x_end, y_end = _roi(x, y, a, b, w, h, pixel_width, pixel_height, cfg)
        regions_info.append((x_start, y_start, x_end, y_end))
        prev_region = rgbimg1[y_start:y_end, x_start:x_end]
        next_region = rgbimg2[y_start:y_end, x_start:x_end]
        cfg.mem_cal_times.append(time.time() - t0)
        region_list.append(prev_region.shape[0] * prev_region.shape[1] / (h * w) * 100)
        if prev_region.size > 0 and next_region.size > 0:
            if cfg.stream_pool is not None:
                jobs.append((len(cfg.mem_velocity_times), (y_start, y_end, x_start, x_end), prev_region, next_region))
                cfg.mem_velocity_times.append(0)
                continue
            t0 = time.time()
            flow[y_start:y_end, x_start:x_end] = flow_fn(prev_region, next_region, None,
                                                         **cfg.farneback_params.as_kwargs())
            cfg.mem_velocity_times.append(time.time() - t0)
        else:
            cfg.mem_velocity_times.append(0)
    if jobs:   # all components of the pair at once, one stream each
        t0 = time.time()
        flows = cfg.stream_pool.map([(j[2], j[3]) for j in jobs], cfg.farneback_params)
        dt = (time.time() - t0) / len(jobs)
        for (slot, (ys, ye, xs, xe), _, _), f in zip(jobs, flows):
            flow[ys:ye, xs:xe] = f        # overlapping boxes: later components win, as in the sequential loop
            cfg.mem_velocity_times[slot] = dt
    return flow, cfg.mem_cal_times, cfg.mem_velocity_times, region_list, len(stats), regions_info


def process_merged_region(stats, rgbimg1, rgbimg2, flow, pixel_width, pixel_height, cfg, flow_fn):
    """FLAG == 2 (optical_flow_seg.py:168-209): one flow call on the union bounding box."""
    region_list = []
    h, w = rgbimg1.shape[:2]
    t0 = time.time()
    idx = range(1, len(stats))
    x_min = min(int(stats[i, CC_STAT_LEFT]) for i in idx)
    y_min = min(int(stats[i, CC_STAT_TOP]) for i in idx)
    x_max = max(int(stats[i, CC_STAT_LEFT] + stats[i, CC_STAT_WIDTH]) for i in idx)
    y_max = max(int(stats[i, CC_STAT_TOP] + stats[i, CC_STAT_HEIGHT]) for i in idx)
    x_start, y_start, x_end, y_end = _roi(x_min, y_min, x_max - x_min, y_max - y_min, w, h, pixel_width,
                                          pixel_height, cfg)
    prev_region = rgbimg1[y_start:y_end, x_start:x_end]
    next_region = rgbimg2[y_start:y_end, x_start:x_end]
    cfg.mem_cal_times.append(time.time() - t0)
    region_list.append(prev_region.shape[0] * prev_region.shape[1] / (h * w) * 100)
    if prev_region.size > 0 and next_region.size > 0:
        t0 = time.time()
        flow[y_start:y_end, x_start:x_end] = flow_fn(prev_region, next_region, None,
                                                     **cfg.farneback_params.as_kwargs())
        cfg.mem_velocity_times.append(time.time() - t0)
    return flow, cfg.mem_cal_times, cfg.mem_velocity_times, region_list, (x_start, y_start, x_end, y_end)


def opticalFlow3D(memimg1, memimg2, rgbimg1, rgbimg2, pixel_width, pixel_height, cfg=None,  # noqa: N802
                  flow_fn=calcOpticalFlowFarneback):
    """optical_flow_seg.py:211-252.  Returns ``(flow float64 HxWx2, cal_times, vel_times, region_list,
    num_labels, regions_info)`` for FLAG 1 and ``(flow, cal_times, vel_times, region_list, (x0, y0, x1, y1))`` for
    FLAG 2 -- the caller negates the flow (seg.py:461)."""
    cfg = cfg or GatingConfig()
    t0 = time.time()
    h, w = rgbimg1.shape[:2]
    flow = np.zeros((h, w, 2), cfg.canvas_dtype)
    transition_pic = np.zeros((int(h / pixel_height), int(w / pixel_width)))
    transition_pic = update_transition_pic(memimg2, transition_pic, cfg.THRES).astype(np.uint8)
    num_labels, _, stats, _ = connectedComponentsWithStats(transition_pic, connectivity=cfg.CONNECT)
    if num_labels == 1:
        dt = time.time() - t0
        cfg.mem_cal_times.append(dt)
        cfg.mem_opticalflow_times.append(dt)
        if cfg.FLAG == 1:
            return flow, cfg.mem_cal_times, cfg.mem_opticalflow_times, [], num_labels, []
        return flow, cfg.mem_cal_times, cfg.mem_opticalflow_times, [], (0, 0, 0, 0)
    if cfg.FLAG == 1:
        out = process_separate_regions(stats, rgbimg1, rgbimg2, flow, pixel_width, pixel_height, cfg, flow_fn)
    else:
        out = process_merged_region(stats, rgbimg1, rgbimg2, flow, pixel_width, pixel_height, cfg, flow_fn)
    cfg.mem_opticalflow_times.append(time.time() - t0)
    return out


def roi_from_surface(current, frame_hw, cfg):
    """ROI rectangles ``[(x0, y0, x1, y1), ...]`` of one gating slice through the C ABI (``nsof_roi_from_surface``):
    ``current`` = a slice of the ``constructed3DMatrix`` stack (ampere).  Same arithmetic as ``opticalFlow3D`` up to
    the crop (optical_flow_seg.py:211-252): current -> gray, threshold, connected components, per-component boxes
    (FLAG 1) or their union (FLAG 2), scaled by MEMSIZE and extended.  Needs no GPU."""
    import ctypes as C

    from . import _lib
    cur = np.ascontiguousarray(current, np.float64)
    h, w = frame_hw
    lib = _lib.load()
    cap = 64
    while True:
        rects = (C.c_int * (4 * cap))()
        n = lib.nsof_roi_from_surface(cur.ctypes.data, cur.shape[0], cur.shape[1], int(w), int(h), cfg.MEMSIZE, cfg.THRES,
                                      cfg.EXTEND_WIDTH_LEFT, cfg.EXTEND_WIDTH_RIGHT, cfg.EXTEND_HEIGHT_UPPER,
                                      cfg.EXTEND_HEIGHT_LOWER, cfg.CONNECT, cfg.FLAG, rects, cap)
        if n < 0:
            raise ValueError(f"nsof_roi_from_surface failed ({n})")
        if n <= cap:
            return [tuple(rects[4 * i:4 * i + 4]) for i in range(n)]
        cap = n


def roi_from_surface_dev(d_current, n_maps, map_hw, frame_hw, cfg, max_rects=32, ctx=None, want_gray=False, map_stride=None):
    """Device twin of ``roi_from_surface`` for ``n_maps`` gating maps already in HBM (``d_current``: torch float64 tensor
    / device address, map k at ``+ k * map_stride`` doubles, default rows * cols): one wavefront per map forms the gray
    map, thresholds it, labels the connected components in raster order and writes the crop rectangles
    (``nsof_roi_from_surface_dev``; optical_flow_seg.py:115-121, 211-252).  Returns device tensors
    ``(counts int32 [n_maps], rects int32 [n_maps][max_rects][4][, gray uint8 [n_maps][rows][cols]])`` -- nothing is
    synchronised or copied to the host; ``rects_to_host`` fetches them in one small copy."""
    import torch

    from .context import default_context, dev_ptr
    ctx = ctx or default_context()
    rows, cols = map_hw
    h, w = frame_hw
    dev = torch.device("cuda", ctx.device)
    counts = torch.empty((n_maps,), dtype=torch.int32, device=dev)
    rects = torch.empty((n_maps, max_rects, 4), dtype=torch.int32, device=dev)
    gray = torch.empty((n_maps, rows, cols), dtype=torch.uint8, device=dev) if want_gray else None
    ctx.check(ctx._lib.nsof_roi_from_surface_dev(
        ctx.ptr, dev_ptr(d_current), int(n_maps), int(rows * cols if map_stride is None else map_stride), int(rows), int(cols),
        int(w), int(h), cfg.MEMSIZE, cfg.THRES, cfg.EXTEND_WIDTH_LEFT, cfg.EXTEND_WIDTH_RIGHT, cfg.EXTEND_HEIGHT_UPPER,
        cfg.EXTEND_HEIGHT_LOWER, cfg.CONNECT, cfg.FLAG, int(max_rects), dev_ptr(counts), dev_ptr(rects),
        dev_ptr(gray) if want_gray else None), "roi_from_surface_dev")
    return (counts, rects, gray) if want_gray else (counts, rects)


def rects_to_host(counts, rects, ctx=None):
    """``[[(x0, y0, x1, y1), ...] per map]`` from the device tables of ``roi_from_surface_dev`` (one small D2H copy)."""
    from .context import default_context
    (ctx or default_context()).synchronize()
    c, r = counts.cpu().numpy(), rects.cpu().numpy()
    if (c > r.shape[1]).any():
        raise ValueError(f"a gating map has {int(c.max())} components; max_rects was {r.shape[1]}")
    return [[tuple(int(v) for v in r[k, i]) for i in range(int(c[k]))] for k in range(len(c))]
