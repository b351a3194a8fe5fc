"""The reference's evaluation loop as ONE work list: every flow call of every frame pair of several datasets.

For each frame pair (i, i+1) the reference's scripts make the gated call(s) -- ``opticalFlow3D``: gating map ->
threshold -> 4-connected components -> one ``cv2.calcOpticalFlowFarneback`` per component (FLAG 1,
/root/reference/optical_flow_seg.py:129-164) or one on the union box (FLAG 2, :186-203), pasted into a zero
canvas -- and then the full-frame "Original" call (:492-496), with the dataset's own constants
(``data/*/Parameters.txt``).  BASELINE.json config 4 runs that over grasp + autodriving + uav + uavnew2 +
tabletennis at once.  The gating arithmetic is host work on maps of at most 24x13 cells and does not depend on
any flow, so here all rectangles are derived first and all flow calls of all pairs are issued together:
one ``farneback_pairs`` work list per Farneback parameter set (calls of different shapes share every kernel
launch), ROI results written straight into their canvases.  Calls shard round-robin over ranks.
"""
from dataclasses import dataclass, field

import numpy as np

from . import gating, synth
from .farneback import calcOpticalFlowFarneback, farneback_pairs

# frame (height, width) and number of RGB frames of the reference's data/<name>/RGB
DATASET_FRAMES = {"grasp": (1920, 1080, 101), "autodriving": (801, 801, 100), "uav": (161, 161, 100),
                  "uavnew2": (600, 600, 48), "tabletennis": (160, 160, 21)}


@dataclass
class FlowCall:
    """One ``calcOpticalFlowFarneback(prev_region, next_region, None, **params)`` of the evaluation loop."""
    dataset: str
    pair: int
    kind: str                 # "roi" (gated call) or "full" (the script's full-frame baseline)
    rect: tuple               # (x0, y0, x1, y1) in frame pixels
    params: object            # FarnebackParams
    prev: np.ndarray          # uint8 view of the frame (strided for ROI calls)
    next: np.ndarray
    flow: np.ndarray = field(repr=False, default=None)   # float32 (h, w, 2) target: a view of the pair's canvas ...
    paste_to: np.ndarray = field(repr=False, default=None)   # ... or a private field pasted here afterwards (below)

    @property
    def shape(self):
        return self.prev.shape


def roi_rects(gate_map, frame_hw, cfg):
    """Rectangles the gated path crops for one gating map: per component (FLAG 1) or the union box (FLAG 2);
    ``[]`` when nothing crosses the threshold.  Same arithmetic as ``opticalFlow3D`` (optical_flow_seg.py:211-252)."""
    h, w = frame_hw
    tp = np.zeros((int(h / cfg.MEMSIZE), int(w / cfg.MEMSIZE)))
    tp = gating.update_transition_pic(gate_map, tp, cfg.THRES).astype(np.uint8)
    n, _, stats, _ = gating.connectedComponentsWithStats(tp, connectivity=cfg.CONNECT)
    if n == 1:
        return []
    if cfg.FLAG == 1:
        boxes = [tuple(int(v) for v in stats[i, :4]) for i in range(1, n)]
    else:
        x0 = min(int(stats[i, 0]) for i in range(1, n))
        y0 = min(int(stats[i, 1]) for i in range(1, n))
        x1 = max(int(stats[i, 0] + stats[i, 2]) for i in range(1, n))
        y1 = max(int(stats[i, 1] + stats[i, 3]) for i in range(1, n))
        boxes = [(x0, y0, x1 - x0, y1 - y0)]
    rects = [gating._roi(x, y, a, b, w, h, cfg.MEMSIZE, cfg.MEMSIZE, cfg) for x, y, a, b in boxes]
    return [r for r in rects if r[2] > r[0] and r[3] > r[1]]


def synthetic_sequence(seed, n_frames, height, width, step=(1.5, -0.75)):
    """``n_frames`` consecutive uint8 frames of a textured scene drifting by ``step`` px per frame."""
    pad = int(np.ceil(max(abs(step[0]), abs(step[1])) * n_frames)) + 4
    rng = np.random.default_rng(seed)
    base = rng.random((height + 2 * pad, width + 2 * pad))
    try:
        from scipy.ndimage import gaussian_filter
        base = gaussian_filter(base, 3.0, mode="nearest")
    except ImportError:                      # pragma: no cover
        base = synth._gauss_blur(base, 3.0)
    base = (base - base.min()) / (base.max() - base.min()) * 255.0
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    out = []
    for k in range(n_frames):
        f = synth._bilinear(base, yy + pad - step[1] * k, xx + pad - step[0] * k)
        out.append(np.ascontiguousarray(np.clip(np.rint(f), 0, 255).astype(np.uint8)))
    return out


def build_calls(name, stack, frames, n_pairs=None, cfg=None, with_full=True):
    """All flow calls of dataset ``name`` for pairs 0..n_pairs-1: ``stack`` = its ``constructed3DMatrix`` (rows x cols
    x slices), ``frames`` = gray uint8 frames (pair i = frames[i] -> frames[i+1]).  Returns ``(calls, canvases)``
    where ``canvases[i]`` is the float32 zero canvas the ROI flows of pair i are written into."""
    cfg = cfg or gating.dataset_config(name)
    h, w = frames[0].shape
    limit = min(len(frames) - 1, stack.shape[2] - cfg.OFFSET - (0 if cfg.bug_compatible else 1))
    n_pairs = limit if n_pairs is None else min(n_pairs, limit)
    calls, canvases = [], []
    for i in range(n_pairs):
        _, memimg2 = gating.gating_maps(stack, i, cfg)
        canvas = np.zeros((h, w, 2), np.float32)
        canvases.append(canvas)
        rects = roi_rects(memimg2, (h, w), cfg)
        # FLAG 1 boxes of one pair may overlap; the reference pastes them one after the other, so the later component
        # wins (seg.py:162).  Calls of a work list run concurrently: overlapping boxes get private fields that are
        # pasted in the reference's order once the list is done; disjoint boxes are written into the canvas directly.
        overlap = any(a[0] < b[2] and b[0] < a[2] and a[1] < b[3] and b[1] < a[3]
                      for k, a in enumerate(rects) for b in rects[k + 1:])
        for (x0, y0, x1, y1) in rects:
            view = canvas[y0:y1, x0:x1]
            calls.append(FlowCall(name, i, "roi", (x0, y0, x1, y1), cfg.farneback_params, frames[i][y0:y1, x0:x1],
                                  frames[i + 1][y0:y1, x0:x1],
                                  np.empty((y1 - y0, x1 - x0, 2), np.float32) if overlap else view,
                                  view if overlap else None))
        if with_full:
            calls.append(FlowCall(name, i, "full", (0, 0, w, h), cfg.farneback_params, frames[i], frames[i + 1],
                                  np.empty((h, w, 2), np.float32)))
    return calls, canvases


def mixed_workload(stacks, pairs_per_dataset=None, seed=2024, frames=None, datasets=None):
    """Config 4: the calls of every dataset in ``stacks`` (name -> constructed3DMatrix).  ``frames`` (name -> list of
    gray frames) defaults to synthetic sequences of the datasets' real frame sizes."""
    calls, canvases = [], {}
    for k, name in enumerate(datasets or [n for n in DATASET_FRAMES if n in stacks]):
        h, w, n_frames = DATASET_FRAMES[name]
        cfg = gating.dataset_config(name)
        n = min(n_frames - 2, stacks[name].shape[2] - cfg.OFFSET)   # the scripts walk range(len(imgs) - 2)
        if pairs_per_dataset is not None:
            n = min(n, pairs_per_dataset)
        fr = frames[name] if frames and name in frames else synthetic_sequence(seed + k, n + 1, h, w)
        c, cv = build_calls(name, stacks[name], fr, n, cfg)
        calls += c
        canvases[name] = cv
    return calls, canvases


def shard_calls(calls, rank, world):
    """Round-robin shard of the call list (BASELINE config 4: independent calls, no data-path collective)."""
    return calls[rank::world]


def run_calls(calls, ctx=None, pairs_fn=None):
    """Issue every call: one work list per Farneback parameter set.  ``pairs_fn(pairs, params, flows)`` defaults to
    the GPU work-list entry (``farneback_pairs``); tests inject a CPU stand-in.  Flows land in ``call.flow``."""
    groups = {}
    for c in calls:
        groups.setdefault(c.params, []).append(c)
    for params, group in groups.items():
        pairs = [(c.prev, c.next) for c in group]
        flows = [c.flow for c in group]
        if pairs_fn is None:
            farneback_pairs(pairs, params, flows, ctx=ctx)
        else:
            pairs_fn(pairs, params, flows)
    paste(calls)
    return calls


def paste(calls):
    """Overlapping boxes: paste the private fields into their canvases in call order (later components win)."""
    for c in calls:
        if c.paste_to is not None:
            c.paste_to[...] = c.flow


def run_calls_one_by_one(calls, ctx=None, flow_fn=None):
    """The reference's call pattern: one synchronous call per ROI / frame (``flow_fn`` defaults to the GPU
    ``calcOpticalFlowFarneback``).  Same results as ``run_calls``; kept for A/B timing and as the test reference."""
    flow_fn = flow_fn or (lambda p, q, **kw: calcOpticalFlowFarneback(p, q, None, **kw, ctx=ctx))
    for c in calls:
        c.flow[...] = flow_fn(c.prev, c.next, **c.params.as_kwargs())
        if c.paste_to is not None:
            c.paste_to[...] = c.flow
    return calls
