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
                   silent_v=0.0, snapshot_every=33, ctx=None):
    """Run the accumulator over the stream and return, for every snapshot, the gating map and its ROI rectangles
    (x0, y0, x1, y1) in sensor pixels."""
    H, W = sensor_hw  # noqa: N806
    idx = slice_index_array(t, slice_us)
    acc = Accumulator(H, W, version, polarity, active_v, silent_v, ctx=ctx)
    try:
        acc.step(x, y, p, t, idx, snap_every=snapshot_every)
        snaps = acc.snapshots()[0]
    finally:
        acc.close()
    out = []
    for r in snaps:
        g = gating.current_to_gray(surface_to_block_current(r, cfg.MEMSIZE))
        tp = np.zeros((H // cfg.MEMSIZE, W // cfg.MEMSIZE))
        tp = gating.update_transition_pic(g, tp, cfg.THRES).astype(np.uint8)
        n, _, stats, _ = gating.connectedComponentsWithStats(tp, cfg.CONNECT)
        rects = [gating._roi(*[int(v) for v in stats[i, :4]], W, H, cfg.MEMSIZE, cfg.MEMSIZE, cfg) for i in range(1, n)]
        out.append((g, rects))
    return out


def gated_flow(gray_map, prev, nxt, cfg, flow_fn=None):
    """Flow of a frame pair restricted to the ROI(s) the gating map selects (``opticalFlow3D`` of the reference)."""
    kw = {} if flow_fn is None else {"flow_fn": flow_fn}
    return gating.opticalFlow3D(gray_map, gray_map, prev, nxt, cfg.MEMSIZE, cfg.MEMSIZE, cfg, **kw)
