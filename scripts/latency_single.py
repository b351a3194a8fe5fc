#!/usr/bin/env python3
"""Single-call latency of the cv2-style entry point (host numpy in, host numpy out), as the reference's scripts use it."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neuromorphic-spatiotemporal-optical-flow_amd")]
os.environ.setdefault("NSOF_SKIP_BUILD", "1")
import nsof  # noqa: E402
from nsof import synth  # noqa: E402
from nsof.farneback import PARAMS_A, PARAMS_B  # noqa: E402

import numpy as np  # noqa: E402
from nsof import _lib  # noqa: E402

ctx = nsof.Context(0)
BANDS = [int(v) for v in os.environ.get("BANDS", "0,1").split(",")]   # values of NSOF_OPT_ROW_BANDS to time
CASES = [(1080, 1920, "1080p full frame, params A", PARAMS_A), (200, 520, "520x200 ROI, params A", PARAMS_A),
         (801, 801, "801x801 autodriving frame, params B", PARAMS_B), (161, 161, "161x161 uav, params B", PARAMS_B)]
if os.environ.get("SHAPES"):
    CASES = [CASES[int(v)] for v in os.environ["SHAPES"].split(",")]
for (h, w, name, p) in CASES:
  prev, nxt = synth.make_pair(1, h, w)
  kw = p.as_kwargs()
  base = None
  for bands in BANDS:
    ctx.set_option(_lib.OPT_ROW_BANDS, bands)
    flow = nsof.calcOpticalFlowFarneback(prev, nxt, None, **kw, ctx=ctx)
    if base is None:
        base = flow
    dev = float(np.abs(flow - base).max())
    for _ in range(3):
        nsof.calcOpticalFlowFarneback(prev, nxt, None, **kw, ctx=ctx)
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        nsof.calcOpticalFlowFarneback(prev, nxt, None, **kw, ctx=ctx)
    dt = (time.perf_counter() - t0) / n
    ids = [_lib.K_PREP, _lib.K_POLYEXP, _lib.K_UPSAMPLE, _lib.K_UPDMAT, _lib.K_BLUR, _lib.K_ITERATE]
    ctx.prof_enable(*ids)
    nsof.calcOpticalFlowFarneback(prev, nxt, None, **kw, ctx=ctx)
    parts = {ctx._lib.nsof_kernel_name(k).decode(): round(ctx.prof_collect(k)[0], 3) for k in ids}
    ctx.prof_enable()
    print(f"{name:40s} bands={bands:<3d} {dt * 1e3:8.2f} ms/call   max|d| vs bands=0: {dev:.2e}   kernels(ms): {parts}", flush=True)
ctx.close()
