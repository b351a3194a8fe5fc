#!/usr/bin/env python3
"""Accumulator with a silent voltage OUTSIDE the dead zone (every pixel integrates or leaks in every slice: the
fully dense, arithmetic-heavy case): 3840x2160, 1000 slices.  Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neuromorphic-spatiotemporal-optical-flow_amd")]
os.environ.setdefault("NSOF_SKIP_BUILD", "1")
import numpy as np  # noqa: E402
import nsof  # noqa: E402
from nsof import _lib, synth  # noqa: E402
from nsof.accumulator import Accumulator, slice_index_array  # noqa: E402


def main():
    H, W = 2160, 3840
    x, y, p, t = synth.make_events(5, W, H, 1_000_000, 1_000_000, box=(400, 300))
    idx = slice_index_array(t, 1000)
    ctx = nsof.Context(0)
    acc = Accumulator(H, W, 1, "split", -6.0, 0.4, ctx=ctx)
    acc.step(x, y, p, t, idx, snap_every=0)
    ctx.synchronize()
    acc.reset()
    ctx.prof_enable(_lib.K_ACCUM)
    t0 = time.perf_counter()
    acc.step(x, y, p, t, idx, snap_every=0)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    ms, k = ctx.prof_collect(_lib.K_ACCUM)
    n = len(idx) - 1
    print(json.dumps({"workload": f"{W}x{H}, {n} slices, scheme 1, active_v=-6, silent_v=0.4 (leak: no pixel is a no-op)",
                      "slices_per_s": round(n / dt, 1), "wall_ms": round(dt * 1e3, 2), "kernel_ms": round(ms, 2),
                      "pixel_updates_per_s": round(n * H * W / (ms * 1e-3), 0),
                      "per_slice_equivalent_gbs": round(n * H * W * 8 / (ms * 1e-3) / 1e9, 1),
                      "w_range": [float(np.min(acc.w(0))), float(np.max(acc.w(0)))]}))
    acc.close()


if __name__ == "__main__":
    main()
