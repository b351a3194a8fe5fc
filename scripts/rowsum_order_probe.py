#!/usr/bin/env python3
"""How far the default (per-pixel) row sums move the flow from the library's running-sum order on REAL frames, by
window size: the committed dataset frames (tests/golden/frames), parameter set B/C/A with winsize swept."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neuromorphic-spatiotemporal-optical-flow_amd")]
os.environ.setdefault("NSOF_SKIP_BUILD", "1")
import nsof  # noqa: E402
from nsof import _lib, gating  # noqa: E402
from PIL import Image  # noqa: E402

ctx = nsof.Context(0)
G = os.path.join(ROOT, "tests", "golden")


def load(p):
    return gating.frame_to_gray(np.ascontiguousarray(np.asarray(Image.open(p).convert("RGB"))[..., ::-1]), "RGB2GRAY")


sets = {"autodriving": [load(os.path.join(G, "frames", "autodriving", f"{k}.jpg")) for k in (1, 2, 3)],
        "uav": [load(os.path.join(G, "frames", "uav", f)) for f in sorted(os.listdir(os.path.join(G, "frames", "uav")), key=lambda s: int(s.split(".")[0]))[:3]],
        "tabletennis": [load(os.path.join(G, "frames", "tabletennis", f)) for f in sorted(os.listdir(os.path.join(G, "frames", "tabletennis")), key=lambda s: int(s.split(".")[0]))[:3]],
        "grasp": [load(os.path.join(G, "demo", f"grasp_{k}.jpg")) for k in (1, 2)]}
for name, fr in sets.items():
    for (ps, lv, it, pn, sg) in ((0.6, 3, 3, 10, 1.05), (0.5, 3, 3, 5, 1.2)):
        for ws in (3, 4, 5, 7, 9, 11, 13, 15):
            worst, cnt, npx = 0.0, 0, 0
            for a, b in zip(fr[:-1], fr[1:]):
                ctx.set_option(_lib.OPT_EXACT_ROWSUMS, 1)
                ex = nsof.calcOpticalFlowFarneback(a, b, None, ps, lv, ws, it, pn, sg, 0, ctx=ctx)
                ctx.set_option(_lib.OPT_EXACT_ROWSUMS, 0)
                df = nsof.calcOpticalFlowFarneback(a, b, None, ps, lv, ws, it, pn, sg, 0, ctx=ctx)
                d = np.abs(ex - df).max(-1)
                worst = max(worst, float(d.max()))
                cnt += int((d > 1e-4).sum())
                npx += d.size
            print(f"{name:12s} {a.shape} pyr {ps} poly_n {pn} winsize {ws:2d}: max |default - exact| {worst:.2e}, px > 1e-4: {cnt} of {npx}", flush=True)
ctx.close()
