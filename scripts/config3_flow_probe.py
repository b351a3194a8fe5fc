#!/usr/bin/env python3
"""The flow half of BASELINE config 3 on the sparse stream (6380 ROI crops of ~60x60 px in one work list): wall time of the
surface + gating stage and of the flow stage, for kernel traces (bash scripts/prof_py_kernels.sh <tag> scripts/config3_flow_probe.py)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neuromorphic-spatiotemporal-optical-flow_amd")]
os.environ.setdefault("NSOF_SKIP_BUILD", "1")
import nsof  # noqa: E402
from nsof import gating, pipeline, synth  # noqa: E402
from nsof.farneback import PARAMS_A  # noqa: E402

H, W, every, ms = 720, 1280, 33, 20
cfg = gating.GatingConfig(MEMSIZE=ms, EXTEND_HEIGHT_UPPER=20, EXTEND_HEIGHT_LOWER=20, EXTEND_WIDTH_LEFT=20, EXTEND_WIDTH_RIGHT=20,
                          THRES=240, FLAG=1, farneback_params=PARAMS_A)
n_bg = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
x, y, p, t = synth.make_events(2024, W, H, n_background=n_bg)
ctx = nsof.Context(0)
out = {}
for rep in range(3):
    tm = {}
    frames, rects, flows = pipeline.events_to_roi_flows(x, y, p, t, (H, W), cfg, slice_us=1000, active_v=-6.0, silent_v=0.5,
                                                        snapshot_every=every, ctx=ctx, timings=tm, max_rects=256)
    out = {k: (round(v * 1e3, 3) if k.endswith("_s") else v) for k, v in tm.items()}
print(json.dumps(out))
ctx.close()
