#!/usr/bin/env python3
"""The accumulator half of BASELINE config 5 on its own: 3840x2160 stream, dense scheme-1 update of every slice, an 8-bit surface
frame every 33 slices (what pipeline.events_to_flow_sequence does before the flow): wall time of the loop, for kernel traces."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neuromorphic-spatiotemporal-optical-flow_amd")]
os.environ.setdefault("NSOF_SKIP_BUILD", "1")
import torch  # noqa: E402

import nsof  # noqa: E402
from nsof import synth  # noqa: E402
from nsof.accumulator import Accumulator, slice_index_array  # noqa: E402

H, W, every = 2160, 3840, 33
x, y, p, t = synth.make_event_stream_4k()
ctx = nsof.Context(0)
dev = torch.device("cuda", 0)
idx = slice_index_array(t, 1000)
n_frames = (len(idx) - 1) // every
frames = torch.empty((n_frames, H, W), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
acc = Accumulator(H, W, 1, "split", -6.0, 0.0, ctx=ctx, dense=True)
acc.set_events(x, y, p, t, idx)
for rep in range(3):
    acc.reset()
    ctx.synchronize()
    t0 = time.perf_counter()
    if hasattr(acc, "run_frames") and not os.environ.get("NSOF_ACCUM_PYLOOP"):
        acc.run_frames(0, n_frames, every, frames)
    else:
        for k in range(n_frames):
            acc.run(k * every, every)
            acc.surface_u8(frames[k])
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print(f"rep {rep}: {n_frames} frames, {n_frames * every} slices in {dt * 1e3:.3f} ms = {n_frames * every / dt:.0f} slices/s", flush=True)
print("checksum", int(frames.to(torch.int64).sum().item()))
acc.close()
