#!/usr/bin/env python3
"""Where a step of k_iterate_q goes: shader clocks that one wave of each role of workgroup (0,0,0) spends working and
waiting at the step's two barriers.  Needs the tuning build:
    scripts/build_variant.sh qt farneback_iterate.hip -DNSOF_Q_TIMING
    NSOF_LIB=.../nsof/libnsof_qt.so python scripts/q_timing.py [--winsize 15] [--pairs 32]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neuromorphic-spatiotemporal-optical-flow_amd")]
os.environ.setdefault("NSOF_SKIP_BUILD", "1")
import nsof  # noqa: E402
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--winsize", type=int, default=15)
ap.add_argument("--pairs", type=int, default=32)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
a = ap.parse_args()
dev = torch.device("cuda", 0)
ctx = nsof.Context(0)
lib = ctx._lib
n, h, w = a.pairs, a.height, a.width
g = torch.Generator(device=dev).manual_seed(1)
img = torch.rand((2 * n, h, w), device=dev, generator=g) * 255
R = torch.empty((2 * n, 5, h, w), device=dev)
ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32), torch.arange(w, device=dev, dtype=torch.float32),
                        indexing="ij")
flow_a = torch.stack([2.5 - 0.0035 * (ys - h / 2), -1.25 + 0.0035 * (xs - w / 2)], -1)[None].repeat(n, 1, 1, 1).contiguous()
flow_b = torch.empty_like(flow_a)
torch.cuda.synchronize()
ctx.check(lib.nsof_stage_polyexp(ctx.ptr, 2 * n, img.data_ptr(), w, h, 5, 1.2, R.data_ptr()))
raw = C.CDLL(os.environ["NSOF_LIB"])
raw.nsof_debug_qtiming.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
for _ in range(2):
    ctx.check(lib.nsof_stage_iterate(ctx.ptr, n, R.data_ptr(), flow_a.data_ptr(), w, h, a.winsize, flow_b.data_ptr()))
ctx.synchronize()
raw.nsof_debug_qtiming(None, 1)
reps = 3
ctx.check(lib.nsof_stage_iterate(ctx.ptr, n, R.data_ptr(), flow_a.data_ptr(), w, h, a.winsize, flow_b.data_ptr()))
ctx.synchronize()
out = (C.c_ulonglong * 16)()
raw.nsof_debug_qtiming(out, 0)
steps = (h + 3) // 4
names = ["consumer: column sums", "consumer: wait B1", "consumer: row sums + solve", "consumer: wait B2",
         "producer A: row (B1..B2)", "producer A: wait B2", "producer A: row (B2..B1)", "producer A: wait B1",
         "producer B: row (B1..B2)", "producer B: wait B2", "producer B: row (B2..B1)", "producer B: wait B1"]
print(f"winsize {a.winsize}, {steps} steps; s_memtime ticks per step (100 MHz constant clock on gfx9: x24 for 2.4 GHz shader clocks)")
for k, nm in enumerate(names):
    print(f"  {nm:32s} {out[k] / steps:9.1f}")
for role, sl in (("consumer", range(0, 4)), ("producer A", range(4, 8)), ("producer B", range(8, 12))):
    print(f"  {role} total {sum(out[k] for k in sl) / steps:9.1f}")
ctx.close()
