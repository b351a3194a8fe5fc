#!/usr/bin/env python3
"""Where a step of k_iterate_x goes: s_memtime ticks (shader clocks on gfx950: scripts/lr_timing.py measured 23.3 per 10 ns) that one wave of each role of the workgroup with job
(pair 0, strip 1) spends working and waiting.  Needs the tuning build:
    scripts/build_variant.sh xt farneback_iterate_x.hip -DNSOF_X_TIMING
    NSOF_LIB=.../nsof/libnsof_xt.so python scripts/x_timing.py [--winsize 15] [--pairs 64]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neuromorphic-spatiotemporal-optical-flow_amd")]
os.environ.setdefault("NSOF_SKIP_BUILD", "1")
import nsof  # noqa: E402
import torch  # noqa: E402
from nsof import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--winsize", type=int, default=15)
ap.add_argument("--pairs", type=int, default=64)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
a = ap.parse_args()
dev = torch.device("cuda", 0)
ctx = nsof.Context(0)
ctx.set_option(_lib.OPT_EXACT_ROWSUMS, 1)
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
raw.nsof_debug_xtiming.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
for _ in range(2):
    ctx.check(lib.nsof_stage_iterate(ctx.ptr, n, R.data_ptr(), flow_a.data_ptr(), w, h, a.winsize, flow_b.data_ptr()))
ctx.synchronize()
raw.nsof_debug_xtiming(None, 1)
ctx.prof_enable(_lib.K_ITERATE)
ctx.check(lib.nsof_stage_iterate(ctx.ptr, n, R.data_ptr(), flow_a.data_ptr(), w, h, a.winsize, flow_b.data_ptr()))
ms, cnt = ctx.prof_collect(_lib.K_ITERATE)
out = (C.c_ulonglong * 48)()
raw.nsof_debug_xtiming(out, 0)
steps = (h + 3) // 4 + 1
names = ["consumer: column sums", "consumer: wait at barrier", "consumer: solve", "-",
         "scanner: wait for carry", "scanner: scan", "scanner: wait at barrier", "-",
         "producer rows 0,1: rows", "producer rows 0,1: wait at barrier", "-", "-",
         "remainder wave: publish + row", "remainder wave: wait at barrier", "remainder wave: fetch carry", "remainder wave: solve",
         "producer rows 2,3: rows", "producer rows 2,3: wait at barrier", "-", "-",
         "-", "solver: wait at barrier", "solver: read + solve + store", "-"]
print(f"winsize {a.winsize}, {steps} steps, launch {ms * 1e3 / cnt:.1f} us; 10-ns ticks per step")
for k, nm in enumerate(names):
    print(f"  {nm:32s} {out[k] / steps:9.1f}")
for role, sl in (("consumer", range(0, 4)), ("scanner", range(4, 8)), ("producer 0,1", range(8, 12)), ("remainder", range(12, 16)),
                 ("producer 2,3", range(16, 20)), ("solver", range(20, 24))):
    print(f"  {role} total {sum(out[k] for k in sl) / steps:9.1f}")
print("  barrier wait per wave (0-2 consumers, 3 scanner, 4-6 rows 0,1, 7 remainder, 8-10 rows 2,3):")
print("   ", " ".join(f"{out[32 + k] / steps:7.0f}" for k in range(12)))
ctx.close()
