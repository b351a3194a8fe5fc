#!/usr/bin/env python3
"""Reads the row scan's in-kernel timing (libnsof variant built with -DNSOF_LR_TIMING; NSOF_LIB=.../libnsof_lrt.so): one lone
1080p call, then per role of workgroup 1 the cycles worked and waited per 32-column step, and the shader clock."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neuromorphic-spatiotemporal-optical-flow_amd")]
os.environ.setdefault("NSOF_SKIP_BUILD", "1")
import nsof  # noqa: E402
from nsof import _lib, synth  # noqa: E402
from nsof.farneback import PARAMS_A  # noqa: E402

lib = _lib.load()
lib.nsof_debug_lrtiming.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
ctx = nsof.Context(0)
a, b = synth.make_pair(1, 1080, 1920)
for _ in range(3):
    nsof.calcOpticalFlowFarneback(a, b, None, **PARAMS_A.as_kwargs(), ctx=ctx)
out = (C.c_ulonglong * 16)()
lib.nsof_debug_lrtiming(out, 1)
nsof.calcOpticalFlowFarneback(a, b, None, **PARAMS_A.as_kwargs(), ctx=ctx)
lib.nsof_debug_lrtiming(out, 0)
v = list(out)
steps = max(1, v[6])
print(f"launches with W >= 1024 sampled: steps {steps}")
for r, name in enumerate(("chain", "solver", "loader")):
    print(f"  {name:7s} work {v[2 * r] / steps:8.1f} cycles/step   barrier wait {v[2 * r + 1] / steps:8.1f} cycles/step")
print(f"  kernel: {v[7]} shader cycles in {v[8]} ticks of 10 ns -> {v[7] / max(1, v[8]) * 100:.0f} MHz; {v[7] / steps:.0f} cycles per step")
ctx.close()
