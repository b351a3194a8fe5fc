#!/usr/bin/env python3
"""Host-to-host throughput of nsof_farneback_u8_batch on N 1080p pairs (set NSOF_PIPE_TRACE=1 for the stage timeline)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neuromorphic-spatiotemporal-optical-flow_amd")]
import nsof
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
h, w = 1080, 1920
rng = np.random.default_rng(0)
base = rng.integers(0, 256, (n + 1, h, w), dtype=np.uint8)
pairs = [(base[i], base[i + 1]) for i in range(n)]
p = nsof.farneback.PARAMS_A
ctx = nsof.Context(0)
outs = [nsof.pinned_empty((h, w, 2), np.float32) for _ in range(n)]
nsof.farneback_pairs(pairs[:32], p, outs[:32], ctx=ctx)
for rep in range(2):
    t0 = time.perf_counter()
    nsof.farneback_pairs(pairs, p, outs, ctx=ctx)
    dt = time.perf_counter() - t0
    print(f"rep {rep}: {n / dt:.1f} pairs/s ({dt * 1e3:.1f} ms)", flush=True)
