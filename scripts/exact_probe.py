#!/usr/bin/env python3
"""Throughput of the exact row-sum order mode (NSOF_OPT_EXACT_ROWSUMS) on N 1080p pairs, per-kernel times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neuromorphic-spatiotemporal-optical-flow_amd")]
os.environ.setdefault("NSOF_SKIP_BUILD", "1")
import torch
import nsof
from nsof import _lib
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0)
prevs, nexts = bench.synth_pairs_gpu(torch, dev, n, 1080, 1920, 1234)
flow = torch.empty((n, 1080, 1920, 2), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
ctx = nsof.Context(0)
ctx.set_option(_lib.OPT_EXACT_ROWSUMS, 1)
p = nsof.farneback.PARAMS_A
nsof.farneback_batch(prevs, nexts, flow, n, 1080, 1920, p, ctx=ctx); ctx.synchronize()
ids = [_lib.K_PREP, _lib.K_POLYEXP, _lib.K_UPSAMPLE, _lib.K_UPDMAT, _lib.K_BLUR]
ctx.prof_enable(*ids)
t0 = time.perf_counter()
nsof.farneback_batch(prevs, nexts, flow, n, 1080, 1920, p, ctx=ctx); ctx.synchronize()
dt = time.perf_counter() - t0
print(f"exact mode: {n / dt:.1f} pairs/s", {_lib.load().nsof_kernel_name(k).decode(): round(ctx.prof_collect(k)[0], 2) for k in ids})
