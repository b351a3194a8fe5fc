#!/usr/bin/env python3
"""Throughput of the segmentation head on device-resident flow fields; prints one JSON line.

    python scripts/bench_segment.py [--height 1080 --width 1920 --fields 32 --reps 20]

Algorithmic bytes per field = 8 B/px (flow read) + 1 B/px (mask write); the bit-packed intermediate is 1/64 of that.
Kernel times come from HIP events inside libnsof (nsof_prof_*)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neuromorphic-spatiotemporal-optical-flow_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--fields", type=int, default=32)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    os.environ.setdefault("NSOF_SKIP_BUILD", "1")
    import numpy as np
    import torch
    import nsof
    from nsof import _lib
    from oracle import oracle
    dev = torch.device("cuda", 0)
    ctx = nsof.Context(0)
    h, w, n = a.height, a.width, a.fields
    ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32),
                            torch.arange(w, device=dev, dtype=torch.float32), indexing="ij")
    g = torch.Generator(device=dev).manual_seed(7)
    flows = []
    for i in range(n):   # moving blobs above the threshold + speckle, like a real field
        f = torch.stack([1.4 * torch.sin(xs / (40 + i)) * torch.cos(ys / 55), 0.8 * torch.cos(xs / 31 + ys / (47 + i))], -1)
        f = f + (torch.rand((h, w, 2), device=dev, generator=g) < 0.001) * 3.0
        flows.append(f.contiguous())
    masks = [torch.empty((h, w), dtype=torch.uint8, device=dev) for _ in range(n)]
    torch.cuda.synchronize()

    def step():
        for f, m in zip(flows, masks):
            nsof.motion_mask_dev(f, m, h, w, 1, 10, 5, ctx=ctx)

    step()
    ctx.synchronize()
    want = oracle.motion_mask(flows[0].cpu().numpy(), 1.0, 10, 5)
    exact = bool(np.array_equal(masks[0].cpu().numpy(), want))
    t0 = time.perf_counter()
    for _ in range(a.reps):
        step()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / (a.reps * n)
    ctx.prof_enable(_lib.K_SEGMENT, _lib.K_MORPH)
    for _ in range(3):
        step()
    pack_ms, pack_n = ctx.prof_collect(_lib.K_SEGMENT)
    morph_ms, morph_n = ctx.prof_collect(_lib.K_MORPH)
    ctx.prof_enable()
    t0 = time.perf_counter()
    oracle.motion_mask(flows[0].cpu().numpy(), 1.0, 10, 5)
    cpu = time.perf_counter() - t0
    px = h * w
    print(json.dumps({
        "metric": "segmentation_head_fields_per_s", "value": round(1 / dt, 1), "unit": "fields/s",
        "config": {"workload": f"{h}x{w} flow -> |f|>1 -> 5x(dilate,erode) ellipse 10x10 -> u8 mask", "fields": n},
        "us_per_field": round(dt * 1e6, 2), "bit_exact_vs_oracle": exact,
        "kernels": {"mask_pack_us": round(pack_ms * 1e3 / pack_n, 2), "morph_chain_us": round(morph_ms * 1e3 / morph_n, 2)},
        "roofline": {"bound": "hbm", "achieved": round(9 * px / dt / 1e9, 1), "peak": 8000, "unit": "GB/s",
                     "frac": round(9 * px / dt / 8e12, 4),
                     "mask_pack_frac": round((8.125 * px) / (pack_ms * 1e-3 / pack_n) / 8e12, 4)},
        "cpu_baseline": {"value": round(1 / cpu, 3), "unit": "fields/s", "cores": 1, "kind": "port",
                         "sample": "1 field"}}))


if __name__ == "__main__":
    main()
