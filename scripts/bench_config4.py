#!/usr/bin/env python3
"""BASELINE config 4 on one GPU: every flow call (gated ROI calls + full-frame calls) of all consecutive pairs of
grasp + autodriving + uav + uavnew2 + tabletennis, each with its Parameters.txt, frames of the real sizes
(synthetic content), gating data = the reference's constructed_3D_matrix.mat stacks (tests/golden/gating_stacks.npz).

Compares the reference's call pattern (one synchronous call per ROI / frame) with the work list
(nsof_farneback_u8_batch: calls of different shapes share every launch, upload / compute / download overlap) and
prints one JSON line.  Also the device-resident ROI probe: 64 crops of 520x200 in one work list.
    python scripts/bench_config4.py [--pairs-per-dataset N] [--skip-one-by-one]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neuromorphic-spatiotemporal-optical-flow_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs-per-dataset", type=int, default=None)
    ap.add_argument("--skip-one-by-one", action="store_true")
    args = ap.parse_args()
    import torch

    import nsof
    from nsof import workload as wl
    ctx = nsof.Context(0)
    with np.load(os.path.join(ROOT, "tests", "golden", "gating_stacks.npz")) as z:
        stacks = {k: z[k] for k in z.files}
    t0 = time.perf_counter()
    calls, _ = wl.mixed_workload(stacks, pairs_per_dataset=args.pairs_per_dataset)
    t_build = time.perf_counter() - t0
    n_roi = sum(c.kind == "roi" for c in calls)
    mpx = sum(c.prev.size for c in calls) / 1e6
    out = {"workload": "config 4: grasp+autodriving+uav+uavnew2+tabletennis, gated ROI + full-frame calls",
           "calls": len(calls), "roi_calls": n_roi, "full_calls": len(calls) - n_roi, "megapixels": round(mpx, 1),
           "pairs": {n: len(set(c.pair for c in calls if c.dataset == n)) for n in wl.DATASET_FRAMES},
           "build_workload_s": round(t_build, 2)}
    wl.run_calls(calls[:8], ctx=ctx)                               # warm-up (workspace, pinned staging)
    t0 = time.perf_counter()
    wl.run_calls(calls, ctx=ctx)
    t_list = time.perf_counter() - t0
    t0 = time.perf_counter()
    wl.run_calls(calls, ctx=ctx)
    t_list = min(t_list, time.perf_counter() - t0)
    out["work_list_s"] = round(t_list, 4)
    out["work_list_calls_per_s"] = round(len(calls) / t_list, 1)
    out["work_list_mpx_per_s"] = round(mpx / t_list, 1)
    if not args.skip_one_by_one:
        keep = [c.flow.copy() for c in calls]
        wl.run_calls_one_by_one(calls[:8], ctx=ctx)
        t0 = time.perf_counter()
        wl.run_calls_one_by_one(calls, ctx=ctx)
        t_one = time.perf_counter() - t0
        out["one_by_one_s"] = round(t_one, 4)
        out["one_by_one_calls_per_s"] = round(len(calls) / t_one, 1)
        out["speedup"] = round(t_one / t_list, 2)
        out["identical"] = bool(all(np.array_equal(k, c.flow) for k, c in zip(keep, calls)))

    # device-resident ROI probe: 64 crops of 520x200 (the typical grasp ROI) of frames already in HBM
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    frames = torch.randint(0, 256, (2, 1080, 1920), dtype=torch.uint8, device=dev, generator=g)
    canvas = torch.zeros((64, 200, 520, 2), dtype=torch.float32, device=dev)
    pairs = [(frames[0, 8 * i:8 * i + 200, 13 * i:13 * i + 520], frames[1, 8 * i:8 * i + 200, 13 * i:13 * i + 520])
             for i in range(64)]
    flows = [canvas[i] for i in range(64)]
    p = nsof.farneback.PARAMS_A
    torch.cuda.synchronize()
    for n in (1, 8, 64):
        for _ in range(2):
            nsof.farneback_pairs_dev(pairs[:n], flows[:n], p, ctx=ctx)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            nsof.farneback_pairs_dev(pairs[:n], flows[:n], p, ctx=ctx)
        ctx.synchronize()
        out[f"roi_520x200_x{n}_ms_per_pair"] = round((time.perf_counter() - t0) / 10 / n * 1e3, 4)
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
