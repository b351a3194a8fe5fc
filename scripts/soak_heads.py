#!/usr/bin/env python3
"""Parity soak of the 8f kernels (segmentation head, morphology, remap, fused prediction warp, SSIM): many seeded
random cases, HIP path vs CPU oracle.  Prints one JSON summary line.

    python scripts/soak_heads.py [--cases 300] [--seed 3]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neuromorphic-spatiotemporal-optical-flow_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=150)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--big", action="store_true", help="larger images, elements up to 15x15, up to 6 iterations")
    a = ap.parse_args()
    os.environ.setdefault("NSOF_SKIP_BUILD", "1")
    import numpy as np
    import nsof
    from oracle import oracle
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    trace_path = os.path.join(ROOT, "gpurun_out", "soak_heads.trace")
    os.makedirs(os.path.dirname(trace_path), exist_ok=True)
    trace = open(trace_path, "w")

    def mark(*what):   # last line of the trace = the call in flight
        trace.write(" ".join(str(w_) for w_ in what) + "\n")
        trace.flush()
    bad = {"motion_mask": 0, "morph": 0, "remap": 0, "predict": 0}
    ssim_worst = 0.0
    for case in range(a.cases):
        if case % 10 == 0:   # the CPU oracle dominates the run time: keep the log alive
            print(f"case {case} / {a.cases}  {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
        h, w = (int(rng.integers(1, 300)), int(rng.integers(1, 420))) if a.big else (int(rng.integers(1, 240)), int(rng.integers(1, 320)))
        # --- segmentation head
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        flow = np.stack([rng.uniform(0.5, 2) * np.sin(xx / rng.uniform(5, 40)) * np.cos(yy / rng.uniform(5, 40)),
                         rng.uniform(0.2, 1.5) * np.cos(xx / rng.uniform(5, 30) + yy / rng.uniform(5, 50))], -1)
        flow = (flow + (rng.random((h, w, 2)) < 0.003) * 4).astype(np.float32)
        ks, it = (int(rng.integers(1, 16)), int(rng.integers(0, 7))) if a.big else (int(rng.integers(1, 12)), int(rng.integers(0, 6)))
        th = float(rng.choice([0.5, 1.0, 1.5]))
        mark(case, "motion_mask", h, w, th, ks, it)
        bad["motion_mask"] += not np.array_equal(nsof.motion_mask(flow, th, ks, it), oracle.motion_mask(flow, th, ks, it))
        # --- morphology with a random element / anchor
        kh, kw = int(rng.integers(1, 12)), int(rng.integers(1, 12))
        el = (rng.random((kh, kw)) < 0.6).astype(np.uint8)
        el[rng.integers(0, kh), rng.integers(0, kw)] = 1
        rows = {tuple(r) for r in el.tolist() if any(r)}
        if len(rows) <= 16:
            anchor = (int(rng.integers(0, kw)), int(rng.integers(0, kh)))
            img = np.where(rng.random((h, w)) < rng.choice([0.03, 0.5, 0.95]), 255, 0).astype(np.uint8)
            op = int(rng.integers(0, 2))
            mark(case, "morph", h, w, op, el.tolist(), anchor)
            got = (nsof.dilate if op else nsof.erode)(img, el, anchor=anchor)
            bad["morph"] += not np.array_equal(got, oracle.morph(op, img, el, anchor=anchor))
        # --- remap / prediction warp / SSIM
        sh, sw = max(h, 8), max(w, 8)
        cn = int(rng.choice([1, 3]))
        src = rng.integers(0, 256, (sh, sw) if cn == 1 else (sh, sw, cn), dtype=np.uint8)
        gx, gy = np.meshgrid(np.arange(sw, dtype=np.float32), np.arange(sh, dtype=np.float32))
        amp = float(rng.choice([0.3, 3.0, 40.0]))
        mx = (gx + rng.standard_normal((sh, sw)) * amp).astype(np.float32)
        my = (gy + rng.standard_normal((sh, sw)) * amp).astype(np.float32)
        border = int(rng.integers(0, 2))
        mark(case, "remap", sh, sw, cn, amp, border)
        bad["remap"] += not np.array_equal(nsof.remap(src, mx, my, nsof.INTER_LINEAR, borderMode=border, borderValue=5),
                                           oracle.remap_linear(src, mx, my, border, 5))
        f2 = (rng.standard_normal((sh, sw, 2)) * amp).astype(np.float32)
        x0, y0 = int(rng.integers(0, sw - 1)), int(rng.integers(0, sh - 1))
        x1, y1 = int(rng.integers(x0 + 1, sw + 1)), int(rng.integers(y0 + 1, sh + 1))
        frame = src if cn == 3 else np.repeat(src[..., None], 3, 2)
        mark(case, "predict", sh, sw, (x0, y0, x1, y1), border)
        got = nsof.predict_region(frame, f2, (x0, y0, x1, y1), sign=-1, borderMode=border)
        want = frame.copy()
        pmx, pmy = oracle.flow_map(f2, (x0, y0, x1, y1), -1)
        want[y0:y1, x0:x1] = oracle.remap_linear(frame, pmx, pmy, border)
        bad["predict"] += not np.array_equal(got, want)
        mark(case, "ssim", sh, sw)
        if sh >= 7 and sw >= 7:
            ssim_worst = max(ssim_worst, abs(nsof.structural_similarity(frame[:, :, 2], got[:, :, 2]) -
                                             oracle.ssim_u8(frame[:, :, 2], got[:, :, 2])))
    print(json.dumps({"cases": a.cases, "seed": a.seed, "mismatching_cases": bad, "ssim_worst_abs_diff": ssim_worst,
                      "seconds": round(time.time() - t0, 1)}))


if __name__ == "__main__":
    main()
