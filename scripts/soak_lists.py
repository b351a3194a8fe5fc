#!/usr/bin/env python3
"""Parity soak of the work-list path: random LISTS of crops of random shapes (the gated path's ROI calls) through
nsof_farneback_u8_batch, every crop against the CPU oracle, in both forms of the exact iteration (fused strip walker / the
three-kernel small-batch form).   python scripts/soak_lists.py [--lists 120] [--seed 5]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neuromorphic-spatiotemporal-optical-flow_amd")]
os.environ.setdefault("NSOF_SKIP_BUILD", "1")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lists", type=int, default=120)
    ap.add_argument("--seed", type=int, default=5)
    a = ap.parse_args()
    import numpy as np
    import nsof
    from nsof import _lib, synth
    from nsof.farneback import FarnebackParams, PARAMS_A, PARAMS_B, PARAMS_C
    from oracle import oracle
    oracle.build()
    rng = np.random.default_rng(a.seed)
    ctx = nsof.Context(0)
    t0 = time.time()
    crops = bad = 0
    worst = 0.0
    oracle_cache = {}
    for li in range(a.lists):
        H, W = int(rng.integers(120, 420)), int(rng.integers(160, 640))
        fa, fb = synth.make_pair(int(rng.integers(1 << 30)), H, W)
        kind = rng.integers(4)
        if kind < 3:
            params = (PARAMS_A, PARAMS_B, PARAMS_C)[kind]
        else:
            params = FarnebackParams(pyr_scale=float(rng.choice([0.5, 0.6, 0.75])), levels=int(rng.integers(0, 4)),
                                     winsize=int(rng.integers(2, 16)), iterations=int(rng.integers(1, 4)),
                                     poly_n=int(rng.choice([1, 3, 5, 7])), poly_sigma=float(rng.choice([1.05, 1.2, 1.5])), flags=0)
        n = int(rng.integers(1, 28))
        pairs = []
        for _ in range(n):
            h, w = int(rng.integers(2, H + 1)), int(rng.integers(2, W + 1))
            y0, x0 = int(rng.integers(0, H - h + 1)), int(rng.integers(0, W - w + 1))
            pairs.append((fa[y0:y0 + h, x0:x0 + w], fb[y0:y0 + h, x0:x0 + w]))
        pa = [getattr(params, k) for k in ("pyr_scale", "levels", "winsize", "iterations", "poly_n", "poly_sigma", "flags")]
        refs = [oracle.farneback(np.ascontiguousarray(p), np.ascontiguousarray(q), *pa) for p, q in pairs]
        for jobs in (0, 1 << 30):
            ctx.set_option(_lib.OPT_SMALL_BATCH_JOBS, jobs)
            flows = nsof.farneback_pairs(pairs, params, ctx=ctx)
            for f, r in zip(flows, refs):
                crops += 1
                if not np.array_equal(f, r):
                    bad += 1
                    worst = max(worst, float(np.abs(f - r).max()))
    ctx.set_option(_lib.OPT_SMALL_BATCH_JOBS, 64)
    print(json.dumps({"lists": a.lists, "seed": a.seed, "crop_results_checked": crops, "not_bit_identical": bad,
                      "worst_abs_error": worst, "forms": ["NSOF_OPT_SMALL_BATCH_JOBS=0", "NSOF_OPT_SMALL_BATCH_JOBS=2^30"],
                      "seconds": round(time.time() - t0, 1)}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
