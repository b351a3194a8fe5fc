#!/usr/bin/env python3
"""Parity soak: many seeded (shape, parameter) cases of the whole Farneback call, HIP path vs CPU oracle.
Not part of the test-suite (minutes of CPU oracle time); prints one JSON summary line.

    python scripts/soak_parity.py [--cases 400] [--seed 7] [--mode default|fastrows|fast]

--mode default   the library's operation order in every stage (NSOF_OPT_EXACT_ROWSUMS=1): expected bit-identical everywhere
                 (NSOF_EXACT_IMPL=2k in the environment runs the older two-kernel form of the same order)
--mode fastrows  NSOF_OPT_EXACT_ROWSUMS=0 (per-pixel window sums: deviates at rank-deficient windows)
--mode fast      NSOF_OPT_POLYEXP_F32 (float polynomial expansion: NOT bit-identical; the summary gives the error)
A quarter of the cases are "low texture" frames (flat blocks, straight bars, a little noise: rank-deficient windows
as real footage has them), where the last bits of the sums decide the flow's 4th decimal."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neuromorphic-spatiotemporal-optical-flow_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=400)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--mode", choices=["default", "exact", "fastrows", "fast"], default="default")
    a = ap.parse_args()
    os.environ.setdefault("NSOF_SKIP_BUILD", "1")
    import numpy as np
    import nsof
    from nsof import synth
    from oracle import oracle
    rng = np.random.default_rng(a.seed)
    ctx = nsof.Context(0)
    from nsof import _lib
    if a.mode in ("default", "exact"):
        ctx.set_option(_lib.OPT_EXACT_ROWSUMS, 1)
    elif a.mode == "fastrows":
        ctx.set_option(_lib.OPT_EXACT_ROWSUMS, 0)
    elif a.mode == "fast":
        ctx.set_option(_lib.OPT_POLYEXP_F32, 1)
    above = {"1e-5": 0, "1e-4": 0, "1e-3": 0}
    worst_abs = (0.0, None)
    exact = 0
    worst = (0.0, None)
    t0 = time.time()
    kinds = {"generic": 0, "pow2_aligned": 0, "reference_sets": 0, "low_texture": 0}
    above_by_kind = {k: 0 for k in kinds}          # cases above 1e-4, by kind and by window size class
    above_by_window = {"winsize<=5": [0, 0], "6..10": [0, 0], ">=11": [0, 0]}   # [above 1e-4, cases]
    for case in range(a.cases):
        kind = ("generic", "pow2_aligned", "reference_sets", "low_texture")[case % 4]
        if kind == "pow2_aligned":     # sizes that take the exact-decimation pyramid kernels
            h, w = 8 * int(rng.integers(5, 60)), 16 * int(rng.integers(3, 40))
            p = (0.5, int(rng.integers(1, 5)), int(rng.integers(2, 18)), int(rng.integers(1, 4)),
                 int(rng.integers(1, 11)), float(rng.choice([0.0, 1.1, 1.2, 1.5])), 0)
        elif kind in ("reference_sets", "low_texture"):  # the three parameter sets of data/*/Parameters.txt on random sizes
            h, w = int(rng.integers(40, 500)), int(rng.integers(40, 640))
            p = [(0.5, 3, 15, 3, 5, 1.2, 0), (0.6, 3, 3, 3, 10, 1.05, 0), (0.6, 3, 4, 2, 1, 1.05, 0)][case // 4 % 3]
        else:
            h, w = int(rng.integers(33, 400)), int(rng.integers(33, 600))
            p = (float(rng.choice([0.5, 0.6, 0.75, 0.8, 0.9])), int(rng.integers(0, 7)), int(rng.integers(2, 40)),
                 int(rng.integers(0, 5)), int(rng.integers(1, 11)), float(rng.choice([0.0, 0.8, 1.1, 1.5, 2.0])), 0)
        kinds[kind] += 1
        prev, nxt = synth.make_pair(5000 + case, h, w, shift=(float(rng.uniform(-5, 5)), float(rng.uniform(-5, 5))),
                                    rot_deg=float(rng.uniform(-1.5, 1.5)))
        if kind == "low_texture":       # flat blocks + straight bars + a little noise, shifted by a few pixels
            base = np.full((h + 16, w + 16), float(rng.integers(20, 200)))
            for _ in range(int(rng.integers(2, 7))):
                y0, x0 = int(rng.integers(0, h)), int(rng.integers(0, w))
                base[y0:y0 + int(rng.integers(4, h // 2 + 5)), x0:x0 + int(rng.integers(4, w // 2 + 5))] = float(rng.integers(0, 256))
            for _ in range(int(rng.integers(1, 4))):
                x0 = int(rng.integers(0, w))
                base[:, x0:x0 + int(rng.integers(1, 6))] = float(rng.integers(0, 256))
            base += rng.normal(0, float(rng.choice([0.0, 0.5, 2.0])), base.shape)
            dy, dx = int(rng.integers(0, 5)), int(rng.integers(0, 5))
            to8 = lambda z: np.ascontiguousarray(np.clip(np.rint(z), 0, 255).astype(np.uint8))  # noqa: E731
            prev, nxt = to8(base[8:8 + h, 8:8 + w]), to8(base[8 - dy:8 - dy + h, 8 - dx:8 - dx + w])
        elif case % 7 == 0:               # white noise: no structure to track, large/erratic flow
            prev = rng.integers(0, 256, (h, w), dtype=np.uint8)
            nxt = rng.integers(0, 256, (h, w), dtype=np.uint8)
        got = nsof.calcOpticalFlowFarneback(prev, nxt, None, *p, ctx=ctx)
        want = oracle.farneback(prev, nxt, *p)
        err = float(np.abs(got - want).max())
        rel = err / max(1.0, float(np.abs(want).max()))
        exact += int(err == 0.0)
        for key, th in (("1e-5", 1e-5), ("1e-4", 1e-4), ("1e-3", 1e-3)):
            above[key] += int(err > th)
        wkey = "winsize<=5" if p[2] <= 5 else ("6..10" if p[2] <= 10 else ">=11")
        above_by_window[wkey][1] += 1
        if err > 1e-4:
            above_by_kind[kind] += 1
            above_by_window[wkey][0] += 1
        if err > worst_abs[0]:
            worst_abs = (err, {"case": case, "kind": kind, "shape": [h, w], "params": list(p),
                               "pixels_above_1e-4": int((np.abs(got - want).max(-1) > 1e-4).sum())})
        if rel > worst[0]:
            worst = (rel, {"case": case, "shape": [h, w], "params": list(p), "abs": err,
                           "max_flow": float(np.abs(want).max())})
    print(json.dumps({"mode": a.mode, "cases": a.cases, "seed": a.seed, "kinds": kinds, "bit_identical": exact,
                      "cases_with_max_abs_above": above, "above_1e-4_by_kind": above_by_kind,
                      "above_1e-4_by_window_[above,cases]": above_by_window,
                      "row_bands": int(os.environ.get("NSOF_ROW_BANDS", "0") or 0), "worst_abs_error": worst_abs[0], "worst_abs_case": worst_abs[1],
                      "worst_relative_error": worst[0], "worst_case": worst[1],
                      "seconds": round(time.time() - t0, 1)}))


if __name__ == "__main__":
    main()
