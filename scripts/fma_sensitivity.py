#!/usr/bin/env python3
"""How far the flow moves between the two arithmetic variants of the pyramid stages (NSOF_OPT_PYR_FMA 0 / 1: every product
and sum rounded, vs one fused multiply-add per tap / blend as an AVX2+FMA3 build of the library contracts them) -- the band
inside which a real cv2 wheel may lie whichever of the two it executes.  GPU vs GPU (each variant is bit-identical to its
CPU oracle build: tests/test_farneback_gpu.py::test_pyramid_fma_variant_twin).  Prints one JSON line.
    python scripts/fma_sensitivity.py [--pairs 64]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neuromorphic-spatiotemporal-optical-flow_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=64)
    a = ap.parse_args()
    os.environ.setdefault("NSOF_SKIP_BUILD", "1")
    import numpy as np
    import torch
    import nsof
    from nsof import _lib, gating, synth
    from nsof.farneback import PARAMS_A, PARAMS_B, PARAMS_C
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda", 0)
    ctx = nsof.Context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    out = {"option": "NSOF_OPT_PYR_FMA 0 vs 1 (GPU vs GPU; each variant bit-identical to its oracle build)", "cases": {}}

    def both(fn):
        res = []
        for fma in (0, 1):
            ctx.set_option(_lib.OPT_PYR_FMA, fma)
            res.append(fn())
        ctx.set_option(_lib.OPT_PYR_FMA, 0)
        return res

    n, h, w = a.pairs, 1080, 1920
    prevs, nexts = bench.synth_pairs_gpu(torch, dev, n, h, w, 1234)
    for name, p in (("A", PARAMS_A), ("B", PARAMS_B), ("C", PARAMS_C)):
        def run(p=p):
            f = torch.empty((n, h, w, 2), dtype=torch.float32, device=dev)
            nsof.farneback_batch(prevs, nexts, f, n, h, w, p, ctx=ctx)
            torch.cuda.synchronize()
            return f
        f0, f1 = both(run)
        d = (f0 - f1).abs().amax(dim=-1)
        out["cases"][f"bench_1080p_params_{name}"] = {"pairs": n, "max_abs": float(d.max().item()),
                                                      "pixels_above_1e-4": int((d > 1e-4).sum().item()),
                                                      "pixels": int(d.numel()),
                                                      "pairs_with_max_above_1e-4": int((d.amax(dim=(1, 2)) > 1e-4).sum().item())}
        del f0, f1, d
    try:
        from PIL import Image
        G = os.path.join(ROOT, "tests", "golden")

        def gray(path):
            return gating.frame_to_gray(np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[..., ::-1]), "RGB2GRAY")
        real = {"autodriving_801x801_params_B": ([os.path.join(G, "frames", "autodriving", f"{k}.jpg") for k in (1, 2)], PARAMS_B),
                "grasp_1080x1920_params_A": ([os.path.join(G, "demo", f"grasp_{k}.jpg") for k in (1, 2)], PARAMS_A)}
        for name, (paths, p) in real.items():
            fa, fb = gray(paths[0]), gray(paths[1])
            g0, g1 = both(lambda: nsof.calcOpticalFlowFarneback(fa, fb, None, **p.as_kwargs(), ctx=ctx))
            d = np.abs(g0 - g1).max(-1)
            out["cases"][name] = {"max_abs": float(d.max()), "pixels_above_1e-4": int((d > 1e-4).sum()), "pixels": int(d.size)}
    except ImportError:
        pass
    # a soak-like mix of small shapes and random parameters
    rng = np.random.default_rng(11)
    worst, above, cases = 0.0, 0, 300
    for k in range(cases):
        hh, ww = int(rng.integers(40, 300)), int(rng.integers(40, 400))
        p = (float(rng.choice([0.5, 0.6, 0.75])), int(rng.integers(0, 4)), int(rng.integers(2, 16)), int(rng.integers(1, 4)),
             int(rng.choice([1, 5, 7, 10])), float(rng.choice([1.05, 1.2, 1.5])), 0)
        fa, fb = synth.make_pair(1000 + k, hh, ww)
        g0, g1 = both(lambda: nsof.calcOpticalFlowFarneback(fa, fb, None, *p, ctx=ctx))
        dd = float(np.abs(g0 - g1).max())
        worst = max(worst, dd)
        above += dd > 1e-4
    out["cases"]["random_small_shapes_and_parameters"] = {"cases": cases, "worst_max_abs": worst, "cases_above_1e-4": int(above)}
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
