#!/usr/bin/env python3
"""Exact-order fused iteration (k_iterate_x): stage-level bit-exactness against the CPU oracle on shapes with 1..N
strips, then timing next to the default kernel.   python scripts/x_check.py [--skip-time] [--pairs 64]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neuromorphic-spatiotemporal-optical-flow_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-time", action="store_true")
    ap.add_argument("--time-anyway", action="store_true", help="time the launches even if the stage check failed (timing-only ablation builds)")
    ap.add_argument("--pairs", type=int, default=64)
    a = ap.parse_args()
    os.environ.setdefault("NSOF_SKIP_BUILD", "1")
    import numpy as np
    import torch
    import nsof
    from nsof import _lib, synth
    from oracle import oracle
    oracle.build()
    dev = torch.device("cuda", 0)
    ctx = nsof.Context(0)
    lib = ctx._lib

    def rlayout(aos):
        return np.concatenate([np.ascontiguousarray(aos[..., :4]).ravel(), np.ascontiguousarray(aos[..., 4]).ravel()])

    bad = 0
    for (h, w) in [(20, 40), (135, 240), (97, 531), (64, 193), (33, 800), (270, 480)]:
        prev, nxt = synth.make_pair(9, h, w)
        I0, I1 = oracle.pyr_level(prev, 0.5, 0), oracle.pyr_level(nxt, 0.5, 0)
        R0, R1 = oracle.polyexp(I0, 5, 1.2), oracle.polyexp(I1, 5, 1.2)
        rng = np.random.default_rng(6)
        flow = (rng.standard_normal(I0.shape + (2,)) * 3).astype(np.float32)
        flow[:5, :7] += 40
        M = oracle.update_matrices(R0, R1, flow)
        Rp = np.stack([np.stack([rlayout(R0), rlayout(R1)])] * 3)
        dR = torch.from_numpy(Rp).to(dev)
        dF = torch.from_numpy(np.stack([flow] * 3)).to(dev)
        for winsize in (15, 3, 4, 2, 9, 7, 12):
            want, _ = oracle.update_flow_blur(R0, R1, flow, M, winsize, False)
            out = torch.zeros((3, h, w, 2), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            ctx.set_option(_lib.OPT_EXACT_ROWSUMS, 1)
            t0 = time.time()
            ctx.check(lib.nsof_stage_iterate(ctx.ptr, 3, dR.data_ptr(), dF.data_ptr(), w, h, winsize, out.data_ptr()))
            ctx.synchronize()
            got = out.cpu().numpy()
            same = all(np.array_equal(got[i], want) for i in range(3))
            d = float(np.abs(got[0] - want).max())
            nbad = int((got[0] != want).sum())
            print(f"{h}x{w} winsize {winsize:2d}: bit_identical={same} max_abs={d:.3g} differing={nbad} ({time.time()-t0:.3f}s)", flush=True)
            bad += 0 if same else 1
    print("STAGE_CHECK", "OK" if bad == 0 else f"FAILED {bad}", flush=True)
    if a.skip_time or (bad and not a.time_anyway):
        ctx.close()
        return 1 if bad else 0
    # ---- timing on 1080p
    n, h, w = a.pairs, 1080, 1920
    g = torch.Generator(device=dev).manual_seed(1)
    img = torch.rand((2 * n, h, w), device=dev, generator=g) * 255
    R = torch.empty((2 * n, 5, h, w), device=dev)
    ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32),
                            torch.arange(w, device=dev, dtype=torch.float32), indexing="ij")
    flow_a = torch.stack([2.5 - 0.0035 * (ys - h / 2), -1.25 + 0.0035 * (xs - w / 2)], -1)[None].repeat(n, 1, 1, 1).contiguous()
    flow_b = torch.empty_like(flow_a)
    flow_c = torch.empty_like(flow_a)
    torch.cuda.synchronize()
    ctx.check(lib.nsof_stage_polyexp(ctx.ptr, 2 * n, img.data_ptr(), w, h, 5, 1.2, R.data_ptr()))
    ctx.synchronize()
    for winsize in (15, 3):
        for mode, outb in (("default", flow_b), ("exact", flow_c)):
            ctx.set_option(_lib.OPT_EXACT_ROWSUMS, 1 if mode == "exact" else 0)
            fn = lambda: ctx.check(lib.nsof_stage_iterate(ctx.ptr, n, R.data_ptr(), flow_a.data_ptr(), w, h, winsize, outb.data_ptr()))
            fn()
            ctx.synchronize()
            ctx.prof_enable(_lib.K_ITERATE)
            for _ in range(5):
                fn()
            ms, cnt = ctx.prof_collect(_lib.K_ITERATE)
            ctx.prof_enable()
            us = ms * 1e3 / cnt
            print(f"winsize {winsize:2d} {mode:8s} {us:9.1f} us/launch ({n} pairs)  {n*h*w*56/us/1e3/8000*100:5.1f}% of 8 TB/s", flush=True)
        dd = (flow_b - flow_c).abs().max().item()
        print(f"winsize {winsize}: max |default - exact| = {dd:.3g}", flush=True)
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
