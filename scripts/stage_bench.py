#!/usr/bin/env python3
"""Micro-benchmark of single pipeline stages through the C ABI (HIP-event timed inside libnsof).

    python scripts/stage_bench.py [--stages polyexp,iterate,prep,upsample] [--pairs 32] [--reps 10]

Used while tuning kernels and as the target of `rocprofv3 --pmc` runs (scripts/prof_pmc.sh)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neuromorphic-spatiotemporal-optical-flow_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stages", default="prep0,prep1,prep2,prep3,polyexp,iterate,upsample")
    ap.add_argument("--pairs", type=int, default=32)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--poly-n", type=int, default=5)
    ap.add_argument("--poly-sigma", type=float, default=1.2)
    ap.add_argument("--winsize", type=int, default=15)
    ap.add_argument("--pyr-scale", type=float, default=0.5)
    ap.add_argument("--flow", choices=["smooth", "random"], default="smooth",
                    help="flow field fed to the iteration kernel: rigid motion (as in real sequences) or white noise")
    a = ap.parse_args()
    os.environ.setdefault("NSOF_SKIP_BUILD", "1")
    import nsof
    import torch
    from nsof import _lib
    dev = torch.device("cuda", 0)
    ctx = nsof.Context(0)
    lib = ctx._lib
    n, h, w = a.pairs, a.height, a.width
    g = torch.Generator(device=dev).manual_seed(1)
    u8 = torch.randint(0, 256, (2 * n, h, w), dtype=torch.uint8, device=dev, generator=g)
    img = torch.rand((2 * n, h, w), device=dev, generator=g) * 255
    R = torch.empty((2 * n, 5, h, w), device=dev)
    if a.flow == "random":
        flow_a = (torch.randn((n, h, w, 2), device=dev, generator=g) * 2).contiguous()
    else:   # translation + slow rotation, like nsof.synth.true_flow
        ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32),
                                torch.arange(w, device=dev, dtype=torch.float32), indexing="ij")
        fu = 2.5 - 0.0035 * (ys - h / 2)
        fv = -1.25 + 0.0035 * (xs - w / 2)
        flow_a = torch.stack([fu, fv], -1)[None].repeat(n, 1, 1, 1).contiguous()
    flow_b = torch.empty_like(flow_a)
    torch.cuda.synchronize()
    ctx.check(lib.nsof_stage_polyexp(ctx.ptr, 2 * n, img.data_ptr(), w, h, a.poly_n, a.poly_sigma, R.data_ptr()))
    ctx.synchronize()
    px = h * w

    def run(name, kid, fn, bytes_per_call):
        fn()
        ctx.synchronize()
        ctx.prof_enable(kid)
        for _ in range(a.reps):
            fn()
        ms, cnt = ctx.prof_collect(kid)
        ctx.prof_enable()
        us = ms * 1e3 / cnt
        print(f"{name:10s} {us:9.1f} us/launch  {bytes_per_call / us / 1e3:8.1f} GB/s algorithmic "
              f"({bytes_per_call / us / 1e3 / 8000 * 100:5.1f}% of 8 TB/s)", flush=True)

    for st in a.stages.split(","):
        if st.startswith("prep"):
            k = int(st[4:])
            wk, hk, _, _ = nsof.level_size(w, h, a.pyr_scale, k)
            out = torch.empty((2 * n, hk, wk), device=dev)
            run(st, _lib.K_PREP, lambda: ctx.check(lib.nsof_stage_pyr_level(
                ctx.ptr, 2 * n, u8.data_ptr(), w, h * w, w, h, a.pyr_scale, k, out.data_ptr())),
                2 * n * (px + 4 * wk * hk))
        elif st == "polyexp":
            run(st, _lib.K_POLYEXP, lambda: ctx.check(lib.nsof_stage_polyexp(
                ctx.ptr, 2 * n, img.data_ptr(), w, h, a.poly_n, a.poly_sigma, R.data_ptr())), 2 * n * px * 24)
        elif st == "iterate":
            # stage API layout is pair-major [n][2][5][h][w]: the same buffer viewed that way
            run(st, _lib.K_ITERATE, lambda: ctx.check(lib.nsof_stage_iterate(
                ctx.ptr, n, R.data_ptr(), flow_a.data_ptr(), w, h, a.winsize, flow_b.data_ptr())), n * px * 56)
        elif st == "upsample":
            wk, hk, _, _ = nsof.level_size(w, h, a.pyr_scale, 1)
            src = torch.randn((n, hk, wk, 2), device=dev)
            torch.cuda.synchronize()
            run(st, _lib.K_UPSAMPLE, lambda: ctx.check(lib.nsof_stage_flow_upsample(
                ctx.ptr, n, src.data_ptr(), wk, hk, flow_b.data_ptr(), w, h, a.pyr_scale)), n * 8 * (px + wk * hk))
    ctx.close()


if __name__ == "__main__":
    main()
