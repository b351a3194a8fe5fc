#!/usr/bin/env python3
"""Does running sub-batches on several HIP streams (one nsof context each) overlap the VALU-bound and the
bandwidth-bound kernels?  Splits the 128-pair batch into K chunks, one context/stream per chunk."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neuromorphic-spatiotemporal-optical-flow_amd")]
os.environ.setdefault("NSOF_SKIP_BUILD", "1")
import nsof  # noqa: E402
import torch  # noqa: E402
from nsof.farneback import PARAMS_A  # noqa: E402

sys.path.insert(0, ROOT)
from bench import synth_pairs_gpu  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    n, h, w = 128, 1080, 1920
    prevs, nexts = synth_pairs_gpu(torch, dev, n, h, w, 1234)
    flow = torch.empty((n, h, w, 2), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ref = None
    for K in (1, 2, 4, 8):
        ctxs = [nsof.Context(0) for _ in range(K)]
        c = n // K

        def step():
            for i, ctx in enumerate(ctxs):
                nsof.farneback_batch(prevs[i * c:(i + 1) * c], nexts[i * c:(i + 1) * c], flow[i * c:(i + 1) * c], c, h, w,
                                     PARAMS_A, ctx=ctx)

        def sync():
            for ctx in ctxs:
                ctx.synchronize()

        step(); sync()
        if ref is None:
            ref = flow.clone()
            torch.cuda.synchronize()
        else:
            assert torch.equal(ref, flow), K
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            step()
        sync()
        dt = (time.perf_counter() - t0) / 5
        print(f"K={K}: {dt * 1e3:.2f} ms/step  {n / dt:.0f} pairs/s", flush=True)
        for ctx in ctxs:
            ctx.close()


if __name__ == "__main__":
    main()
