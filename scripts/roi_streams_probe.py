#!/usr/bin/env python3
"""Small ROI crops (the gated path of the reference) do not fill 256 CUs: how much do K contexts (one HIP stream each)
overlap their pipelines?  64 device-resident ROI pairs of 520x200 (params A) and of 161x161 (params B)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neuromorphic-spatiotemporal-optical-flow_amd")]
os.environ.setdefault("NSOF_SKIP_BUILD", "1")
import nsof  # noqa: E402
import torch  # noqa: E402
from nsof.farneback import PARAMS_A, PARAMS_B  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    for (h, w, p, name) in ((200, 520, PARAMS_A, "520x200 A"), (161, 161, PARAMS_B, "161x161 B"), (1140, 760, PARAMS_A, "760x1140 A")):
        n = 64
        g = torch.Generator(device=dev).manual_seed(1)
        prev = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device=dev, generator=g)
        nxt = torch.roll(prev, 2, 2)
        flow = torch.empty((n, h, w, 2), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        ref = None
        for K in (1, 2, 4, 8):
            ctxs = [nsof.Context(0) for _ in range(K)]

            def run():
                for i in range(n):   # one call per ROI pair, as the gated path issues them
                    nsof.farneback_batch(prev[i:i + 1], nxt[i:i + 1], flow[i:i + 1], 1, h, w, p, ctx=ctxs[i % K])
                for c in ctxs:
                    c.synchronize()

            run()
            if ref is None:
                ref = flow.clone()
                torch.cuda.synchronize()
            else:
                assert torch.equal(ref, flow)
            t0 = time.perf_counter()
            for _ in range(3):
                run()
            dt = (time.perf_counter() - t0) / (3 * n)
            print(f"{name}: K={K}: {dt * 1e3:.3f} ms per ROI pair", flush=True)
            for c in ctxs:
                c.close()


if __name__ == "__main__":
    main()
