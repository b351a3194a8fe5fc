#!/usr/bin/env python3
"""Do the pyramid levels of a chunk of frames hit the 256 MB memory-side cache when run back to back?
512 1080p frames, pyr_scale 0.5, levels 0..3 (nsof_stage_pyr_level): level by level over all frames (what the batch
driver does) against chunk by chunk with the four level kernels of a chunk back to back.  Prints ms per 512 frames."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..",
                                               "neuromorphic-spatiotemporal-optical-flow_amd"))
import nsof  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    ctx = nsof.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    lib = ctx._lib
    n, h, w = 512, 1080, 1920
    src = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device=dev)
    sizes = [(h >> k, w >> k) for k in range(4)]
    outs = [torch.empty((n, hk, wk), dtype=torch.float32, device=dev) for hk, wk in sizes]

    def level(k, a, b):
        hk, wk = sizes[k]
        ctx.check(lib.nsof_stage_pyr_level(ctx.ptr, b - a, src[a:b].data_ptr(), w, h * w, w, h, 0.5, k,
                                           outs[k][a:b].data_ptr()))

    def run(chunk):
        for a in range(0, n, chunk):
            for k in range(4):
                level(k, a, min(a + chunk, n))

    res = {}
    for chunk in (512, 128, 64, 32, 16, 8):
        run(chunk)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            run(chunk)
        torch.cuda.synchronize()
        res[f"chunk_{chunk}_frames_ms"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
