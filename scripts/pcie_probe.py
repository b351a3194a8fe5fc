#!/usr/bin/env python3
"""Host<->device copy rates of this box (pinned memory, hipMemcpyAsync via torch): the ceiling of every
host-to-host number (bench.py "e2e").  Prints one JSON line."""
import json
import time

import torch


def rate(fn, nbytes, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return nbytes * reps / (time.perf_counter() - t0) / 1e9


def main():
    dev = torch.device("cuda", 0)
    out = {}
    for mb in (64, 1024):
        n = mb << 20
        h = torch.empty(n, dtype=torch.uint8).pin_memory()
        h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
        d = torch.empty(n, dtype=torch.uint8, device=dev)
        d2 = torch.empty(n, dtype=torch.uint8, device=dev)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        out[f"h2d_{mb}MB_GBps"] = round(rate(lambda: d.copy_(h, non_blocking=True), n), 2)
        out[f"d2h_{mb}MB_GBps"] = round(rate(lambda: h.copy_(d, non_blocking=True), n), 2)

        def both():
            with torch.cuda.stream(s1):
                d2.copy_(h2, non_blocking=True)
            with torch.cuda.stream(s2):
                h.copy_(d, non_blocking=True)
        out[f"bidir_{mb}MB_each_GBps"] = round(rate(both, n), 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
