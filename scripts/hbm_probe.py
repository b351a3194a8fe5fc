#!/usr/bin/env python3
"""What the HBM sustains for pure-write, pure-read and copy streams on this GPU (torch elementwise kernels, 1 GiB
buffers, HIP-event timed) -- the practical ceilings the stage kernels are compared with in DESIGN.md."""
import json

import torch


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def main():
    n = 1 << 28   # 1 GiB of float32
    x = torch.empty(n, dtype=torch.float32, device="cuda")
    y = torch.empty_like(x)
    x.fill_(1.0)
    out = {"bytes": 4 * n}
    out["write_fill_TBps"] = round(4 * n / timed(lambda: x.fill_(2.0)) / 1e12, 3)
    out["read_sum_TBps"] = round(4 * n / timed(lambda: x.sum()) / 1e12, 3)
    out["copy_TBps_read_plus_write"] = round(8 * n / timed(lambda: y.copy_(x)) / 1e12, 3)
    out["scale_TBps_read_plus_write"] = round(8 * n / timed(lambda: torch.mul(x, 2.0, out=y)) / 1e12, 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
