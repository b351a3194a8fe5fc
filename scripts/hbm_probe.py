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
    u = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda")   # 1 B read : 4 B written, like pyramid level 0
    out["cast_u8_to_f32_TBps_read_plus_write"] = round(5 * n / timed(lambda: torch.add(u, 0, out=y) if False else y.copy_(u)) / 1e12, 3)
    z = torch.empty((n // 4, 5), dtype=torch.float32, device="cuda")    # 4 B read : 20 B written, like polyexp
    xs = x[: n // 4]
    out["expand_1_to_5_TBps_read_plus_write"] = round(24 * (n // 4) / timed(lambda: z.copy_(xs[:, None].expand(-1, 5))) / 1e12, 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
