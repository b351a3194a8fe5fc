#!/usr/bin/env python3
"""One call with more pairs than the workspace of a single launch fits (288 GB HBM): the driver chunks by free memory."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neuromorphic-spatiotemporal-optical-flow_amd")]
os.environ.setdefault("NSOF_SKIP_BUILD", "1")
import nsof  # noqa: E402
import torch  # noqa: E402
from nsof.farneback import PARAMS_A  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2304
h, w = 1080, 1920
dev = torch.device("cuda", 0)
ctx = nsof.Context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
g = torch.Generator(device=dev).manual_seed(3)
small = torch.randint(0, 256, (64, h, w), dtype=torch.uint8, device=dev, generator=g)
frames = small.repeat((n + 64) // 64 + 1, 1, 1)[:n + 1].contiguous()          # n+1 frames
flow = torch.empty((n, h, w, 2), dtype=torch.float32, device=dev)
free, total = torch.cuda.mem_get_info()
print(f"{n} pairs: inputs {frames.numel() / 2**30:.1f} GiB, flow {flow.numel() * 4 / 2**30:.1f} GiB, free before the call "
      f"{free / 2**30:.1f} of {total / 2**30:.1f} GiB; one launch would need {n * h * w * 56 / 2**30:.1f} GiB of workspace", flush=True)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nsof.farneback_batch(frames[:-1], frames[1:], flow, n, h, w, PARAMS_A, ctx=ctx)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"rep {rep}: {n / dt:.1f} pairs/s ({dt * 1e3:.1f} ms)", flush=True)
ref = torch.empty((2, h, w, 2), dtype=torch.float32, device=dev)
nsof.farneback_batch(frames[n - 2:n], frames[n - 1:n + 1], ref, 2, h, w, PARAMS_A, ctx=ctx)
torch.cuda.synchronize()
print("last two pairs identical to a 2-pair call:", bool(torch.equal(ref, flow[n - 2:])), flush=True)
ctx.close()
