#!/usr/bin/env python3
"""Where the three-kernel small-batch form of the exact-order iteration stops paying: device-resident batches of n pairs, fused
strip walker (NSOF_OPT_SMALL_BATCH_JOBS=0) against the small-batch form (threshold = infinity), per shape."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neuromorphic-spatiotemporal-optical-flow_amd")]
os.environ.setdefault("NSOF_SKIP_BUILD", "1")
import numpy as np  # noqa: E402
import torch  # noqa: E402

import nsof  # noqa: E402
from nsof import _lib, synth  # noqa: E402
from nsof.farneback import PARAMS_A, PARAMS_B  # noqa: E402

ctx = nsof.Context(0)
dev = torch.device("cuda", 0)
out = []
for (h, w, p, name) in [(1080, 1920, PARAMS_A, "1080p A"), (801, 801, PARAMS_B, "801x801 B"), (200, 520, PARAMS_A, "520x200 A")]:
    a, b = synth.make_pair(3, h, w)
    strips = (w + 191) // 192
    for n in (1, 2, 4, 8, 12, 16, 24, 32, 64):
        prevs = torch.from_numpy(np.stack([a] * n)).to(dev)
        nexts = torch.from_numpy(np.stack([b] * n)).to(dev)
        flow = torch.empty((n, h, w, 2), dtype=torch.float32, device=dev)
        ms = {}
        for key, jobs in (("fused", 0), ("small", 1 << 30)):
            ctx.set_option(_lib.OPT_SMALL_BATCH_JOBS, jobs)
            for _ in range(2):
                nsof.farneback_batch(prevs, nexts, flow, n, h, w, p, ctx=ctx)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                nsof.farneback_batch(prevs, nexts, flow, n, h, w, p, ctx=ctx)
            ctx.synchronize()
            ms[key] = (time.perf_counter() - t0) / reps * 1e3
        rec = dict(shape=name, pairs=n, jobs=n * strips, fused_ms=round(ms["fused"], 3), small_ms=round(ms["small"], 3))
        out.append(rec)
        print(rec, flush=True)
ctx.set_option(_lib.OPT_SMALL_BATCH_JOBS, 64)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "small_batch_crossover.json"), "w"), indent=1)
