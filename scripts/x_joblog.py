#!/usr/bin/env python3
"""Per-CU timeline of one k_iterate_x launch (tuning build -DNSOF_X_JOBLOG): for every (pair, strip) job the 100 MHz stamps
of start / pipeline primed / end and the CU it ran on.  Prints: launch span, busy fraction of the CUs, the gap between
consecutive jobs on a CU, start-up cost, job length by strip index and by round, the tail.
    scripts/build_variant.sh xjl farneback_iterate_x.hip -DNSOF_X_JOBLOG
    NSOF_LIB=.../nsof/libnsof_xjl.so python scripts/x_joblog.py [--winsize 15] [--pairs 256] [--size 1920x1080]"""
import argparse
import ctypes as C
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "neuromorphic-spatiotemporal-optical-flow_amd")]
os.environ.setdefault("NSOF_SKIP_BUILD", "1")
import nsof  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
from nsof import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--winsize", type=int, default=15)
ap.add_argument("--pairs", type=int, default=256)
ap.add_argument("--size", default="1920x1080")
a = ap.parse_args()
w, h = (int(v) for v in a.size.split("x"))
dev = torch.device("cuda", 0)
ctx = nsof.Context(0)
ctx.set_option(_lib.OPT_EXACT_ROWSUMS, 1)
ctx.set_option(_lib.OPT_SMALL_BATCH_JOBS, 0)
lib = ctx._lib
n = a.pairs
g = torch.Generator(device=dev).manual_seed(1)
img = torch.rand((2 * n, h, w), device=dev, generator=g) * 255
R = torch.empty((2 * n, 5, h, w), device=dev)
ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32), torch.arange(w, device=dev, dtype=torch.float32), indexing="ij")
flow_a = torch.stack([2.5 - 0.0035 * (ys - h / 2), -1.25 + 0.0035 * (xs - w / 2)], -1)[None].repeat(n, 1, 1, 1).contiguous()
flow_b = torch.empty_like(flow_a)
torch.cuda.synchronize()
ctx.check(lib.nsof_stage_polyexp(ctx.ptr, 2 * n, img.data_ptr(), w, h, 5, 1.2, R.data_ptr()))
raw = C.CDLL(os.environ["NSOF_LIB"])
raw.nsof_debug_xjoblog.argtypes = [C.c_void_p, C.POINTER(C.c_uint), C.c_int]
fn = lambda: ctx.check(lib.nsof_stage_iterate(ctx.ptr, n, R.data_ptr(), flow_a.data_ptr(), w, h, a.winsize, flow_b.data_ptr()))  # noqa: E731
for _ in range(3):
    fn()
ctx.synchronize()
raw.nsof_debug_xjoblog(None, None, 1)
ctx.prof_enable(_lib.K_ITERATE)
fn()
ms, cnt = ctx.prof_collect(_lib.K_ITERATE)
buf = np.zeros(8192 * 4, np.uint64)
cnt_jobs = C.c_uint()
raw.nsof_debug_xjoblog(buf.ctypes.data, C.byref(cnt_jobs), 0)
nj = min(cnt_jobs.value, 8192)
J = buf.reshape(-1, 4)[:nj]
t0 = int(J[:, 0].min())
start = (J[:, 0].astype(np.int64) - t0) / 100.0          # us
primed = (J[:, 1].astype(np.int64) - t0) / 100.0
end = (J[:, 2].astype(np.int64) - t0) / 100.0
meta = J[:, 3]
strip = (meta & np.uint64(0xf)).astype(int)
pair = ((meta >> np.uint64(4)) & np.uint64(0xfff)).astype(int)
hw = ((meta >> np.uint64(16)) & np.uint64(0xffffffff)).astype(np.int64)
xcc = (meta >> np.uint64(48)).astype(int)
cu = (hw >> 8) & 0xf
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
cukey = xcc * 1000 + se * 100 + sh * 20 + cu
span = float(end.max())
dur = end - start
out = {"launch_us_by_events": round(ms * 1e3 / cnt, 1), "jobs": int(nj), "span_us": round(span, 1), "cus_used": int(len(set(cukey.tolist()))),
       "job_us_mean": round(float(dur.mean()), 1), "job_us_min": round(float(dur.min()), 1), "job_us_max": round(float(dur.max()), 1),
       "startup_us_mean": round(float((primed - start).mean()), 2), "startup_us_max": round(float((primed - start).max()), 2)}
by_cu = defaultdict(list)
for i in range(nj):
    by_cu[int(cukey[i])].append(i)
gaps, busy, first_start, last_end, jobs_per_cu = [], [], [], [], []
for k, idx in by_cu.items():
    idx.sort(key=lambda i: start[i])
    busy.append(sum(dur[i] for i in idx))
    first_start.append(start[idx[0]])
    last_end.append(end[idx[-1]])
    jobs_per_cu.append(len(idx))
    for a_, b_ in zip(idx[:-1], idx[1:]):
        gaps.append(start[b_] - end[a_])
gaps = np.array(gaps) if gaps else np.zeros(1)
out.update(busy_frac_of_span=round(float(np.sum(busy) / (span * len(by_cu))), 4), gap_us_mean=round(float(gaps.mean()), 2),
           gap_us_p50=round(float(np.median(gaps)), 2), gap_us_p95=round(float(np.percentile(gaps, 95)), 2), gap_us_max=round(float(gaps.max()), 2),
           gap_negative=int((gaps < 0).sum()),
           first_start_us_max=round(float(max(first_start)), 1), last_end_us_min=round(float(min(last_end)), 1),
           jobs_per_cu_min=int(min(jobs_per_cu)), jobs_per_cu_max=int(max(jobs_per_cu)))
out["job_us_by_strip"] = {int(s_): round(float(dur[strip == s_].mean()), 1) for s_ in sorted(set(strip.tolist()))}
out["start_lag_vs_strip0_us_first_round"] = {}
for p_ in sorted(set(pair.tolist()))[:4]:
    m = pair == p_
    s0 = start[m & (strip == 0)]
    if len(s0):
        out["start_lag_vs_strip0_us_first_round"][int(p_)] = {int(s_): round(float(start[m & (strip == s_)][0] - s0[0]), 1) for s_ in sorted(set(strip[m].tolist()))}
        out.setdefault("end_lag_vs_strip0_us", {})[int(p_)] = {int(s_): round(float(end[m & (strip == s_)][0] - end[m & (strip == 0)][0]), 1) for s_ in sorted(set(strip[m].tolist()))}
order = np.argsort(start)
q = nj // 4 or 1
out["job_us_by_start_quartile"] = [round(float(dur[order[i * q:(i + 1) * q]].mean()), 1) for i in range(4)]
print(json.dumps(out, indent=1))
ctx.close()
