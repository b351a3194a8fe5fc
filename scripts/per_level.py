#!/usr/bin/env python3
"""Per-level launch times of the bench's kernels from a rocprofv3 kernel trace (the *_kernel_trace.csv under the directory
given): launches of one kernel family are grouped by their position in the step (12 iterate launches per step = 4 levels x
3 iterations, coarsest level first).   python scripts/per_level.py gpurun_out/prof_<tag> [kernel-substring ...]"""
import csv
import glob
import json
import sys
from collections import defaultdict

d = sys.argv[1]
subs = sys.argv[2:] or ["k_iterate_x", "k_iterate_q", "k_polyexp", "k_prep", "k_flow_upsample"]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = {}
for s in subs:
    sel = [r for r in rows if s in r["Kernel_Name"]]
    if not sel:
        continue
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in sel]
    # classes by grid size (a level's launches share a grid)
    by = defaultdict(list)
    for r, u in zip(sel, dur):
        by[(r["Kernel_Name"][:60], int(r.get("Grid_Size", r.get("Grid_Size_X", 0))))].append(u)
    out[s] = {f"{k[0]} grid={k[1]}": {"n": len(v), "avg_us": round(sum(v) / len(v), 1), "min_us": round(min(v), 1), "max_us": round(max(v), 1)}
              for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))}
print(json.dumps(out, indent=1))
