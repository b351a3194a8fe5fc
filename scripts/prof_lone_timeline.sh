#!/bin/bash
# Timeline of ONE lone call (the last of scripts/latency_single.py's calls of case SHAPES): per kernel start offset, duration and
# the idle gap before it, from rocprofv3's kernel trace.   SHAPES=0 bash scripts/prof_lone_timeline.sh <tag>
set -e
TAG=${1:-tl}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export NSOF_SKIP_BUILD=1 BANDS=0 SHAPES=${SHAPES:-0}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/prof_$TAG -- \
    python3 $REPO/scripts/latency_single.py > $REPO/gpurun_out/prof_$TAG.log 2>&1
F=$(find $REPO/gpurun_out/prof_$TAG -name '*kernel_trace.csv' | head -1)
python3 - "$F" > $REPO/gpurun_out/timeline_$TAG.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# the last call = the trailing run of kernels that starts with the first pyramid-level kernel after a long idle gap
starts = [int(r["Start_Timestamp"]) for r in rows]
ends = [int(r["End_Timestamp"]) for r in rows]
# calls are separated by host work (> 150 us idle); the LAST call is the one latency_single.py instruments with events
# (prof_enable), so the one before it is shown
cuts = [0] + [i for i in range(1, len(rows)) if starts[i] - ends[i - 1] >= 150_000] + [len(rows)]
g = max(0, len(cuts) - 3)
sel = rows[cuts[g]:cuts[g + 1]]
t0 = int(sel[0]["Start_Timestamp"])
busy = 0
prev_end = t0
print(f"{len(sel)} kernels in the call before the instrumented one")
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {r['Kernel_Name'][:70]}")
    prev_end = e
print(f"span {(prev_end - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, idle {(prev_end - t0 - busy) / 1e3:.1f} us")
PY
grep ms/call $REPO/gpurun_out/prof_$TAG.log
tail -1 $REPO/gpurun_out/timeline_$TAG.txt
rm -rf $REPO/gpurun_out/prof_$TAG
