#!/bin/bash
# rocprofv3 kernel stats of lone calls (scripts/latency_single.py) under the environment given: usage
#   [SHAPES=0] NSOF_EXACT_IMPL=2k bash scripts/prof_lone_call.sh <tag>      (SHAPES: indices of latency_single's cases)
set -e
TAG=${1:-lone}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export NSOF_SKIP_BUILD=1 BANDS=0
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$TAG -- \
    python3 $REPO/scripts/latency_single.py > $REPO/gpurun_out/prof_$TAG.log 2>&1
find $REPO/gpurun_out/prof_$TAG -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $REPO/gpurun_out/prof_${TAG}_kernel_stats.csv
grep ms/call $REPO/gpurun_out/prof_$TAG.log
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$REPO/gpurun_out/prof_${TAG}_kernel_stats.csv")))[:10]:
    print("  %-70s calls %5s avg %9.1f us  %6s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
rm -rf $REPO/gpurun_out/prof_$TAG
