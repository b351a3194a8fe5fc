#!/bin/bash
# rocprofv3 kernel stats of any python script of this repo:  bash scripts/prof_py_kernels.sh <tag> scripts/<x>.py [args...]
set -e
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export NSOF_SKIP_BUILD=1
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$TAG -- python3 $REPO/"$@" > $REPO/gpurun_out/prof_$TAG.log 2>&1
find $REPO/gpurun_out/prof_$TAG -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $REPO/gpurun_out/prof_${TAG}_kernel_stats.csv
grep -v "^[WE]20" $REPO/gpurun_out/prof_$TAG.log | tail -6
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$REPO/gpurun_out/prof_${TAG}_kernel_stats.csv")))[:10]:
    print("  %-70s calls %5s avg %9.1f us  tot %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
rm -rf $REPO/gpurun_out/prof_$TAG
