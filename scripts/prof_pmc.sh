#!/bin/bash
# rocprofv3 PMC pass over scripts/stage_bench.py (counters only: no trace domains next to --pmc).
# usage: bash scripts/prof_pmc.sh <tag> "<counters>" [stage_bench args...]
set -e
TAG=$1; CTRS=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export NSOF_SKIP_BUILD=1
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --output-format csv -d $REPO/gpurun_out/pmc_$TAG -- \
    python3 $REPO/scripts/stage_bench.py "$@" > $REPO/gpurun_out/pmc_$TAG.log 2>&1
F=$(find $REPO/gpurun_out/pmc_$TAG -name '*counter_collection.csv' | head -1)
python3 - "$F" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    import re
    m = re.search(r"(k_[a-z0-9_]+(<[0-9, ]+>)?)", r["Kernel_Name"])
    if m:
        k = m.group(1)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k, {c: round(v / cnt[(k, c)], 1) for c, v in acc[k].items()})
PY
