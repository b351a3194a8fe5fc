#!/bin/bash
# rocprofv3 PMC pass over scripts/stage_bench.py (counters only: no trace domains next to --pmc).
# usage: bash scripts/prof_pmc.sh <tag> "<counters>" [stage_bench args...]
set -e
TAG=$1; CTRS=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export NSOF_SKIP_BUILD=1
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
# One --pmc list must fit the hardware's counters per block (TCC: FETCH_SIZE costs 3 of 4, WRITE_SIZE 2; SQ: ~8):
# an over-long list aborts rocprofv3 before any kernel runs ("Request exceeds the capabilities of the hardware to
# collect").  Split such lists over several calls of this script.
set +e
rocprofv3 --pmc $CTRS --output-format csv -d $REPO/gpurun_out/pmc_$TAG -- \
    python3 $REPO/scripts/stage_bench.py "$@" > $REPO/gpurun_out/pmc_$TAG.log 2>&1
RC=$?
set -e
if grep -q "Could not construct profile cfg\|exceeds the capabilities of the hardware" $REPO/gpurun_out/pmc_$TAG.log; then
    echo "prof_pmc: rocprofv3 could not configure the counter list '$CTRS' -- too many counters for one pass; split it (log: gpurun_out/pmc_$TAG.log)" >&2
    exit 2
fi
if [ $RC -ne 0 ]; then
    echo "prof_pmc: rocprofv3 exited with status $RC (log: gpurun_out/pmc_$TAG.log)" >&2
    tail -5 $REPO/gpurun_out/pmc_$TAG.log >&2
    exit $RC
fi
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
