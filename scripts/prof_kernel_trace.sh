#!/bin/bash
# rocprofv3 kernel-trace + stats of the headline bench (run on the GPU box via gpurun).
# usage: bash scripts/prof_kernel_trace.sh <tag> [bench args...]
# (pass --no-config5 --e2e-pairs 0 --no-fast-leg --no-param-legs to profile the headline steps alone: the kernel averages
#  then cover exactly the (steps + warmup) x 12 launches the bench line's roofline is computed from)
set -e
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export NSOF_SKIP_BUILD=1            # no child processes from the profiled (GPU-initialised) process
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$TAG -- \
    python3 $REPO/bench.py --steps 5 --warmup 2 --cpu-sample 0 "$@" > $REPO/gpurun_out/prof_$TAG.log 2>&1
find $REPO/gpurun_out/prof_$TAG -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $REPO/gpurun_out/prof_${TAG}_kernel_stats.csv
tail -2 $REPO/gpurun_out/prof_$TAG.log
cat $REPO/gpurun_out/prof_${TAG}_kernel_stats.csv
