#!/bin/bash
# A/B of libnsof variants (scripts/build_variant.sh) on lone calls: iterate milliseconds of scripts/latency_single.py per variant.
#   SHAPES=0 bash scripts/ab_lone.sh base v1 v2 ...
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export NSOF_SKIP_BUILD=1 BANDS=0
for v in "$@"; do
  if [ "$v" = base ]; then unset NSOF_LIB; else export NSOF_LIB=$REPO/neuromorphic-spatiotemporal-optical-flow_amd/nsof/libnsof_$v.so; fi
  echo "$v: $(timeout -k 10 150 python3 $REPO/scripts/latency_single.py 2>&1 | grep ms/call | sed 's/  */ /g' | cut -c1-60,100-)"
done
