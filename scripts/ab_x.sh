#!/bin/bash
# A/B timing of libnsof variants (scripts/build_variant.sh) on the exact-order iteration stage: prints the exact-mode launch time
# of scripts/x_check.py for each variant ("base" = the product library).   bash scripts/ab_x.sh base v1 v2 ...
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export NSOF_SKIP_BUILD=1
for v in "$@"; do
  if [ "$v" = base ]; then unset NSOF_LIB; else export NSOF_LIB=$REPO/neuromorphic-spatiotemporal-optical-flow_amd/nsof/libnsof_$v.so; fi
  timeout -k 10 200 python3 $REPO/scripts/x_check.py --pairs ${AB_PAIRS:-256} ${AB_ARGS:-} > $REPO/gpurun_out/ab_x_$v.log 2>&1 || { echo "$v FAILED"; tail -3 $REPO/gpurun_out/ab_x_$v.log; [ -n "${AB_ARGS:-}" ] || continue; }
  echo "$v: $(grep STAGE_CHECK $REPO/gpurun_out/ab_x_$v.log) | $(grep 'exact' $REPO/gpurun_out/ab_x_$v.log | grep 'winsize 15' | sed 's/  */ /g')"
done
