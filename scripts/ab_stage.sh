#!/bin/bash
# A/B timing of libnsof variants on any stage of scripts/stage_bench.py:  AB_ARGS="--stages polyexp --poly-n 10 ..." bash scripts/ab_stage.sh base v1 v2+ENV=1
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export NSOF_SKIP_BUILD=1
for spec in "$@"; do
  v=${spec%%+*}
  envs=""
  if [ "$spec" != "$v" ]; then envs=$(echo "${spec#*+}" | tr '+' ' '); fi
  if [ "$v" = base ]; then lib=""; else lib="NSOF_LIB=$REPO/neuromorphic-spatiotemporal-optical-flow_amd/nsof/libnsof_$v.so"; fi
  echo "== $spec"
  env $lib $envs timeout -k 10 120 python3 $REPO/scripts/stage_bench.py ${AB_ARGS} 2>&1 | grep -v amdgpu.ids || exit 1
done
