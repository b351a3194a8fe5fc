#!/bin/bash
# A/B timing of libnsof variants (scripts/build_variant.sh) on the fused iteration stage.
#   bash scripts/ab_iterate.sh name1 name2 ...     name = lib variant ("base" = the product library), optionally
#   followed by +ENV=VALUE pairs, e.g.  base+NSOF_ITER_SPLIT=1
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export NSOF_SKIP_BUILD=1
for spec in "$@"; do
  v=${spec%%+*}
  envs=""
  if [ "$spec" != "$v" ]; then envs=$(echo "${spec#*+}" | tr '+' ' '); fi
  if [ "$v" = base ]; then lib=""; else lib="NSOF_LIB=$REPO/neuromorphic-spatiotemporal-optical-flow_amd/nsof/libnsof_$v.so"; fi
  echo "== $spec"
  env $lib $envs timeout -k 10 120 python3 $REPO/scripts/stage_bench.py --stages iterate --pairs ${AB_PAIRS:-128} --reps 6 ${AB_ARGS} 2>&1 | grep -v amdgpu.ids || exit 1
done
