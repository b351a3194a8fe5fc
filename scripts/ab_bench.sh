#!/bin/bash
# A/B of libnsof variants (scripts/build_variant.sh) on the HEADLINE bench itself (256 pairs 1080p, all levels, real pyramid
# flow), alternating the variants REPS times on the same box: box-to-box spread (+-3 %) is larger than most kernel changes.
#   bash scripts/ab_bench.sh base old c1i1 ...      (AB_PARAMS=A|B|C, AB_REPS=2)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export NSOF_SKIP_BUILD=1
for rep in $(seq 1 ${AB_REPS:-2}); do
for v in "$@"; do
  if [ "$v" = base ]; then unset NSOF_LIB; else export NSOF_LIB=$REPO/neuromorphic-spatiotemporal-optical-flow_amd/nsof/libnsof_$v.so; fi
  timeout -k 10 200 python3 $REPO/bench.py --steps 6 --warmup 2 --params ${AB_PARAMS:-A} --no-fast-leg --no-param-legs --no-config5 --e2e-pairs 0 --cpu-sample 0 \
     > $REPO/gpurun_out/ab_bench_$v.json 2> $REPO/gpurun_out/ab_bench_$v.err || { echo "$v FAILED"; tail -3 $REPO/gpurun_out/ab_bench_$v.err; continue; }
  python3 - "$v" "$REPO/gpurun_out/ab_bench_$v.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
k = d["kernel_ms_per_step"]
print(f"{sys.argv[1]:10s} {d['value']:8.1f} pairs/s  iterate {k['iterate']:7.3f}  polyexp {k['polyexp']:6.3f}  prep {k['prep']:6.3f}  ups {k['flow_upsample']:6.3f}  ms/step {d['ms_per_step']:7.3f}")
PY
done
done
