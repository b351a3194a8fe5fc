#!/bin/bash
# HBM traffic of the BENCH's own launches (all pyramid levels, the bench's batch) from PMC counters: two rocprofv3
# passes over `bench.py --steps 2 --warmup 1` (FETCH_SIZE, then WRITE_SIZE; counters only, no trace domains).
# Per kernel family: (2 x FETCH_SIZE + WRITE_SIZE) KiB summed over the launches of the timed steps / launches, against the
# same average of the algorithmic bytes bench.py prices (gfx950 correction as in scripts/prof_traffic.sh).
# Writes gpurun_out/hbm_traffic_bench_<tag>.json.   usage: bash scripts/prof_traffic_bench.sh <tag> [bench args...]
set -e
TAG=${1:-r02}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export NSOF_SKIP_BUILD=1
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-include-regex "k_(iterate|polyexp|flow_upsample|prep)" --output-format csv -d $REPO/gpurun_out/pmcb_${TAG}_$C -- \
      python3 $REPO/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-config5 --e2e-pairs 0 --no-fast-leg --no-param-legs \
      "$@" > $REPO/gpurun_out/pmcb_${TAG}_$C.json 2> $REPO/gpurun_out/pmcb_${TAG}_$C.log
  if grep -q "Could not construct profile cfg\|exceeds the capabilities" $REPO/gpurun_out/pmcb_${TAG}_$C.log; then
    echo "rocprofv3 could not configure counter $C (see gpurun_out/pmcb_${TAG}_$C.log)" >&2; exit 1
  fi
done
python3 - "$REPO" "$TAG" <<'PY'
import csv, glob, json, re, sys, collections
repo, tag = sys.argv[1], sys.argv[2]
line = json.loads(open(f"{repo}/gpurun_out/pmcb_{tag}_FETCH_SIZE.json").read().strip().splitlines()[-1])
fam = {"k_iterate_x": "iterate", "k_iterate_q": "iterate_fast", "k_polyexp_rs": "polyexp", "k_polyexp": "polyexp", "k_flow_upsample_walk": "flow_upsample"}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{repo}/gpurun_out/pmcb_{tag}_{c}/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
        if m and m.group(1) in fam and r["Counter_Name"] == c:
            acc[fam[m.group(1)]][c].append(float(r["Counter_Value"]))
out = {"_doc": "HBM bytes per launch from rocprofv3 PMC over bench.py's OWN launches (scripts/prof_traffic_bench.sh: "
               "FETCH_SIZE and WRITE_SIZE in separate passes over `bench.py --steps 2 --warmup 1`, all pyramid levels, "
               f"{line['config']['pairs_per_gpu_per_step']} pairs per step; mean over every launch of the run, warm-up included -- "
               "the same mix of levels as the timed steps).  Counters are KiB; gfx950 correction per "
               "/opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of wide coalesced streaming reads, "
               "so it is doubled; WRITE_SIZE is exact for 16-B streaming stores.  bench.py multiplies its algorithmic "
               "bytes per launch by traffic_over_algorithmic and labels the result as an estimate.",
       "round": 4, "kernels": {}}
for name, key in (("iterate", "roofline"), ("polyexp", "roofline_polyexp")):
    v = acc.get(name)
    if not v or not v["FETCH_SIZE"] or not v["WRITE_SIZE"] or not line.get(key):
        continue
    fk = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"])
    wk = sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
    hbm = (2 * fk + wk) * 1024
    alg = line[key]["algorithmic_bytes_per_launch"]
    out["kernels"][name] = {"launches_counted": len(v["FETCH_SIZE"]), "fetch_kb_per_launch": round(fk, 1),
                            "write_kb_per_launch": round(wk, 1), "algorithmic_bytes_per_launch": alg,
                            "hbm_bytes_per_launch": int(hbm), "traffic_over_algorithmic": round(hbm / alg, 4), "fetch_multiplier": 2.0,
                            "fetch_multiplier_basis": "measured: scripts/fetch_calib.sh (profiles/r04_fetch_calibration.json)"}
json.dump(out, open(f"{repo}/gpurun_out/hbm_traffic_bench_{tag}.json", "w"), indent=1)
print(json.dumps(out["kernels"]))
PY
