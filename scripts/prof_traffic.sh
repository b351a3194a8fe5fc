#!/bin/bash
# HBM traffic per launch of the stage kernels from PMC counters: two rocprofv3 passes over scripts/stage_bench.py
# (FETCH_SIZE, then WRITE_SIZE: they do not fit one pass), counters only (no trace domains next to --pmc).
# Writes gpurun_out/hbm_traffic_<tag>.json; copy to profiles/hbm_traffic.json.
# gfx950 correction (/opt/skills/guides/MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half the bytes of wide coalesced
# streaming reads -> doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.  Units: KiB.
# usage: bash scripts/prof_traffic.sh <tag> [stage_bench args...]
set -e
TAG=${1:-r02}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export NSOF_SKIP_BUILD=1
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $REPO/gpurun_out/pmc_${TAG}_$C -- \
      python3 $REPO/scripts/stage_bench.py --pairs 32 --reps 3 "$@" > $REPO/gpurun_out/pmc_${TAG}_$C.log 2>&1
  if grep -q "Could not construct profile cfg\|exceeds the capabilities" $REPO/gpurun_out/pmc_${TAG}_$C.log; then
    echo "rocprofv3 could not configure counter $C (see gpurun_out/pmc_${TAG}_$C.log)" >&2; exit 1
  fi
done
python3 - "$REPO" "$TAG" <<'PY'
import csv, glob, json, re, sys, collections
repo, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{repo}/gpurun_out/pmc_{tag}_{c}/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
        if m and r["Counter_Name"] == c:
            acc[m.group(1)][c].append(float(r["Counter_Value"]))
px, n = 1920 * 1080, 32
alg = {"k_polyexp": 2 * n * px * 24, "k_polyexp_rs": 2 * n * px * 24, "k_iterate_q": n * px * 56, "k_iterate_pc": n * px * 56,
       "k_flow_upsample_walk": n * 8 * (px + 960 * 540), "k_prep_same3_vec": 2 * n * px * 5}
out = {"_doc": "HBM bytes per launch from rocprofv3 PMC (scripts/prof_traffic.sh: FETCH_SIZE and WRITE_SIZE in separate "
               "passes over scripts/stage_bench.py --pairs 32, 1920x1080 level-0 launches, smooth flow; median over "
               "the launches).  Counters are KiB.  gfx950 correction per /opt/skills/guides/MI355X_MICROARCH.md: "
               "FETCH_SIZE reports half the bytes of wide coalesced streaming reads, so it is doubled; WRITE_SIZE is "
               "exact for 16-B streaming stores.  bench.py scales its algorithmic bytes per launch by "
               "traffic_over_algorithmic and labels the result as an estimate.", "round": 2, "kernels": {}}
names = {"k_polyexp": "polyexp_mono", "k_polyexp_rs": "polyexp", "k_iterate_q": "iterate", "k_iterate_pc": "iterate_pc", "k_flow_upsample_walk": "flow_upsample",
         "k_prep_same3_vec": "prep"}
for k, v in acc.items():
    if k not in alg or not v["FETCH_SIZE"] or not v["WRITE_SIZE"]:
        continue
    med = lambda a: sorted(a)[len(a) // 2]
    fk, wk = med(v["FETCH_SIZE"]), med(v["WRITE_SIZE"])
    # measured multiplier (scripts/fetch_calib.sh -> profiles/r04_fetch_calibration.json): 2.0 for 4-, 8- and 16-byte-per-lane
    # streaming reads alike, and 2 x FETCH_SIZE = 1.0001 x the frame bytes for the level-0 pyramid kernel's own read pattern
    fmul = 2
    hbm = (fmul * fk + wk) * 1024
    out["kernels"][names[k]] = {"kernel": k, "fetch_kb": fk, "write_kb": wk, "algorithmic_bytes": alg[k],
                                "hbm_bytes": int(hbm), "traffic_over_algorithmic": round(hbm / alg[k], 4), "fetch_multiplier": fmul,
                                "fetch_multiplier_basis": "measured: scripts/fetch_calib.sh (profiles/r04_fetch_calibration.json)"}
json.dump(out, open(f"{repo}/gpurun_out/hbm_traffic_{tag}.json", "w"), indent=1)
print(json.dumps(out["kernels"]))
PY
