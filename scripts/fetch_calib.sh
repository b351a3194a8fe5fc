#!/bin/bash
# FETCH_SIZE multipliers for 4 / 8 / 16-byte-per-lane streaming reads (scripts/fetch_calib.hip) -> gpurun_out/fetch_calibration.json
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $REPO/gpurun_out/fetch_calib
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $REPO/gpurun_out/fetch_calib -- $REPO/scripts/fetch_calib > $REPO/gpurun_out/fetch_calib.log 2>&1
python3 - "$REPO" <<'PY'
import csv, glob, json, sys, collections
repo = sys.argv[1]
f = glob.glob(f"{repo}/gpurun_out/fetch_calib/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE" and "k_stream_read" in r["Kernel_Name"]:
        w = "4" if "<unsigned int>" in r["Kernel_Name"] else "8" if "uint2" in r["Kernel_Name"] or "HIP_vector_type<unsigned int, 2" in r["Kernel_Name"] else "16"
        acc[w].append(float(r["Counter_Value"]))
known = 1 << 30
out = {"_doc": "rocprofv3 FETCH_SIZE (KiB) of one streaming pass over a 1 GiB buffer, one load of W bytes per lane (scripts/fetch_calib.hip); "
               "multiplier = known bytes / (FETCH_SIZE x 1024): what scripts/prof_traffic*.sh multiply a kernel's FETCH_SIZE with, by its access width",
       "known_bytes": known, "widths": {}}
for w, v in sorted(acc.items(), key=lambda kv: int(kv[0])):
    v = v[1:] if len(v) > 1 else v    # the first pass also fetches nothing extra, but skip it as warm-up
    m = sum(v) / len(v)
    out["widths"][w] = {"fetch_size_kib": round(m, 1), "multiplier": round(known / (m * 1024), 4), "launches": len(v)}
rows = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and "k_rows_10_per_8" in r["Kernel_Name"]]
if rows:
    fb = 517 * 1920 * 1080
    m = sum(rows[1:]) / len(rows[1:])
    out["rows_10_per_8_dword_per_lane"] = {"frame_bytes": fb, "requested_bytes": int(fb * 1.25), "fetch_size_kib": round(m, 1),
                                           "fetch_x2_over_frame_bytes": round(2 * m * 1024 / fb, 4),
                                           "note": "the level-0 pyramid kernel's read pattern: with the x2 of the streaming calibration the fabric "
                                                   "sees this multiple of the frame bytes (1.0 = the L2 absorbs the row overlap, 1.25 = none of it)"}
rows = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and "k_bytes_3_per_px" in r["Kernel_Name"]]
if rows:
    fb = 517 * 1920 * 1080
    m = sum(rows[1:]) / len(rows[1:])
    out["bytes_3_per_px"] = {"frame_bytes": fb, "fetch_size_kib": round(m, 1), "fetch_x2_over_frame_bytes": round(2 * m * 1024 / fb, 4),
                             "fetch_x1_over_frame_bytes": round(m * 1024 / fb, 4),
                             "note": "the read pattern of the vertical pass of k_polyexp_rs<.., U8> (three byte loads per row and column, "
                                     "strips overlapping by 16 of 256 columns, segments by 12 of 76 rows): which multiple of the frame bytes "
                                     "the fabric sees with the counter doubled (x2) and as it is (x1)"}
json.dump(out, open(f"{repo}/gpurun_out/fetch_calibration.json", "w"), indent=1)
print(json.dumps(out["widths"])); print(json.dumps(out.get("rows_10_per_8_dword_per_lane"))); print(json.dumps(out.get("bytes_3_per_px")))
PY
