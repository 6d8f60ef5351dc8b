#!/usr/bin/env python3
"""profiles/tools/fetch_calib.hip under rocprofv3 -> profiles/r04_fetch_calibration.json: per access pattern, the factor that turns the
counter (KiB as rocprofv3 reports it) into the bytes the kernel is KNOWN to have moved.  On the GPU box, from the repo root:
    bash profiles/tools/fetch_calib.sh          (compiles, two --pmc passes, this script)
"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "fetch_calib")
known = json.load(open(os.path.join(src, "known.json")))
meas = {}
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        meas.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
out = {"note": "factor = known bytes / (counter KiB x 1024); arrays of 768 MiB touched exactly once (no re-use for a cache to hide); "
               "reads: FETCH_SIZE, writes: WRITE_SIZE; tool profiles/tools/fetch_calib.hip", "patterns": {}}
for k, kb in known.items():
    ctr = "WRITE_SIZE" if k.startswith("w") else "FETCH_SIZE"
    v = meas.get(k, {}).get(ctr)
    other = meas.get(k, {}).get("FETCH_SIZE" if ctr == "WRITE_SIZE" else "WRITE_SIZE")
    row = dict(kb); row["counter"] = ctr; row["counter_KiB"] = v
    if v:
        row["factor_vs_useful_bytes"] = round(kb["useful_bytes"] / (v * 1024), 4)
        row["factor_vs_array_bytes"] = round(kb["bytes_in_64B_lines"] / (v * 1024), 4)
    if other is not None:
        row["other_counter_KiB"] = other
    out["patterns"][k] = row
dst = os.path.join(os.environ.get("MM_PROFILE_OUT") or os.path.join(ROOT, "profiles"), "r04_fetch_calibration.json")
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
