#!/usr/bin/env python3
"""Summarises gpurun_out/pmc/pass*/p_counter_collection.csv: mean counter value per kernel per launch."""
import csv, glob, collections, sys, os, json
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "pass*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mm::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in agg.items():
    if "at::" in k or "rocclr" in k: continue
    out[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    print(k)
    for c, v in sorted(out[k].items()):
        print("   %-22s %14.1f  (n=%d)" % (c, v, len(cs[c])))
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
