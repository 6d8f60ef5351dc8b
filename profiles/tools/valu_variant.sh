#!/bin/bash
# SQ_INSTS_VALU / SQ_WAVES of every kernel for an alternative build (MM_DBG_LIB=...), forward only matters for the raster variants
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/pv; MM_DBG_LIB=/root/repo/3d-magic-mirror_amd/lib/libmm_var$v.so timeout 90 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d /tmp/pv -o p -- python /root/repo/profiles/tools/kernel_times.py > /tmp/pv.log 2>&1
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(glob.glob("/tmp/pv/*counter_collection.csv")[0])):
    agg[r["Kernel_Name"].split("(")[0].replace("void ","").replace("mm::","")][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("variant $v:", {k: int(sum(c["SQ_INSTS_VALU"]) / len(c["SQ_INSTS_VALU"])) for k, c in agg.items() if "raster" in k or "gather" in k or "pixel" in k or "walk" in k or "vertex" in k})
PY
done
