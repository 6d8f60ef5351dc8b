"""Rows of a rocprofv3 --stats kernel_stats.csv whose kernel name contains a pattern:  python profiles/tools/kstat.py <dir> <pattern> [label]"""
import csv, glob, os, sys
pat = sys.argv[2]
label = sys.argv[3] if len(sys.argv) > 3 else ""
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if pat in row["Name"]:
            print("%s %s: calls %s avg %.2f us min %.2f max %.2f" % (label, row["Name"].split("(")[0], row["Calls"], float(row["AverageNs"]) / 1e3,
                                                                   float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
