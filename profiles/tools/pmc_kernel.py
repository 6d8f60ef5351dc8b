"""Mean of one PMC counter per launch of the kernels whose name contains a pattern, from a rocprofv3 --pmc run:
   python profiles/tools/pmc_kernel.py <dir> <pattern> [label]        (counter_collection.csv: one row per dispatch and counter)"""
import csv, glob, os, sys, collections
pat, label = sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if pat in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    v = v[len(v) // 5:]                                          # (the first launches: cold caches, page faults)
    print("%s %s %s: mean %.1f over %d launches" % (label, pat, k, sum(v) / len(v), len(v)))
