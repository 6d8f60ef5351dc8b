"""Analyses a rocprofv3 --kernel-trace CSV of the 3-stream bench: per-kernel durations under overlap, gaps between
consecutive kernels of a stream, how many kernels run at once."""
import csv, glob, sys, collections
import numpy as np
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "mm::" in r["Kernel_Name"]]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mm::", ""), r["Queue_Id"]) for r in rows))
# steady state: drop the first and last 20 %
n = len(ev); ev = ev[n // 5: n - n // 5]
t0, t1 = ev[0][0], max(e[1] for e in ev)
dur = collections.defaultdict(list)
for s, e, k, q in ev: dur[k].append((e - s) / 1e3)
print("window %.1f us, %d kernels, %d queues" % ((t1 - t0) / 1e3, len(ev), len(set(e[3] for e in ev))))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("  %-28s n=%4d  mean %6.1f us  p90 %6.1f  total %8.0f us" % (k, len(v), np.mean(v), np.percentile(v, 90), sum(v)))
# per-queue gaps between consecutive kernels
gaps = []
byq = collections.defaultdict(list)
for s, e, k, q in ev: byq[q].append((s, e, k))
for q, L in byq.items():
    L.sort()
    for (s0, e0, k0), (s1, e1, k1) in zip(L, L[1:]): gaps.append(((s1 - e0) / 1e3, k0, k1))
g = np.array([x[0] for x in gaps])
print("gaps between consecutive kernels of a queue: mean %.2f us, p50 %.2f, p90 %.2f, total %.0f us" % (g.mean(), np.median(g), np.percentile(g, 90), g.sum()))
by = collections.defaultdict(list)
for x, k0, k1 in gaps: by[k0 + " -> " + k1].append(x)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:9]:
    print("  %-46s mean %6.2f us (n=%d)" % (k, np.mean(v), len(v)))
# concurrency histogram
pts = sorted([(s, 1) for s, e, k, q in ev] + [(e, -1) for s, e, k, q in ev])
lvl, last, hist = 0, t0, collections.Counter()
for t, d in pts:
    hist[lvl] += t - last; last = t; lvl += d
tot = sum(hist.values())
print("kernels in flight: " + ", ".join("%d: %.0f%%" % (k, 100 * v / tot) for k, v in sorted(hist.items())))
steps = sum(1 for e in ev if e[2].startswith("vertex_bwd"))
print("steps in window: %d -> %.1f us per step" % (steps, (t1 - t0) / 1e3 / max(1, steps)))
if len(sys.argv) > 2:
    base = ev[len(ev) // 2][0]
    for s, e, k, q in ev[len(ev) // 2: len(ev) // 2 + int(sys.argv[2])]:
        print("  q%-3s %9.1f -> %9.1f  (%6.1f us)  %s" % (q, (s - base) / 1e3, (e - base) / 1e3, (e - s) / 1e3, k))
