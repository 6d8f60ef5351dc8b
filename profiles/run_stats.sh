#!/bin/bash
# rocprofv3 kernel-trace + stats of the bench workload (eager launches).  Run on the GPU box from the repo root.
set -u
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/stats
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python /root/repo/bench.py --mode eager --streams 1 --cpu-seconds 0 --profile-steps 0 --steps 50 --warmup 5 ${BENCH_ARGS:-} > $OUT/bench.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/root/repo/gpurun_out/stats/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
print("%-60s %8s %12s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for r in rows[:14]:
    print("%-60s %8s %12.1f %10.2f %7.2f" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
tail -1 $OUT/bench.log | cut -c1-200
