#!/usr/bin/env python3
"""Turns gpurun_out/{stats,pmc} (profiles/run_stats.sh, profiles/run_pmc.sh) into the tracked summaries:
  profiles/<tag>_kernel_stats.md   rocprofv3 --kernel-trace --stats per-kernel table
  profiles/<tag>_pmc_summary.json  mean PMC counters per kernel per launch
  profiles/traffic_latest.json     HBM bytes per launch per kernel, read by bench.py for roofline.traffic
HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports both in KiB, and on gfx950 FETCH_SIZE counts 128-B
requests as 64 B (MI355X_MICROARCH.md, HBM section).  The correction is checked in-run on recon_bwd_kernel, a pure
streaming kernel whose bytes are known (reads rgba+gt = 32 B/pixel, writes 16 B/pixel)."""
import csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
st = glob.glob(os.path.join(root, "gpurun_out", "stats", "*kernel_stats.csv"))
lines = []
if st:
    rows = [r for r in csv.DictReader(open(st[0])) if "mm::" in r["Name"]]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    calls = max(int(r["Calls"]) for r in rows)
    lines += ["# rocprofv3 --kernel-trace --stats -- python bench.py --mode eager --steps 50 --warmup 5 (config 2, B=48, 128x128)", "",
              "| kernel | calls | avg us | total us | % of path |", "|---|---|---|---|---|"]
    for r in rows:
        lines.append("| %s | %s | %.2f | %.1f | %.1f |" % (r["Name"].split("(")[0].replace("void ", ""), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                         float(r["TotalDurationNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
    lines += ["", "sum of per-step kernel time: %.1f us  (%.0f images/s at B=48 if launches were back to back)" % (tot / calls / 1e3, 48 / (tot / calls / 1e9))]
    log = os.path.join(root, "gpurun_out", "stats", "bench.log")
    if os.path.exists(log):
        js = [l for l in open(log).read().splitlines() if l.startswith("{")]
        if js:
            lines += ["", "bench line of the profiled run:", "```", js[-1], "```"]
pm = os.path.join(root, "gpurun_out", "pmc", "summary.json")
if os.path.exists(pm):
    s = json.load(open(pm))
    json.dump(s, open(os.path.join(out, tag + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
    traffic = {}
    lines += ["", "## HBM traffic per launch from PMC (separate --pmc passes; (2*FETCH_SIZE + WRITE_SIZE) KiB)", "",
              "| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM MB | L2 hit % | EA atomics |", "|---|---|---|---|---|---|"]
    for k, c in sorted(s.items()):
        if "FETCH_SIZE" not in c: continue
        b = (2 * c["FETCH_SIZE"] + c.get("WRITE_SIZE", 0)) * 1024
        name = k.replace("_kernel", "").replace("<true>", "").replace("<false>", "")
        traffic[name] = int(b)
        hit = c.get("TCC_HIT_sum", 0); miss = c.get("TCC_MISS_sum", 0)
        lines.append("| %s | %.0f | %.0f | %.2f | %.0f | %.0f |" % (k, c["FETCH_SIZE"], c.get("WRITE_SIZE", 0), b / 1e6, 100 * hit / max(1, hit + miss), c.get("TCC_EA0_ATOMIC_sum", 0)))
    # VALU issue: a wave64 vector instruction occupies its SIMD16 for 4 cycles, so a kernel cannot finish before
    # SQ_INSTS_VALU * 4 / (256 CUs * 4 SIMDs) cycles; at 2.4 GHz that bound is compared with the measured duration.
    if st:
        dur = {r["Name"].split("(")[0].replace("void ", "").replace("mm::", ""): float(r["AverageNs"]) / 1e3 for r in rows}
        lines += ["", "## Vector-instruction issue (SQ_INSTS_VALU per launch; floor = 4 cycles each over 1024 SIMDs at 2.4 GHz)", "",
                  "| kernel | waves | VALU instr | VALU/wave | SALU instr | LDS instr | issue floor us | measured us | floor/measured |", "|---|---|---|---|---|---|---|---|---|"]
        tot_floor = 0.0
        for k, c in sorted(s.items()):
            if "SQ_INSTS_VALU" not in c or k not in dur: continue
            floor = c["SQ_INSTS_VALU"] * 4 / 1024 / 2400.0
            tot_floor += floor
            lines.append("| %s | %.0f | %.0f | %.0f | %.0f | %.0f | %.1f | %.1f | %.2f |" % (
                k, c.get("SQ_WAVES", 0), c["SQ_INSTS_VALU"], c["SQ_INSTS_VALU"] / max(1, c.get("SQ_WAVES", 1)), c.get("SQ_INSTS_SALU", 0),
                c.get("SQ_INSTS_LDS", 0), floor, dur[k], floor / dur[k]))
        json.dump({"config2": {k.replace("_kernel", "").replace("<true>", "").replace("<false>", ""): c["SQ_INSTS_VALU"] for k, c in s.items()
                               if "SQ_INSTS_VALU" in c and k in dur},
                   "note": "SQ_INSTS_VALU per launch, " + tag}, open(os.path.join(out, "valu_latest.json"), "w"), indent=1, sort_keys=True)
        lines += ["", "sum of issue floors: %.1f us per step (%.0f images/s at B=48): the ceiling of this instruction mix however well launches overlap" % (
            tot_floor, 48 / (tot_floor * 1e-6))]
    if "recon_bwd" in traffic:
        lines += ["", "calibration: recon_bwd should move 48 B/pixel * 48*128*128 = %.2f MB; PMC-derived %.2f MB" % (48 * 48 * 128 * 128 / 1e6, traffic["recon_bwd"] / 1e6)]
    json.dump({"config2": traffic, "note": "(2*FETCH_SIZE+WRITE_SIZE)*1024 per launch, " + tag}, open(os.path.join(out, "traffic_latest.json"), "w"), indent=1, sort_keys=True)
open(os.path.join(out, tag + "_kernel_stats.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
