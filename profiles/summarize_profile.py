#!/usr/bin/env python3
"""gpurun_out/prof_<config>/{stats,pmc} (profiles/run_profile.sh) -> the tracked summaries

  profiles/<tag>_<config>_kernel_stats.md   rocprofv3 --kernel-trace --stats table, HBM traffic per launch, roofline table
                                            (algorithmic bytes vs PMC traffic vs 8 TB/s), vector-issue table, occupancy
  profiles/<tag>_<config>_pmc_summary.json  mean PMC counters per kernel per launch
  profiles/traffic_latest.json / valu_latest.json   per config: HBM bytes / SQ_INSTS_VALU per launch per kernel + the digest of
                                            the kernel sources they were measured on (bench.py refuses stale counters)

HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports both in KiB, and on gfx950 FETCH_SIZE counts 128-B requests
as 64 B (MI355X_MICROARCH.md, HBM section).  Usage: python profiles/summarize_profile.py r02 config2
"""
import re, collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (CONFIGS, algorithmic_bytes, csrc_digest, VALU_CYCLES: one definition for bench line and profiles)

tag, config = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", "prof_" + config)
out = os.environ.get("MM_PROFILE_OUT") or os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)


def short(name):
    return name.split("(")[0].replace("void ", "").replace("mm::", "")


def kname(k):
    return re.sub(r"<.*>", "", k.replace("_kernel", ""))


# ---- PMC passes --------------------------------------------------------------------------------------------------------
agg = collections.defaultdict(lambda: collections.defaultdict(list))
vgpr, lds = {}, {}
for f in sorted(glob.glob(os.path.join(src, "pmc", "pass*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if "at::" in k or "rocclr" in k or "Cijk" in k:
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "VGPR_Count" in r:
            vgpr[k] = int(float(r.get("Arch_VGPR_Count") or r["VGPR_Count"]))
        if "LDS_Block_Size" in r:
            lds[k] = int(float(r["LDS_Block_Size"]))
pmc = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
json.dump(pmc, open(os.path.join(out, "%s_%s_pmc_summary.json" % (tag, config)), "w"), indent=1, sort_keys=True)

# ---- kernel stats ------------------------------------------------------------------------------------------------------
name, B, S, ratio = bench.CONFIGS[config]
import numpy as np  # noqa: E402
tmpl = np.load(os.path.join(ROOT, "tests", "golden", "templates", name + ".npz"))
F, V = int(tmpl["faces"].shape[0]), int(tmpl["vertices"].shape[0])
H, W = round(ratio * S), S
HW, T = H * W, 2 * H * W
lines = ["# %s, %s: template %s (V=%d, F=%d), B=%d, %dx%d, texture %dx%d, one stream, fused loss" % (tag, config, name, V, F, B, H, W, 2 * H, W), ""]
dur = {}
st = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if st:
    rows = [r for r in csv.DictReader(open(st[0])) if "mm::" in r["Name"]]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    calls = max(int(r["Calls"]) for r in rows)
    lines += ["## rocprofv3 --kernel-trace --stats -- python bench.py --config %s --mode eager --streams 1 --steps 50" % config, "",
              "| kernel | calls | avg us | % of path |", "|---|---|---|---|"]
    for r in rows:
        dur[short(r["Name"])] = float(r["AverageNs"]) / 1e3
        lines.append("| %s | %s | %.2f | %.1f |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
    step_us = tot / calls / 1e3
    A = (140 * F + 36 * T + 56 * HW + 12 * HW) * B              # SURVEY 8(d)'s A + the imnormal output the step writes
    lines += ["", "sum of per-step kernel time: **%.1f us** (%.0f images/s if launches were back to back); algorithmic bytes per step "
                  "A = (140F + 36T + 56HW + 12HW imnormal)B = %.1f MB -> %.0f GB/s = **%.1f %% of 8 TB/s**" % (step_us, B / (step_us * 1e-6), A / 1e6, A / step_us / 1e3, 100 * A / step_us / 1e3 / 8000)]
    log = os.path.join(src, "stats", "bench.log")
    if os.path.exists(log):
        js = [l for l in open(log).read().splitlines() if l.startswith("{")]
        if js:
            j = json.loads(js[-1])
            lines += ["", "bench line of the profiled run: value %.0f images/s, %.4f ms/step (rocprofv3 attached)" % (j["value"], j["ms_per_step"])]

traffic, traffic_rw, valu = {}, {}, {}
if pmc:
    lines += ["", "## HBM roofline per kernel (PMC in separate --pmc passes; traffic = (2*FETCH_SIZE + WRITE_SIZE) KiB)", "",
              "| kernel | algorithmic MB/launch | avg us | achieved GB/s | frac of 8 TB/s | PMC traffic MB | traffic / algorithmic | L2 hit % |",
              "|---|---|---|---|---|---|---|---|"]
    for k, c in sorted(pmc.items()):
        if "FETCH_SIZE" not in c:
            continue
        b = (2 * c["FETCH_SIZE"] + c.get("WRITE_SIZE", 0)) * 1024
        traffic[kname(k)] = int(b)
        traffic_rw[kname(k)] = [int(2 * c["FETCH_SIZE"] * 1024), int(c.get("WRITE_SIZE", 0) * 1024)]
        alg = bench.algorithmic_bytes(kname(k), B, F, V, HW, T)
        hit, miss = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
        if alg and k in dur:
            lines.append("| %s | %.1f | %.1f | %.0f | %.3f | %.1f | %.2f | %.0f |" % (k, alg / 1e6, dur[k], alg / dur[k] / 1e3, alg / dur[k] / 1e3 / 8000,
                                                                               b / 1e6, b / alg, 100 * hit / max(1, hit + miss)))
    lines += ["", "## Vector-instruction issue and occupancy (floor = SQ_INSTS_VALU / (1024 SIMD-32 x %.1f wave-instructions/us, the sustained "
                  "rate MEASURED in profiles/r02_valu_calibration.json = 2 cycles each at ~1.75 GHz; at the nominal 2.4 GHz the floor would be x0.73)" % bench.VALU_RATE, "",
              "| kernel | waves | VGPR (rocprofv3) | LDS B/WG | VALU instr | VALU/wave | LDS instr | issue floor us | measured us | floor/measured | wave-cycles waiting % |",
              "|---|---|---|---|---|---|---|---|---|---|---|"]
    tot_floor = 0.0
    for k, c in sorted(pmc.items()):
        if "SQ_INSTS_VALU" not in c or k not in dur:
            continue
        valu[kname(k)] = c["SQ_INSTS_VALU"]
        floor = c["SQ_INSTS_VALU"] / bench.SIMDS / bench.VALU_RATE
        tot_floor += floor
        wait = 100 * c.get("SQ_WAIT_ANY", 0) / max(1.0, c.get("SQ_WAVE_CYCLES", 1))
        lines.append("| %s | %.0f | %s | %s | %.0f | %.0f | %.0f | %.1f | %.1f | %.2f | %.0f |" % (
            k, c.get("SQ_WAVES", 0), vgpr.get(k, "?"), lds.get(k, "?"), c["SQ_INSTS_VALU"], c["SQ_INSTS_VALU"] / max(1, c.get("SQ_WAVES", 1)),
            c.get("SQ_INSTS_LDS", 0), floor, dur[k], floor / dur[k], wait))
    lines += ["", "sum of issue floors: %.1f us per step" % tot_floor]


def merge(fname, key, values, note):
    path = os.path.join(out, fname)
    try:
        j = json.load(open(path))
    except Exception:
        j = {}
    if not isinstance(j.get("csrc_digest"), dict):
        j["csrc_digest"] = {}
    j[key] = values
    j["csrc_digest"][key] = bench.csrc_digest()
    j["note"] = note
    json.dump(j, open(path, "w"), indent=1, sort_keys=True)


if traffic:
    merge("traffic_latest.json", config, traffic, "(2*FETCH_SIZE+WRITE_SIZE)*1024 per launch, " + tag)
    merge("traffic_latest.json", config + "_rw", traffic_rw, "(2*FETCH_SIZE+WRITE_SIZE)*1024 per launch, " + tag)
if valu:
    merge("valu_latest.json", config, valu, "SQ_INSTS_VALU per launch, " + tag)
open(os.path.join(out, "%s_%s_kernel_stats.md" % (tag, config)), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
