"""Table of the SURVEY 8(f) kernels from gpurun_out/f8/*kernel_stats.csv: average duration, algorithmic bytes, GB/s against 8 TB/s."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B, S, V, F, Ht, Wt = 48, 128, 642, 1280, 256, 128
MB = 1e6
# algorithmic bytes per launch: every operand read once, every result written once (fp32)
ALG = {
    "texflow_fwd": ("flow + image read, (B,3,2H,W) texture written", 4 * B * (2 * S * S + 3 * S * S + 3 * Ht * Wt)),
    "texflow_bwd": ("texture gradient + flow + image read, flow gradient written", 4 * B * (3 * Ht * Wt + 2 * S * S + 3 * S * S + 2 * S * S)),
    "mesh_reg_fwd": ("vertices + delta + face normals read, 8 partial sums", 4 * B * (2 * 3 * V + 3 * F)),
    "mesh_reg_bwd": ("the same read, three gradients written", 4 * B * (2 * 3 * V + 3 * F) * 2),
    "att_fwd": ("two attribute sets read (textures dominate)", 4 * B * 2 * (3 * Ht * Wt + 3 * V + 9 + 5)),
    "att_bwd": ("two sets read, two sets of gradients written", 4 * B * 4 * (3 * Ht * Wt + 3 * V + 9 + 5)),
    "nn": ("two clouds read, (dist, idx) written; 2 B N M = %.0f M distance evaluations per chamfer" % (2 * B * V * V / 1e6), 4 * B * (2 * 3 * V + 2 * V)),
}
rows = []
for f in glob.glob(os.path.join(ROOT, "gpurun_out", "f8", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Name"]
        for k in ALG:
            if ("mm::" + k + "_kernel") in name:
                rows.append((k, int(r["Calls"]), float(r["AverageNs"]) / 1e3))
lines = ["# SURVEY 8(f) kernels, per launch (rocprofv3 --kernel-trace --stats of profiles/tools/f8_workload.py; B=48, 128x128 image, 256x128 texture, V=642, F=1280)", "",
         "| kernel | calls | avg us | algorithmic MB | GB/s | frac of 8 TB/s | what moves |", "|---|---|---|---|---|---|---|"]
for k, calls, us in sorted(rows):
    what, nbytes = ALG[k]
    lines.append("| %s | %d | %.1f | %.2f | %.0f | %.3f | %s |" % (k, calls, us, nbytes / MB, nbytes / us / 1e3, nbytes / us / 1e3 / 8000.0, what))
out = "\n".join(lines) + "\n"
print(out)
open(os.path.join(ROOT, "gpurun_out", "f8", "f8_kernels.md"), "w").write(out)
