"""Per-phase cycle totals of the face sweep in gather_bwd (library built with -DMM_GATHER_PROF as lib/libmm_gprof.so:
   profiles/tools/build_variant.sh gprof mm_backward.hip -DMM_GATHER_PROF && mv lib/libmm_vargprof.so lib/libmm_gprof.so)."""
import sys, importlib, os, ctypes, torch, numpy as np
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("3d-magic-mirror_amd"); stepmod = importlib.import_module("3d-magic-mirror_amd.step")
N = pkg._native
N.LIB_PATH = "/root/repo/3d-magic-mirror_amd/lib/libmm_gprof.so"
importlib.import_module("3d-magic-mirror_amd.build_native").needs_build = lambda: False
dev = torch.device("cuda:0")
dr = pkg.DiffRender("/root/repo/tests/golden/templates/smpl_uv_642.npz", 128, emit_imnormal=False)
att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, 48, 128, 128)
datt = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}; gtd = gt.to(dev)
st = stepmod.RenderLossStep(dr, datt, gtd, fused=True)
for _ in range(4): st.run()
torch.cuda.synchronize()
L = ctypes.CDLL(N.LIB_PATH)
out = (ctypes.c_ulonglong * (16384 * 8))()
assert L.mm_debug_gather_prof(out) == 0
m = np.frombuffer(out, dtype=np.uint64).reshape(16384, 8).astype(np.float64)
m = m[m[:, 6] > 0]
names = ["setup", "sweep (loads)", "compaction", "items"]
print("%d waves; total cycles per wave: mean %.0f p50 %.0f p90 %.0f max %.0f" % (len(m), m[:, 6].mean(), np.median(m[:, 6]), np.percentile(m[:, 6], 90), m[:, 6].max()))
for i, n in enumerate(names):
    print("  %-14s mean %8.0f  p90 %8.0f  max %8.0f  (%.1f%% of wave time)" % (n, m[:, i].mean(), np.percentile(m[:, i], 90), m[:, i].max(), 100 * m[:, i].sum() / m[:, 6].sum()))
trips = m[:, 5]; items = m[:, 4]
print("  trips per wave: mean %.2f max %d; items per trip: mean %.1f; cycles per trip: sweep %.0f compaction %.0f items %.0f" % (
    trips.mean(), trips.max(), items.sum() / max(1, trips.sum()), m[:, 1].sum() / trips.sum(), m[:, 2].sum() / trips.sum(), m[:, 3].sum() / trips.sum()))
heavy = m[np.argsort(-m[:, 6])[:5]]
rest = m[:, 6] - m[:, 0] - m[:, 1] - m[:, 2] - m[:, 3]
print("  unaccounted (after the last trip): mean %.0f p90 %.0f max %.0f" % (rest.mean(), np.percentile(rest, 90), rest.max()))
for h in heavy: print("  heaviest: total %.0f setup %.0f sweep %.0f compact %.0f items %.0f | trips %d items %d nmax %d" % (h[6], h[0], h[1], h[2], h[3], h[5], h[4], h[7]))
