"""Wall-clock timeline of the workgroups of raster_fwd / pixel_bwd / gather_bwd for one config-2 step on one stream.

Needs a debug build of the library with -DMM_TIMELINE as lib/libmm_timeline.so:
    MM_EXTRA_FLAGS=-DMM_TIMELINE python -c "import importlib,sys; sys.path.insert(0,'.'); bn=importlib.import_module('3d-magic-mirror_amd.build_native'); \\
        bn.LIB=bn.LIB.replace('libmm_render','libmm_timeline'); bn.OBJ+='_tl'; bn.build(force=True)"
Prints, per kernel: its span, when workgroups start, how long they run, how many run at once, and the last finishers --
i.e. whether the launch is bounded by its tail (a few long workgroups that started early) or by its rounds."""
import sys, importlib, os, ctypes, torch, numpy as np
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("3d-magic-mirror_amd"); stepmod = importlib.import_module("3d-magic-mirror_amd.step")
N = pkg._native
N.LIB_PATH = "/root/repo/3d-magic-mirror_amd/lib/libmm_timeline.so"
importlib.import_module("3d-magic-mirror_amd.build_native").needs_build = lambda: False
dev = torch.device("cuda:0")
dr = pkg.DiffRender("/root/repo/tests/golden/templates/smpl_uv_642.npz", 128, emit_imnormal=False)
att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, 48, 128, 128)
datt = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}; gtd = gt.to(dev)
st = stepmod.RenderLossStep(dr, datt, gtd, fused=True)
for _ in range(5): st.run()
torch.cuda.synchronize()
L = ctypes.CDLL(N.LIB_PATH)
MAXB = 16384
for name in ("raster_fwd", "pixel_bwd", "gather_bwd"):
    out = (ctypes.c_ulonglong * (MAXB * 2))()
    assert getattr(L, "mm_debug_timeline_" + name)(out) == 0
    m = np.frombuffer(out, dtype=np.uint64).reshape(MAXB, 2).astype(np.float64)
    m = m[m[:, 1] > 0]
    t0 = m[:, 0].min(); s = (m[:, 0] - t0) / 100.0; e = (m[:, 1] - t0) / 100.0
    d = e - s
    print("%s: %d workgroups, span %.1f us" % (name, len(m), e.max()))
    print("   start: p50 %.1f p90 %.1f last %.1f us | duration: mean %.1f p90 %.1f p99 %.1f max %.1f us" % (np.median(s), np.percentile(s, 90), s.max(), d.mean(), np.percentile(d, 90), np.percentile(d, 99), d.max()))
    qs = np.linspace(0, e.max(), 9)[1:-1]
    print("   running at t: " + "  ".join("%.0fus:%d" % (q, int(((s <= q) & (e > q)).sum())) for q in qs))
    late = np.argsort(-e)[:5]
    print("   last finishers (index, start, end): " + ", ".join("(%d, %.1f, %.1f)" % (int(i), s[i], e[i]) for i in late))
    if name == "gather_bwd":                                     # kind split: texture tiles come first in the grid (ntex = tiles x images)
        ntex = ((256 + 31) // 32) * ((128 + 31) // 32) * 48
        idx = np.nonzero(np.frombuffer(out, dtype=np.uint64).reshape(MAXB, 2)[:, 1] > 0)[0]
        for kind, sel in (("texture tiles", idx < ntex), ("face sweeps", idx >= ntex)):
            print("   %s: %d workgroups, start p50 %.1f p90 %.1f last %.1f | duration mean %.1f p90 %.1f max %.1f | end p50 %.1f p90 %.1f p99 %.1f last %.1f" % (
                kind, sel.sum(), np.median(s[sel]), np.percentile(s[sel], 90), s[sel].max(), d[sel].mean(), np.percentile(d[sel], 90), d[sel].max(),
                np.median(e[sel]), np.percentile(e[sel], 90), np.percentile(e[sel], 99), e[sel].max()))
