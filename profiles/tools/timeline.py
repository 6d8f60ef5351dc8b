"""Wall-clock timeline of the workgroups of raster_fwd / pixel_bwd / gather_bwd for one step on one stream:
    python profiles/tools/timeline.py [config2|config3|config5]
Builds a debug variant of the library with -DMM_TIMELINE as lib/libmm_timeline.so (the product library is untouched).
Prints, per kernel: its span, when workgroups start, how long they run, how many run at once, and the last finishers --
i.e. whether the launch is bounded by its tail (a few long workgroups that started early) or by its rounds."""
import sys, importlib, os, ctypes, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "config2"
name, B, S, ratio = bench.CONFIGS[cfg]
bn = importlib.import_module("3d-magic-mirror_amd.build_native")
var = os.path.join(os.path.dirname(bn.LIB), "libmm_timeline.so")
bn.build(out=var, extra_flags=["-DMM_TIMELINE"])
pkg = importlib.import_module("3d-magic-mirror_amd"); stepmod = importlib.import_module("3d-magic-mirror_amd.step")
N = pkg._native
N.LIB_PATH = var
bn.needs_build = lambda: False
dev = torch.device("cuda:0")
dr = pkg.DiffRender(os.path.join(ROOT, "tests", "golden", "templates", name + ".npz"), S, ratio=ratio, emit_imnormal=False)
H, W = dr.render_height, dr.image_size
att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, B, H, W)
datt = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}; gtd = gt.to(dev)
st = stepmod.RenderLossStep(dr, datt, gtd, fused=True)
for _ in range(5): st.run()
torch.cuda.synchronize()
L = ctypes.CDLL(var)
MAXB = 81920
print("== %s: %s B=%d %dx%d" % (cfg, name, B, H, W))
for kn in ("raster_fwd", "pixel_bwd", "gather_bwd"):
    out = (ctypes.c_ulonglong * (MAXB * 2))()
    assert getattr(L, "mm_debug_timeline_" + kn)(out) == 0
    raw = np.frombuffer(out, dtype=np.uint64).reshape(MAXB, 2)
    m = raw.astype(np.float64)
    live = m[:, 1] > 0
    m = m[live]
    t0 = m[:, 0].min(); s = (m[:, 0] - t0) / 100.0; e = (m[:, 1] - t0) / 100.0
    d = e - s
    print("%s: %d workgroups recorded (first %d of the grid), span %.1f us" % (kn, len(m), MAXB, e.max()))
    print("   start: p50 %.1f p90 %.1f last %.1f us | duration: mean %.1f p90 %.1f p99 %.1f max %.1f us" % (np.median(s), np.percentile(s, 90), s.max(), d.mean(), np.percentile(d, 90), np.percentile(d, 99), d.max()))
    qs = np.linspace(0, e.max(), 17)[1:-1]
    print("   running at t: " + "  ".join("%.0fus:%d" % (q, int(((s <= q) & (e > q)).sum())) for q in qs))
    late = np.argsort(-e)[:6]
    print("   last finishers (index, start, end): " + ", ".join("(%d, %.1f, %.1f)" % (int(np.nonzero(live)[0][i]), s[i], e[i]) for i in late))
    long_ = np.argsort(-d)[:6]
    print("   longest (index, start, duration): " + ", ".join("(%d, %.1f, %.1f)" % (int(np.nonzero(live)[0][i]), s[i], d[i]) for i in long_))
