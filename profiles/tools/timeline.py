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
    out = (ctypes.c_ulonglong * (MAXB * 3))()
    assert getattr(L, "mm_debug_timeline_" + kn)(out) == 0
    raw = np.frombuffer(out, dtype=np.uint64).reshape(MAXB, 3)
    m = raw[:, :2].astype(np.float64)
    live = m[:, 1] > 0
    m = m[live]
    hw = raw[live, 2]
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
    # ---- placement (r06): which CU every recorded workgroup ran on (HW_REG_HW_ID + HW_REG_XCC_ID of its first wave)
    idx = np.nonzero(live)[0]
    xcc = ((hw >> np.uint64(32)) & np.uint64(0xF)).astype(int); h = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64)
    cu = ((h >> 8) & 0xF).astype(int); sh = ((h >> 12) & 1).astype(int); se = ((h >> 13) & 7).astype(int)
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    cus = np.unique(key)
    per = {k: np.nonzero(key == k)[0] for k in cus}
    cnt = np.array([len(per[k]) for k in cus]); busy = np.array([d[per[k]].sum() for k in cus]); last = np.array([e[per[k]].max() for k in cus])
    print("   placement: %d distinct CUs (xcc,se,sh,cu); XCDs used %s" % (len(cus), sorted(set(xcc.tolist()))))
    print("   workgroups per CU: min %d mean %.1f max %d | sum of their durations per CU: min %.0f mean %.0f max %.0f us | CU's last end: min %.1f mean %.1f max %.1f us" %
          (cnt.min(), cnt.mean(), cnt.max(), busy.min(), busy.mean(), busy.max(), last.min(), last.mean(), last.max()))
    print("   corr(CU's last end, CU's summed durations) = %.2f ; corr(last end, workgroups on the CU) = %.2f" % (np.corrcoef(last, busy)[0, 1], np.corrcoef(last, cnt)[0, 1]))
    conc0 = np.array([int(((s[per[k]] <= 3.0) & (e[per[k]] > 3.0)).sum()) for k in cus])
    print("   workgroups running on a CU at t = 3 us: min %d mean %.1f max %d (histogram %s)" % (conc0.min(), conc0.mean(), conc0.max(), np.bincount(conc0).tolist()))
    for k in cus[np.argsort(-last)[:4]]:
        ii = per[k][np.argsort(s[per[k]])]
        print("   slowest CU xcc %d se %d sh %d cu %2d: " % (k // 256, (k // 32) % 8, (k // 16) % 2, k % 16) + " ".join("[#%d %.1f+%.1f]" % (int(idx[i]), s[i], d[i]) for i in ii))
    for k in cus[np.argsort(last)[:2]]:
        ii = per[k][np.argsort(s[per[k]])]
        print("   fastest CU xcc %d se %d sh %d cu %2d: " % (k // 256, (k // 32) % 8, (k // 16) % 2, k % 16) + " ".join("[#%d %.1f+%.1f]" % (int(idx[i]), s[i], d[i]) for i in ii))
    first = np.argsort(idx)[:48]
    print("   first 48 recorded workgroups -> (xcc,se,sh,cu): " + " ".join("%d:%d.%d.%d.%d" % (int(idx[i]), xcc[i], se[i], sh[i], cu[i]) for i in first))
