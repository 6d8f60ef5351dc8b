"""Per-wave wall-clock phase totals of the instrumented kernels (library variant built with -DMM_PHASE_PROF as lib/libmm_pp.so; the
product library is untouched):   python profiles/tools/phase_prof.py [config2|config3|config5]
Every mark waits for the wave's outstanding memory operations, so a phase is charged with the latency of its own loads (and the
kernel as a whole runs slower than the product build: read the SHARES, not the absolute times)."""
import sys, importlib, os, ctypes, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "config2"
name, B, S, ratio = bench.CONFIGS[cfg]
bn = importlib.import_module("3d-magic-mirror_amd.build_native")
var = os.environ.get("MM_PP_LIB") or os.path.join(os.path.dirname(bn.LIB), "libmm_pp.so")   # MM_PP_LIB: a prebuilt -DMM_PHASE_PROF variant (e.g. + -DMM_WALK_TWICE=1)
if os.environ.get("MM_PP_LIB"):
    assert os.path.exists(var), var
elif not os.path.exists(var) or any(os.path.getmtime(os.path.join(bn.CSRC, f)) > os.path.getmtime(var) for f in os.listdir(bn.CSRC)):
    bn.build(out=var, extra_flags=["-DMM_PHASE_PROF"])          # (prebuilt in the build container when possible: hipcc minutes are GPU-box minutes)
pkg = importlib.import_module("3d-magic-mirror_amd"); stepmod = importlib.import_module("3d-magic-mirror_amd.step")
pkg._native.LIB_PATH = var
bn.needs_build = lambda: False
dev = torch.device("cuda:0")
dr = pkg.DiffRender(os.path.join(ROOT, "tests", "golden", "templates", name + ".npz"), S, ratio=ratio, emit_imnormal=False)
dr.options = int(os.environ.get("MM_OPTIONS", "0"))
H, W = dr.render_height, dr.image_size
att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, B, H, W)
datt = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}
st = stepmod.RenderLossStep(dr, datt, gt.to(dev), fused=True)
for _ in range(4): st.run()
torch.cuda.synchronize()
L = ctypes.CDLL(var)
KERNELS = {
    "vertex_fwd": (["clear counters", "camera (fp64 trig + look-at)", "face records", "binning"], ("-", "-")),
    "vertex_bwd": (["T + vertex loads", "corner gather", "reductions + partial store", "ticket", "last WG: lights", "last WG: camera chain"], ("-", "-")),
    "gather_face": (["setup", "sweep: face_idx loads", "compaction", "item loads", "item math + LDS adds", "stores"], ("trips", "items")),
    "gather_tex": (["count + first record", "clear LDS", "records", "tile store"], ("records", "-")),
    "raster_fwd": (["tile setup", "mask -> id list", "geo fetch + stage + box tests + transposes", "colour pairs (coop: pair list)", "silhouette pairs (coop: pair evaluation)", "winner + shade + store", "coop: barrier + counts",
                   "coop: barrier after the pairs", "first mask-row load: latency alone", "first record fetch of a window: latency alone"], ("candidates", "batches (coop: pairs)")),
}
SL, MAXW = 10, 16384
print("== %s: %s B=%d %dx%d (100 MHz ticks -> us)" % (cfg, name, B, H, W))
for kn, (phases, cn) in KERNELS.items():
    fn = getattr(L, "mm_debug_pp_" + kn, None)
    if fn is None:
        continue
    out = (ctypes.c_ulonglong * (MAXW * (SL + 3)))()
    assert fn(out) == 0
    m = np.frombuffer(out, dtype=np.uint64).reshape(MAXW, SL + 3).astype(np.float64)
    m = m[m[:, SL] > 0]
    if not len(m):
        continue
    if os.environ.get("MM_PP_DUMP"):                             # raw per-wave rows for offline analysis
        np.save(os.path.join(ROOT, "gpurun_out", "pp_%s_%s.npy" % (cfg, kn)), m)
    tot = m[:, SL] / 100.0
    print("%s: %d waves recorded; wave time mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f us" % (kn, len(m), tot.mean(), np.median(tot), np.percentile(tot, 90), np.percentile(tot, 99), tot.max()))
    for i, ph in enumerate(phases):
        v = m[:, i] / 100.0
        print("   %-46s mean %6.2f  p90 %6.2f  max %6.1f us   %5.1f %% of all wave time" % (ph, v.mean(), np.percentile(v, 90), v.max(), 100 * v.sum() / tot.sum()))
    acc = m[:, :len(phases)].sum(1) / 100.0
    print("   %-46s mean %6.2f us" % ("(unaccounted)", (tot - acc).mean()))
    print("   %s: mean %.2f max %d;  %s: mean %.1f max %d" % (cn[0], m[:, SL + 1].mean(), m[:, SL + 1].max(), cn[1], m[:, SL + 2].mean(), m[:, SL + 2].max()))
    if kn == "raster_fwd":                                       # single-wave tiles: c0 = candidates | flushes << 32, c1 = colour pairs | silhouette pairs << 32
        c0, c1 = m[:, SL + 1].astype(np.uint64), m[:, SL + 2].astype(np.uint64)
        cand, fl, hp, sp = (c0 & 0xFFFFFFFF).astype(float), (c0 >> 32).astype(float), (c1 & 0xFFFFFFFF).astype(float), (c1 >> 32).astype(float)
        print("   per wave: candidates %.1f  flushes %.2f  colour pairs %.1f  silhouette pairs %.1f" % (cand.mean(), fl.mean(), hp.mean(), sp.mean()))
        for lo_, hi_ in ((0, 64), (64, 192), (192, 512), (512, 1024), (1024, 1e9)):
            sel = (cand >= lo_) & (cand < hi_)
            if sel.any():
                print("   tiles with %4d <= candidates < %-6g: %5d waves, %4.1f %% of wave time; mean %.1f us, flushes %.1f, colour pairs %.0f, silhouette pairs %.0f"
                      % (lo_, hi_, sel.sum(), 100 * tot[sel].sum() / tot.sum(), tot[sel].mean(), fl[sel].mean(), hp[sel].mean(), sp[sel].mean()))
        for h in np.argsort(-tot)[:8]:
            print("   slow wave %.1f us: candidates %d flushes %d colour pairs %d silhouette pairs %d" % (tot[h], cand[h], fl[h], hp[h], sp[h]))
    heavy = np.argsort(-tot)[:int(os.environ.get('MM_PP_SLOWEST', '4'))]
    for h in heavy:
        print("   slowest wave: total %.1f us | " % tot[h] + "  ".join("%s %.1f" % (ph.split(":")[0].split(" ")[0], m[h, i] / 100.0) for i, ph in enumerate(phases)) + " | %s %d %s %d" % (cn[0], m[h, SL + 1], cn[1], m[h, SL + 2]))
