"""Per-kernel HIP-event times (one stream, fused loss, rotating inputs) of a bench config, for the current build or an alternative
library (MM_DBG_LIB=path):   python profiles/tools/kernel_times.py config2 [config3 ...]"""
import sys, importlib, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("3d-magic-mirror_amd"); stepmod = importlib.import_module("3d-magic-mirror_amd.step")
if os.environ.get("MM_DBG_LIB"):
    pkg._native.LIB_PATH = os.environ["MM_DBG_LIB"]
    importlib.import_module("3d-magic-mirror_amd.build_native").needs_build = lambda: False
dev = torch.device("cuda:0")
for cfg in (sys.argv[1:] or ["config2"]):
    name, B, S, ratio = bench.CONFIGS[cfg] if cfg in bench.CONFIGS else (lambda t: (t[0], int(t[1]), int(t[2]), int(t[3])))(cfg.split(":"))   # or template:B:size:ratio
    dr = pkg.DiffRender(os.path.join(ROOT, "tests", "golden", "templates", name + ".npz"), S, ratio=ratio, emit_imnormal=os.environ.get("MM_IMN") == "1")
    dr.options = int(os.environ.get("MM_OPTIONS", "0"))          # MMRenderDesc.options (e.g. 2 / 4: force a walk-kernel shape)
    H, W = dr.render_height, dr.image_size
    batches = []
    for r in range(6):
        att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, B, H, W, seed=100 * r)
        if os.environ.get("MM_DIST"):                              # camera distances drawn from lo,hi instead of SURVEY 8(d)'s U(2,7): far cameras fold the mesh into a few heavy tiles
            lo_, hi_ = map(float, os.environ["MM_DIST"].split(",")); att["distances"] = torch.rand(B, generator=torch.Generator().manual_seed(r)) * (hi_ - lo_) + lo_
        batches.append(({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}, gt.to(dev)))
    st = stepmod.RenderLossStep(dr, batches[0][0], batches[0][1], fused=True, emit_imnormal=os.environ.get("MM_IMN") == "1")
    st.enable_profiling()
    acc = {}
    for i in range(33):
        st.set_inputs(*batches[i % 6]); st.run(); torch.cuda.synchronize()
        if i >= 3:
            for k, v in st.kernel_times_ms().items(): acc.setdefault(k, []).append(v * 1e3)
    st.disable_profiling()
    import time
    for _ in range(20): st.run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 300
    for i in range(n):
        st.set_inputs(*batches[i % 6]); st.run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    ks = {k: round(float(np.mean(v)), 1) for k, v in acc.items() if np.isfinite(np.mean(v))}
    print(cfg, ks, "sum %.1f us | one-stream step %.1f us = %.0f img/s" % (sum(ks.values()), dt * 1e6, B / dt))
