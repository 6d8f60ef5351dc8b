"""Per-kernel HIP-event times of one config-2 step for an alternative build of the library (MM_DBG_LIB=path)."""
import sys, importlib, os, torch, numpy as np
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("3d-magic-mirror_amd"); stepmod = importlib.import_module("3d-magic-mirror_amd.step")
if os.environ.get("MM_DBG_LIB"):
    pkg._native.LIB_PATH = os.environ["MM_DBG_LIB"]
    importlib.import_module("3d-magic-mirror_amd.build_native").needs_build = lambda: False
dev = torch.device("cuda:0")
dr = pkg.DiffRender("/root/repo/tests/golden/templates/smpl_uv_642.npz", 128, emit_imnormal=False)
att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, 48, 128, 128)
datt = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}; gtd = gt.to(dev)
st = stepmod.RenderLossStep(dr, datt, gtd, fused=True)
st.enable_profiling()
acc = {}
for i in range(23):
    st.run(); torch.cuda.synchronize()
    if i >= 3:
        for k, v in st.kernel_times_ms().items(): acc.setdefault(k, []).append(v * 1e3)
print({k: round(float(np.mean(v)), 2) for k, v in acc.items() if np.isfinite(np.mean(v))})
