"""How many sweep items (128-pixel chunks of face boxes) the backward plans per image, with the forward's face flags (compacting walk) and
without (per-batch walk: every face is swept over its inflated box):   python profiles/tools/sweep_items.py config5"""
import sys, importlib, os, ctypes, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("3d-magic-mirror_amd"); stepmod = importlib.import_module("3d-magic-mirror_amd.step"); N = pkg._native
dev = torch.device("cuda:0")
for cfg in (sys.argv[1:] or ["config5"]):
    name, B, S, ratio = bench.CONFIGS[cfg]
    for opt, label in ((1 << 10, "compacting walk + face flags"), (1 << 11, "per-batch walk, no flags")):
        dr = pkg.DiffRender(os.path.join(ROOT, "tests", "golden", "templates", name + ".npz"), S, ratio=ratio, emit_imnormal=False)
        dr.options = opt
        H, W = dr.render_height, dr.image_size
        att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, B, H, W, seed=0)
        st = stepmod.RenderLossStep(dr, {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}, gt.to(dev), fused=True)
        st.run(); torch.cuda.synchronize()
        out = (ctypes.c_size_t * 16)()
        assert N.lib().mm_debug_workspace_layout(ctypes.byref(st.d), out) == 0
        ni = st.ws[out[2]:out[2] + B * 8].view(torch.int32).reshape(B, 2).cpu().numpy()
        cm = st.ws[out[0]:out[0] + B * dr.num_faces * 8].view(torch.int32).reshape(B, dr.num_faces, 2).cpu().numpy()
        print("%s | %s: items per image mean %.0f (min %d max %d), chunk px %s; faces with items: %.1f %% ; item cap %d" % (
            cfg, label, ni[:, 0].mean(), ni[:, 0].min(), ni[:, 0].max(), sorted(set(ni[:, 1].tolist())), 100.0 * (cm[:, :, 1] > 0).mean(), out[4]))
