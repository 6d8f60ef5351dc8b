import sys, importlib, os, torch, time
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("3d-magic-mirror_amd"); stepmod = importlib.import_module("3d-magic-mirror_amd.step")
dev = torch.device("cuda:0")
if os.environ.get("MM_DBG_LIB"):
    pkg._native.LIB_PATH = os.environ["MM_DBG_LIB"]
    importlib.import_module("3d-magic-mirror_amd.build_native").needs_build = lambda: False

name, B, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
fused = sys.argv[4] == "fused"
dr = pkg.DiffRender("/root/repo/tests/golden/templates/%s.npz" % name, S, emit_imnormal=False)
dr.options = int(sys.argv[5]) if len(sys.argv) > 5 else 0
att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, B, dr.render_height, dr.image_size)
datt = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}; gtd = gt.to(dev)
st = stepmod.RenderLossStep(dr, datt, gtd, fused=fused)
print("forward...", flush=True)
import ctypes
N = pkg._native
s = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
N.check(N.lib().mm_render_forward(ctypes.byref(st.d), s), "fwd")
torch.cuda.synchronize(); print("forward ok", flush=True)
st.run(); torch.cuda.synchronize(); print("step ok", float(st.loss), flush=True)
