"""Where the HOST time of the un-fused, kaolin-shaped operator chain (shim_chain.py = the reference's render through the shim) goes:
cProfile over N steps of render -> recon_data -> backward, top functions by own time.   python profiles/tools/shim_cprofile.py [steps]"""
import sys, importlib, os, time, cProfile, pstats, io, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3d-magic-mirror_amd"); chain = importlib.import_module("3d-magic-mirror_amd.shim_chain")
dev = torch.device("cuda:0")
LEAVES = ("vertices", "textures", "lights", "bg", "azimuths", "elevations", "distances", "biases")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dr = pkg.DiffRender(os.path.join(ROOT, "tests/golden/templates/smpl_uv_642.npz"), 128, emit_imnormal=False)
att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, 48, 128, 128, seed=0)
datt = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}
for k in LEAVES: datt[k] = datt[k].clone().requires_grad_(True)
gtd = gt.to(dev)
def one():
    for k in LEAVES: datt[k].grad = None
    rgbs, fn, fidx = chain.render(dr, no_mask=True, **datt)
    chain.recon_data(dr, rgbs, gtd).backward()
for i in range(20): one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(N): one()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("shim chain: host %.1f us/step, wall %.1f us/step" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
pr = cProfile.Profile(); pr.enable()
for i in range(N): one()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30); print(s.getvalue()[:8000])
