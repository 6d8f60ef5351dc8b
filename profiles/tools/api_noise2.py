"""Why bench.py's first autograd flavour reads ~240 k images/s where profiles/tools/api_noise.py reads 400 k: the same three flavours, timed like
bench.py times them, under the conditions bench.py adds one at a time (MODE env: plain | queues | streams | longwarm | all)."""
import sys, os
MODE = os.environ.get("MODE", "plain")
if MODE in ("queues", "all"):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
import importlib, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
pkg = importlib.import_module("3d-magic-mirror_amd"); stepmod = importlib.import_module("3d-magic-mirror_amd.step")
dev = torch.device("cuda:0")
LEAVES = ("vertices", "textures", "lights", "bg", "azimuths", "elevations", "distances", "biases")
dr = pkg.DiffRender(os.path.join(ROOT, "tests/golden/templates/smpl_uv_642.npz"), 128)
sets = []
for r in range(8):
    att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, 48, 128, 128, seed=r)
    datt = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}
    sets.append((datt, {k: datt[k].clone().requires_grad_(True) for k in LEAVES}, gt.to(dev)))
if MODE in ("streams", "all"):                                   # what bench.py does before its API section: C-ABI steps on four streams
    streams = [torch.cuda.Stream(dev) for _ in range(4)]
    steps = [stepmod.RenderLossStep(dr, sets[i][0], sets[i][2], no_mask=True, fused=True, emit_imnormal=True) for i in range(4)]
    for it in range(2000):
        steps[it % 4].run(streams[it % 4])
    torch.cuda.synchronize()
ctr = [0]
def one(fused=False):
    datt, lv, gtd = sets[ctr[0] % 8]; ctr[0] += 1
    for v in lv.values(): v.grad = None
    a = dict(datt); a.update(lv)
    if fused: dr.render_recon(gtd, no_mask=True, **a)[0].backward()
    else:
        rgbs, _ = dr.render(no_mask=True, **a)
        dr.recon_data(rgbs, gtd, no_mask=True).backward()
def host(fn, n=60):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = (time.perf_counter() - t0) / n * 1e6; torch.cuda.synchronize(); return dt
def thr(fn, n=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 48 * n / (time.perf_counter() - t0)
flav = (("deferred", True, False), ("undeferred", False, False), ("fused", True, True))
if MODE in ("longwarm", "all"):                                  # bench.py's run-in: 300 steps of each flavour, no synchronisation in between
    for name, defer, fused in flav:
        dr.defer_recon_fusion = defer
        for _ in range(300): one(fused)
    torch.cuda.synchronize()
for rep in range(2):
    for name, defer, fused in flav:
        dr.defer_recon_fusion = defer
        f = lambda: one(fused)
        for _ in range(10): f()
        t = [thr(f) for _ in range(3)]
        print(MODE, rep, name, "img/s %.0f %.0f %.0f" % tuple(t), "host us %.1f" % host(f), flush=True)
