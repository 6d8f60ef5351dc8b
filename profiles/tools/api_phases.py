"""Host time of the autograd-API step by phase (render / recon_data / backward), C++ host path vs Python path (MM_NO_TORCH_EXT=1)."""
import sys, importlib, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3d-magic-mirror_amd")
dev = torch.device("cuda:0")
LEAVES = ("vertices", "textures", "lights", "bg", "azimuths", "elevations", "distances", "biases")
for imn in (True, False):
    dr = pkg.DiffRender(os.path.join(ROOT, "tests/golden/templates/smpl_uv_642.npz"), 128, emit_imnormal=imn)
    sets = []
    for r in range(8):
        att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, 48, 128, 128, seed=r)
        datt = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}
        sets.append((datt, {k: datt[k].clone().requires_grad_(True) for k in LEAVES}, gt.to(dev)))
    acc = [0.0, 0.0, 0.0, 0.0]
    def one(i, rec):
        datt, lv, gtd = sets[i % 8]
        t0 = time.perf_counter()
        for v in lv.values(): v.grad = None
        a = dict(datt); a.update(lv)
        t1 = time.perf_counter()
        rgbs, _ = dr.render(no_mask=True, **a)
        t2 = time.perf_counter()
        loss = dr.recon_data(rgbs, gtd, no_mask=True)
        t3 = time.perf_counter()
        loss.backward()
        t4 = time.perf_counter()
        if rec:
            for k, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)): acc[k] += d
    for i in range(30): one(i, False)
    torch.cuda.synchronize()
    n = 300
    for i in range(n):
        one(i, True)
        if i % 10 == 9: torch.cuda.synchronize()          # keep the queue short: pure enqueue cost, never a full launch queue
    print("emit_imnormal=%s ext=%s: prep %.1f  render %.1f  recon_data %.1f  backward %.1f us per step" % (
        imn, pkg._native.torch_ext() is not None, *[1e6 * x / n for x in acc]))
