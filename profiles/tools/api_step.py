"""The DiffRender autograd path (render -> recon_data -> backward) at config 2: host time per step (enqueue only), wall time per step
(enqueue + GPU, asynchronous), and GPU time per step (events around a step with the host far ahead)."""
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
    def one(i):
        datt, lv, gtd = sets[i % 8]
        for v in lv.values(): v.grad = None
        a = dict(datt); a.update(lv)
        rgbs, _ = dr.render(no_mask=True, **a)
        dr.recon_data(rgbs, gtd, no_mask=True).backward()
    for i in range(30): one(i)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for i in range(n): one(i)
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_wall = (time.perf_counter() - t0) / n
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(2e8))                                   # let the host run ahead: the events then bracket GPU time only
    e0.record()
    for i in range(50): one(i)
    e1.record(); torch.cuda.synchronize()
    print("emit_imnormal=%s: host %.1f us/step, wall %.1f us/step (%.0f img/s), GPU %.1f us/step" % (imn, t_host * 1e6, t_wall * 1e6, 48 / t_wall, e0.elapsed_time(e1) * 1e3 / 50))
