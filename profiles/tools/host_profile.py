"""cProfile of the autograd-API step (DiffRender.render + recon_data + backward) to see where host time goes."""
import sys, importlib, cProfile, pstats, torch
sys.path.insert(0, '/root/repo')
pkg = importlib.import_module("3d-magic-mirror_amd")
dev = torch.device("cuda:0")
import os
dr = pkg.DiffRender("/root/repo/tests/golden/templates/smpl_uv_642.npz", 128, emit_imnormal=bool(int(os.environ.get("MM_IMNORMAL", "0"))))
att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, 48, 128, 128)
datt = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}; gtd = gt.to(dev)
LEAVES = ("vertices", "textures", "lights", "bg", "azimuths", "elevations", "distances", "biases")
leaves = {k: datt[k].clone().requires_grad_(True) for k in LEAVES}
def one():
    for v in leaves.values(): v.grad = None
    a = dict(datt); a.update(leaves)
    rgbs, _ = dr.render(no_mask=True, **a)
    dr.recon_data(rgbs, gtd, no_mask=True).backward()
for _ in range(20): one()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): one()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
