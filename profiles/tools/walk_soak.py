"""Soak of the two forward boundaries (fused render / kaolin-shaped dibr_rasterization through the shim): the same input rendered N times,
every result compared with the oracle's face_idx and with the first run.   python profiles/tools/walk_soak.py [N]"""
import sys, importlib, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("3d-magic-mirror_amd")
import oracle
SHIM = os.path.join(ROOT, "3d-magic-mirror_amd", "shim")
sys.path.insert(0, SHIM)
import kaolin as kal
import test_gpu_shim_ops as T
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
for name, B, S, no_mask, seed in [("smpl_uv_642", 3, 48, False, 2), ("sphere", 4, 64, True, 0), ("smpl_uv_642", 48, 128, True, 0)]:
    dr = pkg.DiffRender(os.path.join(ROOT, "tests", "golden", "templates", name + ".npz"), S)
    att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, B, S, S, seed=seed)
    A = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}
    Tn = oracle.camera(att["distances"].numpy(), att["elevations"].numpy(), att["azimuths"].numpy(), att["biases"].numpy())
    Tm = torch.from_numpy(Tn).to(dev)
    ref = None
    bad = {"fused": 0, "dibr": 0}
    first = {}
    for i in range(N):
        with torch.no_grad():
            r2, _ = dr.render(no_mask=no_mask, **A)
            f2 = dr.last_face_idx.clone()
            r1, fn1, f1 = T._reference_order_render(kal, dr, A, Tm, no_mask)
            f1 = f1.int()
        if ref is None:
            ref = f2.clone()
            print(name, B, S, "first run: fused == dibr:", bool(torch.equal(f1, f2)))
        for k, f in (("fused", f2), ("dibr", f1)):
            if not torch.equal(f, ref):
                bad[k] += 1
                if k not in first:
                    d = (f != ref).nonzero()
                    first[k] = (i, d[:8].tolist(), f[f != ref][:8].tolist(), ref[f != ref][:8].tolist(), int((f != ref).sum()))
    print(name, B, S, "runs", N, "mismatching runs", bad, first)
