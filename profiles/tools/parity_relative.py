"""Per-input RELATIVE gradient error of the fused HIP backward against the CPU oracle (fp32 and fp64 instantiations): max|got - ref| / max|ref|
(no floor of 1) and the relative L2 error, for two upstream gradients -- the loss of recon_data (tiny gradients: a batch mean) and an O(1) random
upstream gradient (rgbs * w).sum() + (face_normals * wfn).sum(), w ~ N(0,1) -- on BASELINE configs 1, 2 (full size) and one image of config 5.

    python profiles/tools/parity_relative.py [out.md]
"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
pkg = importlib.import_module("3d-magic-mirror_amd")
LEAVES = ("vertices", "textures", "lights", "bg", "azimuths", "elevations", "distances", "biases")
TEMPLATES = os.path.join(ROOT, "tests", "golden", "templates")
dev = torch.device("cuda:0")

CASES = [("config 1", "sphere", 4, 64, 0), ("config 2", "smpl_uv_642", 48, 128, 0), ("config 5 (one image)", "smpl_uv", 1, 512, 0)]


def rel(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    m = float(np.abs(b).max())
    return float(np.abs(a - b).max()) / m, float(np.sqrt(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-300)))


rows = []
for label, name, B, S, seed in CASES:
    for upstream in ("recon_data", "unit-normal w"):
        dr = pkg.DiffRender(os.path.join(TEMPLATES, name + ".npz"), S, emit_imnormal=True)
        H, W = dr.render_height, dr.image_size
        att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, B, H, W, seed=seed)
        datt = {k: (v.to(dev).requires_grad_(k in LEAVES) if torch.is_tensor(v) else v) for k, v in att.items()}
        inp = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in att.items()}
        inp["faces"] = dr.faces.numpy().astype(np.int32); inp["face_uvs"] = dr.face_uvs.numpy()[0]
        proj = dr.cam_proj.numpy().reshape(3)
        rng = np.random.default_rng(seed + 77)
        rgbs, out = dr.render(no_mask=True, **datt)
        rgba_o, fidx_o, fn_o, imn_o = oracle.render_forward(inp, H, W, True, proj)
        if upstream == "recon_data":
            dr.recon_data(rgbs, gt.to(dev), no_mask=True).backward()
            _, dpred = oracle.recon_data(rgba_o.transpose(0, 3, 1, 2), gt.numpy(), image_weight=dr.image_weight, want_grad=True)
            dpred = np.ascontiguousarray(dpred.transpose(0, 2, 3, 1)); wfn = None
        else:
            dpred = rng.normal(size=(B, H, W, 4)).astype(np.float32)
            wfn = rng.normal(size=(B, dr.num_faces, 3)).astype(np.float32)
            ((rgbs.permute(0, 2, 3, 1) * torch.from_numpy(dpred).to(dev)).sum() + (out["face_normals"] * torch.from_numpy(wfn).to(dev)).sum()).backward()
        torch.cuda.synchronize()
        nf = int((dr.last_face_idx.cpu().numpy() != fidx_o).sum())
        g32 = oracle.render_backward(inp, H, W, True, proj, dpred, wfn)
        g64 = oracle.render_backward(inp, H, W, True, proj, dpred.astype(np.float64), None if wfn is None else wfn.astype(np.float64), dtype=np.float64)
        for k in LEAVES:
            got = datt[k].grad.cpu().numpy()
            h32, l32 = rel(got, g32[k]); h64, l64 = rel(got, g64[k]); o64, lo64 = rel(g32[k], g64[k])
            rows.append((label, upstream, k, float(np.abs(g32[k]).max()), h32, l32, h64, l64, o64, lo64, nf))
            print("%-22s %-14s %-10s max|ref| %.2e  hip-o32 %.2e (L2 %.2e)  hip-o64 %.2e (L2 %.2e)  o32-o64 %.2e (L2 %.2e)  face_idx diff %d" % rows[-1], flush=True)

if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        f.write("# Relative gradient error of the fused HIP backward against the oracle (round 5)\n\n")
        f.write("`python profiles/tools/parity_relative.py`.  err = max|got - ref| / max|ref| (no floor), L2 = ||got - ref|| / ||ref||.  o32 / o64 = the oracle's fp32 / fp64 "
                "instantiation on the same inputs and upstream gradient.  Upstream `recon_data` = the loss of `networks.py:374-377` (a batch mean: tiny gradients); `unit-normal w` = "
                "`(rgbs * w).sum() + (face_normals * wfn).sum()`, w, wfn ~ N(0,1).\n\n")
        f.write("| config | upstream | input | max\\|ref\\| | HIP vs o32: err | L2 | HIP vs o64: err | L2 | o32 vs o64: err | L2 | face_idx diffs |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            f.write("| %s | %s | %s | %.2e | %.2e | %.2e | %.2e | %.2e | %.2e | %.2e | %d |\n" % r)
