#!/usr/bin/env python3
"""Mints the golden fixtures under tests/golden/ by IMPORTING the reference in this container.

Run from the repo root:  python tests/golden/make_golden.py
Needs /root/reference (absent on the GPU box; the fixtures are committed, the reference never ships).

What is pinned:
  camera.npz        smr_utils.camera_position_from_spherical_angles / generate_transformation_matrix
                    (/root/reference/smr_utils.py:257-311), B=16 seeded draws.
  template_*.npz    the reference's own DiffRender.__init__ (/root/reference/networks.py:165-256) executed against this repo's
                    kaolin-shaped import boundary (3d-magic-mirror_amd/shim; real kaolin is not importable):
                    pins vertices_init, flip_index, edges, edge2faces(as unordered pairs), cam_proj, face_uvs,
                    the Laplacian.
  losses.npz        DiffRender.recon_data / recon_att / recon_flip / calc_reg_* (/root/reference/networks.py:326-491)
                    on seeded inputs (values + autograd gradients); mask_iou is supplied by this repo's torch
                    restatement, chamfer=False.
  templates/*.npz   the template meshes (data files of the reference: vertices / faces / uvs / face_uvs_idx arrays parsed
                    from template/*.obj) so that tests and bench.py can run where /root/reference does not exist.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.environ.get("MM_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
mm = importlib.import_module("3d-magic-mirror_amd")
from importlib import import_module  # noqa: E402

obj_io = import_module("3d-magic-mirror_amd.obj_io")
template = import_module("3d-magic-mirror_amd.template")


def mask_iou(lhs_mask, rhs_mask):
    """torch restatement of kaolin.metrics.render.mask_iou."""
    b = lhs_mask.shape[0]
    mul = lhs_mask * rhs_mask
    up = mul.reshape(b, -1).sum(1)
    down = ((lhs_mask + rhs_mask) - mul).reshape(b, -1).sum(1)
    return 1.0 - torch.mean(up / (down + 1e-10))


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__path__ = []
        sys.modules[name] = m
        return m

    class _Any:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return _Any()

        def __getattr__(self, n):
            return _Any()

    # kaolin / pytorch3d: THIS repo's import boundary (3d-magic-mirror_amd/shim: kaolin's and pytorch3d's own module paths over the
    # MI355X kernels) instead of ad-hoc stand-ins -- the reference's modules bind the very names a deployment would give them.
    sys.path.insert(0, os.path.join(ROOT, "3d-magic-mirror_amd", "shim"))
    import kaolin
    import pytorch3d  # noqa: F401
    # The device operators need an MI355X; this script runs in the (GPU-less) build container and only ever calls, of those, mask_iou
    # (inside recon_data): for MINTING the recon_data fixture it is the torch restatement above.  Everything else the fixtures
    # exercise (import_mesh, index_vertices_by_faces, uniform_laplacian, generate_perspective_projection) is the shim's own code.
    kaolin.metrics.render.mask_iou = mask_iou
    tv = mod("torchvision")
    tv.models = mod("torchvision.models")
    tv.transforms = mod("torchvision.transforms")
    mod("timm")
    mod("fid_score", calculate_fid_given_paths=None)
    mod("inception", InceptionV3=_Any)


def main():
    os.makedirs(os.path.join(OUT, "templates"), exist_ok=True)
    sys.path.insert(0, REF)
    install_stubs()
    import smr_utils  # noqa: E402  (the reference's own module)

    # ---- camera --------------------------------------------------------------------------------------------
    g = torch.Generator().manual_seed(1234)
    B = 16
    dist = torch.rand(B, generator=g) * 5 + 2
    elev = torch.rand(B, generator=g) * 60 - 30
    azim = torch.rand(B, generator=g) * 360 - 180
    bias = torch.rand(B, 2, generator=g) - 0.5
    cam_pos = smr_utils.camera_position_from_spherical_angles(dist, elev, azim, degrees=True)
    at = torch.cat([bias, torch.zeros(B, 1)], 1)
    up = torch.tensor([[0., 1., 0.]]).repeat(B, 1)
    T = smr_utils.generate_transformation_matrix(cam_pos, at, up)
    np.savez(os.path.join(OUT, "camera.npz"), dist=dist.numpy(), elev=elev.numpy(), azim=azim.numpy(), bias=bias.numpy(),
             camera_position=cam_pos.numpy(), transform=T.numpy())

    # ---- templates (data) + the reference's own __init__ ------------------------------------------------------
    import networks  # noqa: E402  (the reference's own module, kaolin entry points stubbed above)
    torch.Tensor.cuda = lambda self, *a, **k: self   # networks.py:252 hard-codes .cuda()
    for name, ell, ratio in (("sphere", 1, 1), ("smpl_uv_642", 1, 1), ("ellipsoid", 1, 1), ("smpl_uv_642", 2, 2), ("sphere", -1, 1)):
        path = os.path.join(REF, "template", name + ".obj")
        dr = networks.DiffRender(path, 64, ratio=ratio, init_ellipsoid=ell)
        e2f = torch.sort(dr.edge2faces, dim=1)[0]
        np.savez_compressed(os.path.join(OUT, "template_%s_e%d_r%d.npz" % (name, ell, ratio)),
                            vertices_init=dr.vertices_init.numpy(), flip_index=dr.flip_index.numpy(), edges=dr.edges.numpy(),
                            edge2faces_sorted=e2f.numpy(), cam_proj=dr.cam_proj.numpy(), face_uvs=dr.face_uvs.numpy(),
                            sign_init=dr.sign_init.numpy(),
                            laplacian_rowsum=dr.vertices_laplacian_matrix.sum(1).numpy(),
                            laplacian_nnz=np.asarray((dr.vertices_laplacian_matrix != 0).sum().item()))
    for name in ("sphere", "ellipsoid", "smpl_uv_642", "smpl_uv", "sphere2"):
        m = obj_io.import_mesh(os.path.join(REF, "template", name + ".obj"))
        np.savez_compressed(os.path.join(OUT, "templates", name + ".npz"), vertices=m.vertices.numpy(),
                            faces=m.faces.numpy().astype(np.int32), uvs=m.uvs.numpy(),
                            face_uvs_idx=m.face_uvs_idx.numpy().astype(np.int32))

    # ---- losses / regularisers -----------------------------------------------------------------------------------
    dr = networks.DiffRender(os.path.join(REF, "template", "sphere.obj"), 64, ratio=1, init_ellipsoid=1,
                             image_weight=0.1, lambda_lpl=0.1, lambda_flat=0.001)
    g = torch.Generator().manual_seed(4321)
    B, V, F, H, W = 4, dr.num_vertices, dr.num_faces, 32, 24
    out = {}

    def att(seed_shift):
        gg = torch.Generator().manual_seed(4321 + seed_shift)
        dv = (0.1 * torch.randn(B, V, 3, generator=gg)).requires_grad_(True)
        fnrm = torch.nn.functional.normalize(torch.randn(B, F, 3, generator=gg), dim=2).requires_grad_(True)
        return {"delta_vertices": dv, "vertices": (dr.vertices_init[None] + dv), "face_normals": fnrm,
                "azimuths": (torch.rand(B, generator=gg) * 360 - 180).requires_grad_(True),
                "elevations": (torch.rand(B, generator=gg) * 30).requires_grad_(True),
                "distances": (torch.rand(B, generator=gg) * 5 + 2).requires_grad_(True),
                "biases": (torch.rand(B, 2, generator=gg) - 0.5).requires_grad_(True),
                "textures": torch.rand(B, 3, 16, 8, generator=gg).requires_grad_(True),
                "lights": torch.rand(B, 9, generator=gg).requires_grad_(True)}

    A, A2 = att(0), att(1)
    for k in ("delta_vertices", "face_normals", "azimuths", "elevations", "distances", "biases", "textures", "lights"):
        out["A_" + k] = A[k].detach().numpy(); out["A2_" + k] = A2[k].detach().numpy()

    def record(name, value, wrt):
        out[name] = np.asarray(value.detach().numpy())
        grads = torch.autograd.grad(value, [w for w in wrt.values()], allow_unused=True, retain_graph=True)
        for (k, w), gr in zip(wrt.items(), grads):
            out[name + "__d_" + k] = (torch.zeros_like(w) if gr is None else gr).numpy()

    record("calc_reg_loss", dr.calc_reg_loss(A), {"delta_vertices": A["delta_vertices"], "face_normals": A["face_normals"]})
    record("calc_reg_edge", dr.calc_reg_edge(A["vertices"]), {"delta_vertices": A["delta_vertices"]})
    record("calc_reg_depth", dr.calc_reg_depth(A["vertices"]), {"delta_vertices": A["delta_vertices"]})
    record("calc_reg_depthR", dr.calc_reg_depthR(A["vertices"], temp=2), {"delta_vertices": A["delta_vertices"]})
    record("calc_reg_depthC", dr.calc_reg_depthC(A["vertices"]), {"delta_vertices": A["delta_vertices"]})
    record("calc_reg_deform", dr.calc_reg_deform(A["delta_vertices"]), {"delta_vertices": A["delta_vertices"]})
    # recon_flip(L1=True) raises in the reference (networks.py:409 broadcasts (B,V,3) against (B,V)); pin that too
    try:
        dr.recon_flip(A, True)
        out["recon_flip_L1_raises"] = np.asarray(0)
    except RuntimeError:
        out["recon_flip_L1_raises"] = np.asarray(1)
    record("recon_flip_L10", dr.recon_flip(A, False), {"delta_vertices": A["delta_vertices"]})
    for L1 in (True, False):
        parts = dr.recon_att(A, A2, L1=L1, chamfer=False, azim=1)
        wrt = {k: A[k] for k in ("azimuths", "elevations", "distances", "biases", "delta_vertices", "textures", "lights")}
        for nm, val in zip(("cam", "shape", "texture", "light", "bias"), parts):
            record("recon_att_L1%d_%s" % (L1, nm), val, wrt)

    pred = torch.rand(B, H, W, 4, generator=g).permute(0, 3, 1, 2).requires_grad_(True)   # NCHW view of NHWC, like render
    gt = torch.rand(B, 4, H, W, generator=g)
    gt[:, 3] = (gt[:, 3] > 0.5).float()
    out["rd_pred_nhwc"] = pred.detach().permute(0, 2, 3, 1).contiguous().numpy(); out["rd_gt"] = gt.numpy()
    import io, contextlib
    for contour in (0.0, 0.5):
        with contextlib.redirect_stdout(io.StringIO()):     # networks.py:387 prints
            val = dr.recon_data(pred, gt, no_mask=True, contour=contour)
        out["recon_data_c%g" % contour] = val.detach().numpy()
        out["recon_data_c%g__d_pred_nhwc" % contour] = torch.autograd.grad(val, pred)[0].permute(0, 2, 3, 1).contiguous().numpy()
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **out)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
