"""Mesh regularisers: the HIP kernels behind DiffRender.calc_reg_* / recon_flip / regularization against
(a) outputs and gradients of the reference itself (tests/golden/losses.npz) and (b) oracle/reg_oracle.py on other templates,
batch sizes and aspect ratios.  Values and gradients are held to 2e-5 / 2e-4 relative (fp32 sums in a different order)."""
import os
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN, TEMPLATES

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _leaf(t):
    return t.clone().to(DEV).requires_grad_(True)


def _cmp(got, ref, rtol, atol):
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().cpu().numpy() if torch.is_tensor(ref) else ref, rtol=rtol, atol=atol)


def test_mesh_regularisers_match_reference_golden(pkg):
    z = np.load(os.path.join(GOLDEN, "losses.npz"))
    dr = pkg.DiffRender(os.path.join(TEMPLATES, "sphere.npz"), 64, image_weight=0.1, lambda_lpl=0.1, lambda_flat=0.001)
    dv = _leaf(torch.from_numpy(z["A_delta_vertices"]))
    fn = _leaf(torch.from_numpy(z["A_face_normals"]))
    A = {"delta_vertices": dv, "face_normals": fn, "vertices": dr.vertices_init[None].to(DEV) + dv}

    def check(name, value, wrt):
        np.testing.assert_allclose(float(value), float(z[name]), rtol=2e-5, atol=1e-7)
        grads = torch.autograd.grad(value, [A[k] for k in wrt], allow_unused=True, retain_graph=True)
        for k, g in zip(wrt, grads):
            ref = z[name + "__d_" + k]
            got = np.zeros_like(ref) if g is None else g.cpu().numpy()
            np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-7, err_msg=name + " d/d" + k)

    check("calc_reg_loss", dr.calc_reg_loss(A), ("delta_vertices", "face_normals"))
    check("calc_reg_edge", dr.calc_reg_edge(A["vertices"]), ("delta_vertices",))
    check("calc_reg_depth", dr.calc_reg_depth(A["vertices"]), ("delta_vertices",))
    check("calc_reg_depthR", dr.calc_reg_depthR(A["vertices"], temp=2), ("delta_vertices",))
    check("calc_reg_depthC", dr.calc_reg_depthC(A["vertices"]), ("delta_vertices",))
    check("calc_reg_deform", dr.calc_reg_deform(A["delta_vertices"]), ("delta_vertices",))
    check("recon_flip_L10", dr.recon_flip(A, False), ("delta_vertices",))
    with pytest.raises(RuntimeError):
        dr.recon_flip(A, True)


@pytest.mark.parametrize("name,B,ratio,seed", [("smpl_uv_642", 5, 2, 0), ("ellipsoid", 48, 1, 1), ("smpl_uv", 3, 1, 2), ("sphere2", 2, 2, 3)])
def test_mesh_regularisers_match_oracle(pkg, name, B, ratio, seed):
    import reg_oracle as R
    dr = pkg.DiffRender(os.path.join(TEMPLATES, name + ".npz"), 32, ratio=ratio, lambda_lpl=0.3, lambda_flat=0.02)
    g = torch.Generator().manual_seed(seed)
    V, F = dr.num_vertices, dr.num_faces
    dv0 = 0.1 * torch.randn(B, V, 3, generator=g)
    dv0[0, :7] = 0.0                                         # zero displacement: |dv| has a kink, both sides must give 0 gradient
    dv0[1, :, 2] = dv0[1, :, 2].abs() * torch.sign(dr.vertices_init[:, 2])       # an image where no vertex crossed the mirror plane
    fn0 = torch.nn.functional.normalize(torch.randn(B, F, 3, generator=g), dim=2)
    host = types.SimpleNamespace(flip_index=dr.flip_index, sign_init=dr.sign_init.cpu(), edges=dr.edges, edge2faces=dr.edge2faces,
                                 vertices_laplacian_matrix=dr.vertices_laplacian_matrix, ratio=dr.ratio, lambda_lpl=dr.lambda_lpl,
                                 lambda_flat=dr.lambda_flat)
    cases = [
        ("reg_loss", lambda t, A: t.calc_reg_loss(A), lambda A: R.calc_reg_loss(host, A)),
        ("edge", lambda t, A: t.calc_reg_edge(A["vertices"]), lambda A: R.calc_reg_edge(host, A["vertices"])),
        ("depth", lambda t, A: t.calc_reg_depth(A["vertices"]), lambda A: R.calc_reg_depth(host, A["vertices"])),
        ("depthR", lambda t, A: t.calc_reg_depthR(A["vertices"], temp=1.5, eps=0.01), lambda A: R.calc_reg_depthR(host, A["vertices"], temp=1.5, eps=0.01)),
        ("depthC", lambda t, A: t.calc_reg_depthC(A["vertices"], eps=0.02), lambda A: R.calc_reg_depthC(host, A["vertices"], eps=0.02)),
        ("deform", lambda t, A: t.calc_reg_deform(A["delta_vertices"]), lambda A: R.calc_reg_deform(host, A["delta_vertices"])),
        ("flip", lambda t, A: t.recon_flip(A, False), lambda A: R.recon_flip(host, A, False)),
    ]
    for tag, hip, ref in cases:
        dv_d, fn_d = _leaf(dv0), _leaf(fn0)
        Ad = {"delta_vertices": dv_d, "face_normals": fn_d, "vertices": dr.vertices_init[None].to(DEV) + dv_d}
        dv_h, fn_h = dv0.clone().double().requires_grad_(True), fn0.clone().double().requires_grad_(True)
        Ah = {"delta_vertices": dv_h, "face_normals": fn_h, "vertices": dr.vertices_init[None].double() + dv_h}
        lv, lr = hip(dr, Ad), ref(Ah)
        assert abs(float(lv) - float(lr)) <= 2e-5 * max(1.0, abs(float(lr))), tag
        (lv * 1.7).backward(); (lr * 1.7).backward()
        for gd, gh, nm in ((dv_d.grad, dv_h.grad, "delta_vertices"), (fn_d.grad, fn_h.grad, "face_normals")):
            if gh is None:
                assert gd is None or float(gd.abs().max()) == 0, (tag, nm)
                continue
            scale = max(float(gh.abs().max()), 1e-12)
            err = float((gd.cpu().double() - gh).abs().max())
            assert err <= 2e-4 * scale + 1e-9, (tag, nm, err, scale)


def test_regularization_matches_trainer_composition(pkg):
    """DiffRender.regularization (all mesh terms of an attribute set in one launch) == trainer.py:54-74 composed from the oracle."""
    import reg_oracle as R
    dr = pkg.DiffRender(os.path.join(TEMPLATES, "smpl_uv_642.npz"), 64, ratio=2)
    host = types.SimpleNamespace(flip_index=dr.flip_index, sign_init=dr.sign_init.cpu(), edges=dr.edges, edge2faces=dr.edge2faces,
                                 vertices_laplacian_matrix=dr.vertices_laplacian_matrix, ratio=dr.ratio, lambda_lpl=dr.lambda_lpl,
                                 lambda_flat=dr.lambda_flat)
    opt = types.SimpleNamespace(lambda_reg=1.0, lambda_flipz=0.1, flipL1=False, lambda_edge=0.5, lambda_depth=0.0, lambda_depthR=0.3,
                                lambda_depthC=0.2, lambda_deform=0.05, temp=2.0, L1=True, chamfer=False, azim=1.0, lambda_ic=1.0)
    B, H, W = 6, dr.render_height, dr.image_size
    sets_d, sets_h = [], []
    for seed in (0, 1, 2):
        att, _ = pkg.synthetic.synthetic_batch(dr.vertices_init, B, H, W, seed=seed)
        g = torch.Generator().manual_seed(seed)
        fn = torch.nn.functional.normalize(torch.randn(B, dr.num_faces, 3, generator=g), dim=2)
        dvh = att["delta_vertices"].clone().requires_grad_(True)
        h = dict(att); h.update(delta_vertices=dvh, vertices=dr.vertices_init[None] + dvh, face_normals=fn.clone().requires_grad_(True))
        dvd = att["delta_vertices"].clone().to(DEV).requires_grad_(True)
        d = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in att.items()}
        d.update(delta_vertices=dvd, vertices=dr.vertices_init[None].to(DEV) + dvd, face_normals=fn.clone().to(DEV).requires_grad_(True))
        sets_d.append(d); sets_h.append(h)
    got = dr.regularization(sets_d[0], sets_d[1], sets_d[2], opt)

    Ae, Ai, Aire = sets_h
    reg = opt.lambda_reg * (R.calc_reg_loss(host, Ae) + R.calc_reg_loss(host, Ai)) / 2.0
    flip = opt.lambda_flipz * (R.recon_flip(host, Ae, False) + R.recon_flip(host, Ai, False) + R.recon_flip(host, Aire, False)) / 3.0
    reg = reg + opt.lambda_edge * (R.calc_reg_edge(host, Ae["vertices"]) + R.calc_reg_edge(host, Ai["vertices"])) / 2.0
    reg = reg + opt.lambda_depthR * (R.calc_reg_depthR(host, Ae["vertices"], temp=opt.temp) + R.calc_reg_depthR(host, Ai["vertices"], temp=opt.temp)) / 2.0
    reg = reg + opt.lambda_depthC * (R.calc_reg_depthC(host, Ae["vertices"]) + R.calc_reg_depthC(host, Ai["vertices"])) / 2.0
    reg = reg + opt.lambda_deform * (R.calc_reg_deform(host, Ae["delta_vertices"]) + R.calc_reg_deform(host, Ai["delta_vertices"])) / 2.0
    ic = opt.lambda_ic * sum(R.recon_att(Aire, pkg.deep_copy(Ai, detach=True), L1=True, azim=1.0))
    for a, b in zip(got, (reg, flip, ic)):
        assert abs(float(a) - float(b)) <= 3e-5 * max(1.0, abs(float(b)))
    (got[0] + got[1]).backward(); (reg + flip).backward()
    for d, h in zip(sets_d, sets_h):
        for k in ("delta_vertices", "face_normals"):
            gh = h[k].grad
            if gh is None:
                continue
            scale = max(float(gh.abs().max()), 1e-12)
            assert float((d[k].grad.cpu() - gh).abs().max()) <= 3e-4 * scale + 1e-9, k


def test_mesh_reg_abi_validation(pkg):
    import ctypes
    N = pkg._native
    d = N.MMMeshRegDesc()
    assert N.lib().mm_mesh_reg_forward(ctypes.byref(d), None) == -2                 # MM_ERR_BAD_SHAPE before any GPU work
    d.B, d.V, d.F, d.E, d.terms = 2, 10, 12, 20, 1 << 3
    assert N.lib().mm_mesh_reg_query_workspace(ctypes.byref(d)) % 256 == 0
    assert N.lib().mm_mesh_reg_forward(ctypes.byref(d), None) == -1                 # DEPTH needs vertices: MM_ERR_NULL_POINTER
    d.terms = 1 << 9
    assert N.lib().mm_mesh_reg_forward(ctypes.byref(d), None) == -2


def _att_sets(z, dev):
    keys = ("delta_vertices", "azimuths", "elevations", "distances", "biases", "textures", "lights")
    A = {k: torch.from_numpy(z["A_" + k]).clone().to(dev).requires_grad_(True) for k in keys}
    A2 = {k: torch.from_numpy(z["A2_" + k]).clone().to(dev) for k in keys}
    return A, A2


def test_recon_att_matches_reference_golden(pkg):
    """DiffRender.recon_att (mm_attribute_loss_*) against the reference's own values and gradients, L1 and L2."""
    z = np.load(os.path.join(GOLDEN, "losses.npz"))
    dr = pkg.DiffRender(os.path.join(TEMPLATES, "sphere.npz"), 64)
    wrt = ("azimuths", "elevations", "distances", "biases", "delta_vertices", "textures", "lights")
    for L1 in (True, False):
        A, A2 = _att_sets(z, DEV)
        A["vertices"] = dr.vertices_init[None].to(DEV) + A["delta_vertices"]
        A2["vertices"] = dr.vertices_init[None].to(DEV) + A2["delta_vertices"]
        parts = dr.recon_att(A, A2, L1=L1, chamfer=False, azim=1)
        for nm, val in zip(("cam", "shape", "texture", "light", "bias"), parts):
            name = "recon_att_L1%d_%s" % (L1, nm)
            np.testing.assert_allclose(float(val), float(z[name]), rtol=2e-5, atol=1e-7)
            grads = torch.autograd.grad(val, [A[k] for k in wrt], allow_unused=True, retain_graph=True)
            for k, g in zip(wrt, grads):
                ref = z[name + "__d_" + k]
                got = np.zeros_like(ref) if g is None else g.cpu().numpy()
                np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-7, err_msg=name + " d/d" + k)


@pytest.mark.parametrize("B,S,L1,seed", [(48, 128, True, 0), (5, 32, False, 1)])
def test_recon_att_matches_oracle_both_sides(pkg, B, S, L1, seed):
    """Full-size textures, gradients to BOTH attribute sets, against the fp64 oracle."""
    import reg_oracle as R
    dr = pkg.DiffRender(os.path.join(TEMPLATES, "smpl_uv_642.npz"), S)
    sets = []
    for s in (seed, seed + 10):
        att, _ = pkg.synthetic.synthetic_batch(dr.vertices_init, B, S, S, seed=s)
        sets.append({k: att[k] for k in ("azimuths", "elevations", "distances", "biases", "vertices", "textures", "lights")})
    dev_sets = [{k: v.clone().to(DEV).requires_grad_(True) for k, v in s.items()} for s in sets]
    host_sets = [{k: v.clone().double().requires_grad_(True) for k, v in s.items()} for s in sets]
    got = dr.recon_att(dev_sets[0], dev_sets[1], L1=L1, chamfer=False, azim=0.7)
    ref = R.recon_att(host_sets[0], host_sets[1], L1=L1, azim=0.7)
    wts = (1.0, 0.5, 2.0, 3.0, 0.25)
    for a, b in zip(got, ref):
        assert abs(float(a) - float(b)) <= 2e-5 * max(1.0, abs(float(b)))
    sum(w * a for w, a in zip(wts, got)).backward()
    sum(w * b for w, b in zip(wts, ref)).backward()
    for d, h in zip(dev_sets, host_sets):
        for k in d:
            scale = max(float(h[k].grad.abs().max()), 1e-12)
            assert float((d[k].grad.cpu().double() - h[k].grad).abs().max()) <= 2e-4 * scale + 1e-10, k
