"""The op boundary the reference imports: every operator of 3d-magic-mirror_amd/shim -- reached through
kaolin's / pytorch3d's OWN module paths -- against the matching piece of the CPU oracle (oracle/mm_oracle.inc), forward and
backward, and the operators composed in the reference's order (networks.py:278-317) against the fused DiffRender.render.
Bar: face_idx bit-exact; values and gradients within 1e-4 (fp32)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as Fnn

from conftest import ROOT, TEMPLATES, make_inputs
from parity_bar import grad_close

pytestmark = pytest.mark.gpu
SHIM = os.path.join(ROOT, "3d-magic-mirror_amd", "shim")
LEAVES = ("vertices", "textures", "lights", "bg", "azimuths", "elevations", "distances", "biases")


@pytest.fixture(scope="module")
def kal():
    if SHIM not in sys.path:
        sys.path.insert(0, SHIM)
    import kaolin
    return kaolin


def _close(got, ref, tol=1e-4):
    """forward values (O(1) quantities): absolute bar"""
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    scale = max(1.0, float(np.abs(ref).max()))
    err = float(np.abs(got - ref).max())
    assert err <= tol * scale, (err, scale)


def _gclose(got, ref, tol=1e-4, what=""):
    """gradients: relative to the reference gradient's own maximum, no floor (tests/parity_bar.py)"""
    grad_close(got, ref, rtol=tol, what=what)


def _dev(a, grad=False):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0").requires_grad_(grad)


def _geometry(oracle, name, B, S, seed):
    inp, gt, proj = make_inputs(name, B, S, S, seed=seed)
    T = oracle.camera(inp["distances"], inp["elevations"], inp["azimuths"], inp["biases"])
    fvc, fvi, fn = oracle.prepare_vertices(inp["vertices"], inp["faces"], T, proj)
    return inp, gt, proj, T, fvc, fvi, fn


@pytest.mark.parametrize("name,B", [("sphere", 4), ("smpl_uv", 2)])
def test_prepare_vertices(kal, oracle, name, B):
    inp, gt, proj, T, fvc_o, fvi_o, fn_o = _geometry(oracle, name, B, 32, 3)
    rng = np.random.default_rng(0)
    v, Tt = _dev(inp["vertices"], True), _dev(T, True)
    faces = torch.from_numpy(inp["faces"]).long()                       # host int64, as kaolin's import_mesh returns them
    cam_proj = torch.from_numpy(proj.astype(np.float32)).reshape(3, 1)
    fvc, fvi, fn = kal.render.mesh.prepare_vertices(vertices=v, faces=faces, camera_proj=cam_proj, camera_transform=Tt)
    assert fvc.shape == fvc_o.shape and fvi.shape == fvi_o.shape and fn.shape == fn_o.shape
    assert np.array_equal(fvc.detach().cpu().numpy(), fvc_o) and np.array_equal(fvi.detach().cpu().numpy(), fvi_o)   # same expressions
    _close(fn, fn_o, 1e-6)
    d1, d2, d3 = (rng.normal(size=s).astype(np.float32) for s in (fvc_o.shape, fvi_o.shape, fn_o.shape))
    (fvc * _dev(d1)).sum().add((fvi * _dev(d2)).sum()).add((fn * _dev(d3)).sum()).backward()
    dv_o, dT_o = oracle.prepare_vertices_backward(inp["vertices"], inp["faces"], T, proj, d1, d2, d3)
    _gclose(v.grad, dv_o)
    _gclose(Tt.grad, dT_o)
    # only one of the three outputs used (the others arrive as None)
    v2 = _dev(inp["vertices"], True)
    _, fvi2, _ = kal.render.mesh.prepare_vertices(v2, faces, cam_proj, camera_transform=_dev(T))
    (fvi2 * _dev(d2)).sum().backward()
    dv2, _ = oracle.prepare_vertices_backward(inp["vertices"], inp["faces"], T, proj, None, d2, None)
    _gclose(v2.grad, dv2)
    # camera_rot / camera_trans form = (p - t) @ R^T
    R = torch.linalg.qr(torch.randn(B, 3, 3))[0]
    t = torch.randn(B, 3) * 0.1 + torch.tensor([0., 0., 3.])
    a, b2, c = kal.render.mesh.prepare_vertices(_dev(inp["vertices"]), faces, cam_proj, camera_rot=R.cuda(), camera_trans=t.cuda())
    ref = (torch.from_numpy(inp["vertices"]) - t[:, None]) @ R.transpose(1, 2)
    _close(a, ref[:, faces.reshape(-1)].reshape(B, -1, 3, 3).numpy(), 1e-5)
    with pytest.raises(AssertionError):
        kal.render.mesh.prepare_vertices(_dev(inp["vertices"]), faces, cam_proj)
    with pytest.raises(RuntimeError):
        kal.render.mesh.prepare_vertices(torch.from_numpy(inp["vertices"]), faces, cam_proj, camera_transform=torch.from_numpy(T))   # host tensors


@pytest.mark.parametrize("unit", [True, False])
def test_face_normals(kal, oracle, unit):
    inp, gt, proj, T, fvc_o, fvi_o, fn_o = _geometry(oracle, "sphere", 3, 32, 5)
    fv = _dev(fvc_o, True)
    out = kal.ops.mesh.face_normals(fv, unit=unit)
    ref_in = torch.from_numpy(fvc_o).double().requires_grad_(True)
    n = torch.cross(ref_in[:, :, 1] - ref_in[:, :, 0], ref_in[:, :, 2] - ref_in[:, :, 0], dim=2)
    if unit:
        n = n / (n.norm(dim=2, keepdim=True) + 1e-10)
        _close(out, fn_o, 1e-6)                                           # = prepare_vertices' third output
    _close(out, n.detach().numpy(), 1e-5)
    w = torch.randn(n.shape, dtype=torch.double)
    (n * w).sum().backward()
    (out * w.float().cuda()).sum().backward()
    _gclose(fv.grad, ref_in.grad.numpy())


@pytest.mark.parametrize("name,B,S,kw", [
    ("sphere", 4, 64, {}),                                                # BASELINE config 1 geometry
    ("smpl_uv_642", 3, 50, {}),                                           # ragged size
    ("smpl_uv", 2, 96, {}),                                               # 13 776 faces
    ("sphere", 3, 64, dict(knum=5, boxlen=0.08, sigmainv=900.0)),         # dibr constants away from their defaults
])
def test_dibr_rasterization(kal, oracle, name, B, S, kw):
    inp, gt, proj, T, fvc_o, fvi_o, fn_o = _geometry(oracle, name, B, S, 7)
    rng = np.random.default_rng(1)
    F = inp["faces"].shape[0]
    D_parts = (1, 2, 3)
    feats_np = [np.ones((B, F, 3, 1), np.float32), np.broadcast_to(inp["face_uvs"][None], (B, F, 3, 2)).copy(),
                rng.normal(size=(B, F, 3, 3)).astype(np.float32)]
    cat = np.concatenate(feats_np, -1)
    valid = (fn_o[..., 2] >= 0).astype(np.uint8)
    okw = {k: v for k, v in kw.items()}
    fidx_o, w_o, interp_o = oracle.rasterize(S, S, fvc_o[..., 2], fvi_o, cat, valid)
    soft_o, prob, idx, typ = oracle.soft_mask(S, S, fvi_o, fidx_o, **okw)
    fvi = _dev(fvi_o, True)
    feats = [_dev(f, True) for f in feats_np]
    (texmask, texcoord, imnormal), soft, fidx = kal.render.mesh.dibr_rasterization(
        S, S, _dev(fvc_o[..., 2]), fvi, feats, _dev(fn_o[..., 2]), **kw)
    assert fidx.dtype == torch.int64 and fidx.shape == (B, S, S) and soft.shape == (B, S, S)
    assert texmask.shape == (B, S, S, 1) and texcoord.shape == (B, S, S, 2) and imnormal.shape == (B, S, S, 3)
    assert np.array_equal(fidx.cpu().numpy(), fidx_o), int((fidx.cpu().numpy() != fidx_o).sum())
    assert (fidx_o >= 0).mean() > 0.03
    _close(torch.cat([texmask, texcoord, imnormal], -1), interp_o, 1e-6)
    _close(soft, soft_o)
    band = ((soft_o > 0.01) & (soft_o < 0.99)).mean()
    assert band > 0.003
    g_i = rng.normal(size=interp_o.shape).astype(np.float32)
    g_s = rng.normal(size=soft_o.shape).astype(np.float32)
    (torch.cat([texmask, texcoord, imnormal], -1) * _dev(g_i)).sum().add((soft * _dev(g_s)).sum()).backward()
    dfvi_o, dfeat_o = oracle.rasterize_backward(g_i, fidx_o, fvi_o, cat)
    dfvi_o = dfvi_o + oracle.soft_mask_backward(g_s, fidx_o, fvi_o, prob, idx, typ, sigmainv=kw.get("sigmainv", 7000.0))
    _gclose(fvi.grad, dfvi_o)
    _gclose(torch.cat([f.grad for f in feats], -1), dfeat_o)
    # a single tensor instead of a list comes back as a single tensor; the soft mask alone back-propagates too
    fvi2 = _dev(fvi_o, True)
    one, soft2, fidx2 = kal.render.mesh.dibr_rasterization(S, S, _dev(fvc_o[..., 2]), fvi2, _dev(cat), _dev(fn_o[..., 2]), **kw)
    assert torch.is_tensor(one) and one.shape == (B, S, S, 6) and torch.equal(fidx2, fidx) and torch.equal(soft2, soft.detach())
    (soft2 * _dev(g_s)).sum().backward()
    _gclose(fvi2.grad, oracle.soft_mask_backward(g_s, fidx_o, fvi_o, prob, idx, typ, sigmainv=kw.get("sigmainv", 7000.0)))


@pytest.mark.parametrize("bad", [float("nan"), float("inf"), -float("inf")])
def test_texture_mapping_backward_propagates_non_finite_upstream_gradients(kal, bad):
    """The deterministic texture-gradient scatter (64-bit fixed point at a scale taken from the image's largest |upstream value|) has no scale for
    a NaN / inf upstream gradient: the image's WHOLE texture gradient then comes out as NaN -- loud, as ATen's grid_sampler backward and kaolin
    propagate it at the texels touched -- never as zeros (a training loop's NaN / inf check on the gradients must keep working).  The other
    images of the batch are unaffected, and so is the coordinate gradient's own propagation."""
    g = torch.Generator().manual_seed(7)
    B, N, C, Ht, Wt = 3, 500, 3, 32, 16
    uv = torch.rand(B, N, 2, generator=g)
    tex = torch.rand(B, C, Ht, Wt, generator=g)
    dout = torch.randn(B, N, C, generator=g)
    clean_uv, clean_tex = uv.cuda().requires_grad_(True), tex.cuda().requires_grad_(True)
    kal.render.mesh.texture_mapping(clean_uv, clean_tex, mode='bilinear').backward(dout.cuda())
    dout_bad = dout.clone(); dout_bad[1, 137, 2] = bad
    uvd, texd = uv.cuda().requires_grad_(True), tex.cuda().requires_grad_(True)
    kal.render.mesh.texture_mapping(uvd, texd, mode='bilinear').backward(dout_bad.cuda())
    assert torch.isnan(texd.grad[1]).all()
    assert torch.equal(texd.grad[0], clean_tex.grad[0]) and torch.equal(texd.grad[2], clean_tex.grad[2])
    assert not torch.isfinite(uvd.grad[1, 137]).all() and torch.equal(uvd.grad[0], clean_uv.grad[0])
    keep = torch.ones(N, dtype=torch.bool); keep[137] = False
    assert torch.equal(uvd.grad[1][keep.cuda()], clean_uv.grad[1][keep.cuda()])


def test_texture_mapping(kal, oracle):
    g = torch.Generator().manual_seed(5)
    B, H, W, C, Ht, Wt = 3, 20, 24, 3, 32, 16
    uv = torch.rand(B, H, W, 2, generator=g) * 1.3 - 0.15                   # includes out-of-range -> border padding
    uv[0, 0, :4] = torch.tensor([[0., 0.], [1., 1.], [0.5, 0.5], [1.0, 0.0]])
    tex = torch.rand(B, C, Ht, Wt, generator=g)
    dout = torch.randn(B, H, W, C, generator=g)
    dout[1] = 0                                                              # a whole image of zero upstream gradient (uncovered pixels)
    uvd, texd = uv.cuda().requires_grad_(True), tex.cuda().requires_grad_(True)
    out = kal.render.mesh.texture_mapping(uvd, texd, mode='bilinear')
    assert out.shape == (B, H, W, C)
    ref = oracle.texture_mapping(uv.reshape(B, -1, 2).numpy(), tex.numpy())
    assert np.array_equal(out.detach().cpu().numpy().reshape(B, -1, C), ref)                 # same expressions as the oracle
    out.backward(dout.cuda())
    duv, dtex = oracle.texture_mapping_backward(uv.reshape(B, -1, 2).numpy(), tex.numpy(), dout.reshape(B, -1, C).numpy())
    _gclose(uvd.grad.reshape(B, -1, 2), duv)
    _gclose(texd.grad, dtex)
    # (B,N,2) coordinates and nearest mode vs torch's own grid_sample
    uv2 = uv.reshape(B, -1, 2)
    near = kal.render.mesh.texture_mapping(uv2.cuda(), tex.cuda())                          # default mode is 'nearest', as upstream
    grid = torch.stack([uv2[..., 0] * 2 - 1, -(uv2[..., 1] * 2 - 1)], -1).unsqueeze(2)
    refn = Fnn.grid_sample(tex, grid, mode="nearest", align_corners=False, padding_mode="border").permute(0, 2, 3, 1).reshape(B, -1, C)
    assert near.shape == (B, H * W, C) and torch.equal(near.cpu(), refn)
    with pytest.raises(ValueError):
        kal.render.mesh.texture_mapping(uv2.cuda(), tex.cuda(), mode='bicubic')


def test_spherical_harmonic_lighting(kal, oracle):
    g = torch.Generator().manual_seed(2)
    B, H, W = 3, 16, 10
    n = torch.randn(B, H, W, 3, generator=g)
    n[0, 0] = 0                                                              # uncovered pixels carry zero normals
    lights = torch.randn(B, 9, generator=g)
    nd, ld = n.cuda().requires_grad_(True), lights.cuda().requires_grad_(True)
    out = kal.render.mesh.spherical_harmonic_lighting(nd, ld)
    assert out.shape == (B, H, W)
    ref = oracle.sh_lighting(n.reshape(B, -1, 3).numpy(), lights.numpy())
    assert np.array_equal(out.detach().cpu().numpy().reshape(B, -1), ref)
    dc = torch.randn(B, H, W, generator=g)
    out.backward(dc.cuda())
    dn, dl = oracle.sh_lighting_backward(n.reshape(B, -1, 3).numpy(), lights.numpy(), dc.reshape(B, -1).numpy())
    _gclose(nd.grad.reshape(B, -1, 3), dn)
    _gclose(ld.grad, dl)


def test_mask_iou(kal, oracle):
    g = torch.Generator().manual_seed(9)
    B, H, W = 5, 40, 24
    a = torch.rand(B, H, W, generator=g)
    b = (torch.rand(B, H, W, generator=g) > 0.5).float()
    ad, bd = a.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    loss = kal.metrics.render.mask_iou(ad, bd)
    a2, b2 = a.double().requires_grad_(True), b.double().requires_grad_(True)
    mul = a2 * b2
    ref = 1.0 - torch.mean(mul.reshape(B, -1).sum(1) / (((a2 + b2) - mul).reshape(B, -1).sum(1) + 1e-10))
    assert abs(float(loss) - float(ref)) < 1e-6
    (loss * 3.0).backward(); (ref * 3.0).backward()
    _gclose(ad.grad, a2.grad.numpy()); _gclose(bd.grad, b2.grad.numpy())
    # the oracle's recon_data with image_weight 0 is exactly this term
    m = np.zeros((B, 4, H, W), np.float32); m[:, 3] = a.numpy()
    t = np.zeros((B, 4, H, W), np.float32); t[:, 3] = b.numpy()
    assert abs(oracle.recon_data(m, t, image_weight=0.0) - float(loss)) < 1e-6
    # identical / disjoint masks, and the evaluation loop's host tensors (trainer.py:793): computed on the device, returned on the host
    assert abs(float(kal.metrics.render.mask_iou(bd.detach(), bd.detach()))) < 1e-6
    assert abs(float(kal.metrics.render.mask_iou(bd.detach(), 1 - bd.detach())) - 1.0) < 1e-6
    host = kal.metrics.render.mask_iou(a[:1], b[:1])
    assert not host.is_cuda and abs(float(host) - float(kal.metrics.render.mask_iou(a[:1].cuda(), b[:1].cuda()))) < 1e-7
    with pytest.raises(RuntimeError):
        kal.metrics.render.mask_iou(a.cuda(), b[:2].cuda())


def test_chamfer_through_pytorch3d_path(kal):
    if SHIM not in sys.path:
        sys.path.insert(0, SHIM)
    from pytorch3d.loss import chamfer_distance
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 642, 3, generator=g).cuda().requires_grad_(True)
    y = (torch.randn(4, 642, 3, generator=g) * 0.9).cuda()
    loss, nrm = chamfer_distance(x, y)
    d = torch.cdist(x.detach().double(), y.double()).pow(2)
    assert nrm is None and abs(float(loss) - float(d.min(2)[0].mean(1).mean(0) + d.min(1)[0].mean(1).mean(0))) < 1e-5
    loss.backward()
    assert float(x.grad.abs().max()) > 0
    with pytest.raises(NotImplementedError):
        chamfer_distance(x, y, point_reduction="sum")


def _reference_order_render(kal, dr, att, T, no_mask):
    """The operators in the order DiffRender.render composes them (/root/reference/networks.py:284-317), through the shim.
    T = the look-at transform of networks.py:281-282 (smr_utils, pure torch in the reference), handed in as a leaf."""
    dev = att["azimuths"].device
    B = att["azimuths"].shape[0]
    fvc, fvi, fn = kal.render.mesh.prepare_vertices(vertices=att["vertices"], faces=dr.faces, camera_proj=dr.cam_proj, camera_transform=T)
    nrm = kal.ops.mesh.face_normals(fvc, unit=True).unsqueeze(-2).repeat(1, 1, 3, 1)
    feats = [torch.ones((B, dr.num_faces, 3, 1), device=dev), dr.face_uvs.to(dev).repeat(B, 1, 1, 1), nrm]
    (texmask, texcoord, imnormal), soft, fidx = kal.render.mesh.dibr_rasterization(
        dr.render_height, dr.image_size, fvc[:, :, :, -1], fvi, feats, fn[:, :, -1])
    texcolor = kal.render.mesh.texture_mapping(texcoord, att["textures"], mode='bilinear')
    coef = kal.render.mesh.spherical_harmonic_lighting(imnormal, att["lights"])
    if no_mask:
        image = (texcolor * texmask + att["bg"].permute(0, 2, 3, 1) * (1 - texmask)) * coef.unsqueeze(-1)
    else:
        image = texcolor * texmask * coef.unsqueeze(-1) + torch.ones_like(texcolor) * (1 - texmask)
    rgbs = torch.cat([torch.clamp(image, 0, 1), soft[..., None]], -1).permute(0, 3, 1, 2)
    return rgbs, fn, fidx


@pytest.mark.parametrize("name,B,S,no_mask,seed", [("sphere", 4, 64, True, 0), ("smpl_uv_642", 3, 48, False, 2)])
def test_operators_composed_like_the_reference_match_the_fused_render(pkg, kal, oracle, name, B, S, no_mask, seed):
    dev = torch.device("cuda:0")
    dr = pkg.DiffRender(os.path.join(TEMPLATES, name + ".npz"), S)
    att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, B, S, S, seed=seed)
    A1 = {k: (v.to(dev).requires_grad_(k in LEAVES) if torch.is_tensor(v) else v) for k, v in att.items()}
    A2 = {k: (v.to(dev).requires_grad_(k in LEAVES) if torch.is_tensor(v) else v) for k, v in att.items()}
    gtd = gt.to(dev)
    # the camera transform exactly as the library builds it (oracle.camera is pinned to smr_utils' golden and evaluates the same
    # fp32 expressions as the vertex stage), as a leaf: its gradient is pushed through the camera chain by the oracle below
    Tn = oracle.camera(att["distances"].numpy(), att["elevations"].numpy(), att["azimuths"].numpy(), att["biases"].numpy())
    T = _dev(Tn, True)
    r1, fn1, fidx1 = _reference_order_render(kal, dr, A1, T, no_mask)
    r2, out2 = dr.render(no_mask=no_mask, **A2)
    if not torch.equal(fidx1.int(), dr.last_face_idx):                  # the SAME walk serves both boundaries
        f1, f2 = fidx1.int(), dr.last_face_idx
        d = (f1 != f2).nonzero()
        inp = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in att.items()}
        inp["faces"] = dr.faces.numpy().astype(np.int32); inp["face_uvs"] = dr.face_uvs.numpy()[0]
        fo = torch.from_numpy(oracle.render_forward(inp, S, S, no_mask, dr.cam_proj.numpy().reshape(3))[1]).to(dev)
        f2b = dr.render(no_mask=no_mask, **A2)[1] and dr.last_face_idx
        extra = ""
        raise AssertionError("face_idx differs between the boundaries at %d pixels; first (b,y,x): %s; dibr %s; fused %s; vs the oracle: dibr %d, "
                             "fused %d wrong pixels; fused rendered again: %d wrong"
                             % (d.shape[0], d[:12].tolist(), f1[f1 != f2][:12].tolist(), f2[f1 != f2][:12].tolist(),
                                int((f1 != fo).sum()), int((f2 != fo).sum()), int((f2b != fo).sum())) + extra)
    assert torch.equal(r1[:, 3], r2[:, 3].detach())                     # ... the soft mask too, bit for bit
    _close(r1, r2.detach().cpu().numpy(), 1e-6)
    w = torch.randn(B, dr.num_faces, 3, device=dev) * 1e-3
    l1 = kal.metrics.render.mask_iou(r1[:, 3], gtd[:, 3]) + 0.1 * (r1[:, :3] * gtd[:, 3:4] - gtd[:, :3] * gtd[:, 3:4]).abs().mean() + (fn1 * w).sum()
    l2 = dr.recon_data(r2, gtd, no_mask=no_mask) + (out2["face_normals"] * w).sum()
    assert abs(float(l1) - float(l2)) < 2e-5
    l1.backward(); l2.backward()
    for k in ("vertices", "textures", "lights", "bg"):
        if k == "bg" and not no_mask:
            continue
        _gclose(A1[k].grad, A2[k].grad.cpu().numpy(), 2e-4)
        assert float(A2[k].grad.abs().max()) > 0
    dd, de, da, db = oracle.camera_backward(att["distances"].numpy(), att["elevations"].numpy(), att["azimuths"].numpy(), att["biases"].numpy(),
                                            T.grad.cpu().numpy())
    for k, ref in (("distances", dd), ("elevations", de), ("azimuths", da), ("biases", db)):
        _gclose(A2[k].grad, ref, 2e-4)


@pytest.mark.parametrize("name", ["sphere", "smpl_uv"])
def test_device_built_vertex_corner_csr_is_the_host_builders(kal, name):
    """prepare_vertices builds its vertex -> corner CSR on the device, every call, without reading anything back (the reference hands it a
    fresh `faces` tensor per render): offsets and items are those of the host builder, list order included, and the build is repeatable."""
    import importlib
    ops = importlib.import_module("3d-magic-mirror_amd.ops")
    tmpl = importlib.import_module("3d-magic-mirror_amd.template")
    z = np.load(os.path.join(TEMPLATES, name + ".npz"))
    faces = torch.from_numpy(z["faces"].astype(np.int64))
    V = int(z["vertices"].shape[0])
    off_h, items_h = tmpl.vertex_corner_adjacency(V, faces)
    for _ in range(3):
        fi, off, items = ops._faces_tables(faces.cuda(), V, torch.device("cuda:0"))
        assert torch.equal(off.cpu().long(), off_h.long()) and torch.equal(items.cpu().long(), items_h.long())
        assert torch.equal(fi.cpu().long(), faces)


def test_vertex_ids_outside_the_cloud_are_reported_not_dereferenced(kal):
    """A face list that indexes a vertex the cloud does not have: nothing is read out of bounds, the device reports it through a pinned
    status word, and the NEXT prepare_vertices raises -- no synchronisation on the hot path."""
    import importlib
    ops = importlib.import_module("3d-magic-mirror_amd.ops")
    z = np.load(os.path.join(TEMPLATES, "sphere.npz"))
    faces = torch.from_numpy(z["faces"].astype(np.int64)).cuda()
    V = int(z["vertices"].shape[0])
    verts = torch.from_numpy(z["vertices"].astype(np.float32)).cuda()[None].repeat(2, 1, 1)
    T = torch.eye(4, 3, device="cuda:0")[None].repeat(2, 1, 1).contiguous()
    T[:, 3, 2] = -3.0
    proj = torch.tensor([[2.5], [2.5], [-1.0]], device="cuda:0")
    bad = faces.clone(); bad[5, 1] = V + 7
    fvc, fvi, fn = kal.render.mesh.prepare_vertices(verts, bad, proj, camera_transform=T)
    torch.cuda.synchronize()
    assert torch.isfinite(fvc).all()
    with pytest.raises(RuntimeError, match="outside"):
        kal.render.mesh.prepare_vertices(verts, faces, proj, camera_transform=T)
    kal.render.mesh.prepare_vertices(verts, faces, proj, camera_transform=T)       # reported once
    # ... or by the offending call's own backward (the device has long finished the forward's table builder by then), or by the poll function
    v2 = verts.clone().requires_grad_(True)
    fvc, fvi, fn = kal.render.mesh.prepare_vertices(v2, bad, proj, camera_transform=T)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="outside"):
        fvi.sum().backward()
    assert ops.poll_reported_faces("cuda:0", synchronize=True) is None                # (reported once)
    kal.render.mesh.prepare_vertices(verts, bad, proj, camera_transform=T)
    with pytest.raises(RuntimeError, match="outside"):
        ops.poll_reported_faces("cuda:0", synchronize=True)                           # waits for the device: no later call needed
    assert ops.poll_reported_faces() is None


def test_the_unfused_operator_chain_is_bitwise_reproducible(pkg, kal):
    """The compatibility path end to end (shim_chain: the reference's render through the kaolin-shaped operators + recon_data + backward): no
    float atomic is left in it -- dibr_rasterization's backward adds per lane in registers and by a fixed butterfly, texture_mapping's scatters
    in 64-bit fixed point, spherical_harmonic_lighting's light gradient and prepare_vertices' gathers reduce in fixed orders (the vertex ->
    corner lists are put in ascending order on the device) -- so two runs on the same inputs give the same bits for every gradient."""
    import importlib
    chain = importlib.import_module("3d-magic-mirror_amd.shim_chain")
    dev = torch.device("cuda:0")
    dr = pkg.DiffRender(os.path.join(TEMPLATES, "smpl_uv_642.npz"), 96)
    att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, 6, 96, 96, seed=5)
    gtd = gt.to(dev)
    runs = []
    for rep in range(3):
        A = {k: (v.to(dev).requires_grad_(k in LEAVES) if torch.is_tensor(v) else v) for k, v in att.items()}
        rgbs, fn, fidx = chain.render(dr, no_mask=True, **A)
        (chain.recon_data(dr, rgbs, gtd) + 1e-3 * fn.sum()).backward()
        torch.cuda.synchronize()
        runs.append((rgbs.detach().clone(), {k: A[k].grad.clone() for k in LEAVES}))
    for k in LEAVES:
        assert float(runs[0][1][k].abs().max()) > 0, k
        assert torch.equal(runs[0][1][k], runs[1][1][k]) and torch.equal(runs[0][1][k], runs[2][1][k]), k
    assert torch.equal(runs[0][0], runs[1][0])
