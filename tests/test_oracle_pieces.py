"""Known-answer and cross-implementation tests that pin the oracle's building blocks (last row)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as Fnn

from conftest import make_inputs


def _tri(*pts):
    return np.asarray(pts, np.float32).reshape(1, -1, 3, 2)


def test_raster_single_screen_filling_triangle(oracle):
    fvi = _tri([-3, -3], [3, -3], [0, 3])
    fz = np.full((1, 1, 3), -2.0, np.float32)
    feats = np.eye(3, dtype=np.float32).reshape(1, 1, 3, 3)
    fidx, w, out = oracle.rasterize(8, 8, fz, fvi, feats, np.ones((1, 1), np.uint8))
    assert (fidx == 0).all()
    np.testing.assert_allclose(w.sum(-1), 1.0, atol=1e-6)
    np.testing.assert_allclose(out, w, atol=0)          # identity features reproduce the weights
    assert (w >= 0).all()


def test_raster_pixel_centre_convention_2x2(oracle):
    # pixel centres of a 2x2 image sit at ndc (+-0.5, +-0.5); row 0 is the TOP (y = +0.5)
    fvi = _tri([0.0, 0.0], [1.0, 0.0], [0.0, 1.0])        # covers only the centre (+0.25..), i.e. px=1, py=0? no centre inside
    fz = np.full((1, 1, 3), -1.0, np.float32)
    feats = np.ones((1, 1, 3, 1), np.float32)
    fidx, _, _ = oracle.rasterize(2, 2, fz, fvi, feats, np.ones((1, 1), np.uint8))
    # (0.5,0.5) lies exactly on the hypotenuse x+y=1 -> inclusive edge -> covered; it is pixel (row 0, col 1)
    assert fidx[0].tolist() == [[-1, 0], [-1, -1]]
    fvi2 = _tri([-1, -1], [0, -1], [-1, 0])               # lower-left corner; centre (-0.5,-0.5) on its hypotenuse
    fidx2, _, _ = oracle.rasterize(2, 2, fz, fvi2, feats, np.ones((1, 1), np.uint8))
    assert fidx2[0].tolist() == [[-1, -1], [0, -1]]


def test_raster_depth_order_ties_and_culling(oracle):
    big = [[-3, -3], [3, -3], [0, 3]]
    fvi = _tri(big, big, big)
    feats = np.ones((1, 3, 3, 1), np.float32)
    fz = np.asarray([[-5, -5, -5], [-2, -2, -2], [-2, -2, -2]], np.float32).reshape(1, 3, 3)
    fidx, _, _ = oracle.rasterize(4, 4, fz, fvi, feats, np.ones((1, 3), np.uint8))
    assert (fidx == 1).all()                               # nearest = largest z; tie 1 vs 2 -> lowest index
    fidx, _, _ = oracle.rasterize(4, 4, fz, fvi, feats, np.asarray([[1, 0, 1]], np.uint8))
    assert (fidx == 2).all()                               # culled face 1 never wins colour
    fidx, w, out = oracle.rasterize(4, 4, fz, fvi, feats, np.zeros((1, 3), np.uint8))
    assert (fidx == -1).all() and (w == 0).all() and (out == 0).all()
    soft, prob, idx, typ = oracle.soft_mask(4, 4, fvi, fidx)
    assert (idx[..., :3] >= 0).all()                       # ...but culled faces still feed the silhouette
    # clockwise winding is handled by copysign(eps, norm): same coverage
    cw = _tri(big[0], big[2], big[1])
    f2, _, _ = oracle.rasterize(4, 4, fz[:, :1], cw, feats[:, :1], np.ones((1, 1), np.uint8))
    assert (f2 == 0).all()


def _brute_soft(fvi, H, W, covered, sigmainv=7000.0, boxlen=0.02, knum=30):
    """fp64 numpy brute force of SURVEY 8(a)-a8 soft mask, NDC units."""
    out = np.zeros((H, W))
    F = fvi.shape[0]
    for py in range(H):
        for px in range(W):
            if covered[py, px]:
                out[py, px] = 1.0
                continue
            p = np.array([(2 * px + 1 - W) / W, (H - 2 * py - 1) / H])
            keep, cnt = 1.0, 0
            for f in range(F):
                t = fvi[f].astype(np.float64)
                lo, hi = t.min(0) - boxlen, t.max(0) + boxlen
                if (p < lo).any() or (p > hi).any():
                    continue
                d2 = np.inf
                for e in range(3):
                    u, v = t[e], t[(e + 1) % 3]
                    s = np.clip(np.dot(p - u, v - u) / np.dot(v - u, v - u), 0, 1)
                    d2 = min(d2, np.sum((p - (u + s * (v - u))) ** 2))
                keep *= 1 - np.exp(-d2 * sigmainv)
                cnt += 1
                if cnt >= knum:
                    break
            out[py, px] = 1 - keep
    return out


@pytest.mark.parametrize("knum", [30, 3])
def test_soft_mask_vs_bruteforce(oracle, knum):
    inp, _, proj = make_inputs("sphere", 2, 24, 24, seed=3)
    T = oracle.camera(inp["distances"], inp["elevations"], inp["azimuths"], inp["biases"])
    fvc, fvi, fn = oracle.prepare_vertices(inp["vertices"], inp["faces"], T, proj)
    feats = np.ones(fvi.shape[:3] + (1,), np.float32)
    fidx, _, _ = oracle.rasterize(24, 24, fvc[..., 2], fvi, feats, (fn[..., 2] >= 0).astype(np.uint8))
    soft, prob, idx, typ = oracle.soft_mask(24, 24, fvi, fidx, knum=knum)
    for b in range(2):
        ref = _brute_soft(fvi[b], 24, 24, fidx[b] >= 0, knum=knum)
        np.testing.assert_allclose(soft[b], ref, atol=5e-4)
    band = (soft > 0) & (soft < 1)
    assert band.sum() > 20
    if knum == 3:
        assert ((idx >= 0).sum(-1) == 3).any()             # truncation actually exercised
    soft64, *_ = oracle.soft_mask(24, 24, fvi, fidx, knum=knum, dtype=np.float64)
    np.testing.assert_allclose(soft64, soft, atol=5e-4)


def test_texture_mapping_matches_grid_sample(oracle):
    g = torch.Generator().manual_seed(5)
    B, N, C, Ht, Wt = 2, 500, 3, 16, 12
    uv = (torch.rand(B, N, 2, generator=g) * 1.3 - 0.15).double()     # includes out-of-range -> border padding
    uv[0, :4] = torch.tensor([[0., 0.], [1., 1.], [0.5, 0.5], [1.0, 0.0]], dtype=torch.double)
    tex = torch.rand(B, C, Ht, Wt, generator=g).double()
    dout = torch.rand(B, N, C, generator=g).double()
    uvt, text = uv.clone().requires_grad_(True), tex.clone().requires_grad_(True)
    grid = uvt.reshape(B, N, 1, 2) * 2.0 - 1.0
    grid = torch.stack([grid[..., 0], -grid[..., 1]], -1)
    ref = Fnn.grid_sample(text, grid, mode="bilinear", align_corners=False, padding_mode="border")   # (B,C,N,1)
    ref = ref.permute(0, 2, 3, 1).reshape(B, N, C)
    ref.backward(dout)
    out = oracle.texture_mapping(uv.numpy(), tex.numpy(), dtype=np.float64)
    np.testing.assert_allclose(out, ref.detach().numpy(), atol=1e-12)
    duv, dtex = oracle.texture_mapping_backward(uv.numpy(), tex.numpy(), dout.numpy(), dtype=np.float64)
    np.testing.assert_allclose(dtex, text.grad.numpy(), atol=1e-12)
    np.testing.assert_allclose(duv, uvt.grad.numpy(), atol=1e-10)
    out32 = oracle.texture_mapping(uv.numpy().astype(np.float32), tex.numpy().astype(np.float32))
    np.testing.assert_allclose(out32, ref.detach().numpy(), atol=2e-5)


def test_sh_lighting_closed_form(oracle):
    lights = np.arange(1, 10, dtype=np.float64).reshape(1, 9) / 10
    n = np.asarray([[[1, 0, 0], [0, 1, 0], [0, 0, 1], [0, 0, 0]]], np.float64)
    c = oracle.sh_lighting(n, lights, dtype=np.float64)[0]
    l = lights[0]
    k0, k1, k6a, k6b, k8 = 0.28209479177, 0.4886025119, 0.94617469575, 0.31539156525, 0.38627420202
    np.testing.assert_allclose(c[0], k0 * l[0] + k1 * l[1] - k6b * l[6] + k8 * l[8], rtol=1e-12)
    np.testing.assert_allclose(c[1], k0 * l[0] + k1 * l[3] - k6b * l[6] - k8 * l[8], rtol=1e-12)
    np.testing.assert_allclose(c[2], k0 * l[0] + k1 * l[2] + (k6a - k6b) * l[6], rtol=1e-12)
    np.testing.assert_allclose(c[3], k0 * l[0] - k6b * l[6], rtol=1e-12)       # uncovered pixels (zero normal)
    # backward vs autograd of the same closed form in torch
    g = torch.Generator().manual_seed(1)
    nn_ = torch.randn(2, 7, 3, generator=g, dtype=torch.double, requires_grad=True)
    ll = torch.randn(2, 9, generator=g, dtype=torch.double, requires_grad=True)
    x, y, z = nn_[..., 0], nn_[..., 1], nn_[..., 2]
    bands = torch.stack([0.28209479177 * torch.ones_like(x), 0.4886025119 * x, 0.4886025119 * z, 0.4886025119 * y,
                         1.09254843059 * x * y, 1.09254843059 * y * z, 0.94617469575 * z * z - 0.31539156525,
                         0.77254840404 * x * z, 0.38627420202 * (x * x - y * y)], -1)
    coef = (bands * ll[:, None]).sum(-1)
    dc = torch.randn(2, 7, generator=g, dtype=torch.double)
    coef.backward(dc)
    dn, dl = oracle.sh_lighting_backward(nn_.detach().numpy(), ll.detach().numpy(), dc.numpy(), dtype=np.float64)
    np.testing.assert_allclose(dn, nn_.grad.numpy(), atol=1e-12)
    np.testing.assert_allclose(dl, ll.grad.numpy(), atol=1e-12)


def test_mask_iou_identical_and_disjoint(oracle):
    B, H, W = 2, 8, 8
    m = np.zeros((B, 4, H, W), np.float32); m[:, 3, :4] = 1; m[:, :3] = 0.3
    assert abs(oracle.recon_data(m, m, image_weight=0.0)) < 1e-6                 # identical masks -> 1 - 1
    d = m.copy(); d[:, 3] = 1 - m[:, 3]
    assert abs(oracle.recon_data(d, m, image_weight=0.0) - 1.0) < 1e-6           # disjoint -> 1 - 0
    e = np.zeros_like(m)
    assert abs(oracle.recon_data(e, e, image_weight=0.0) - 1.0) < 1e-6           # empty/empty: 0/(0+1e-10) -> loss 1


def test_prepare_vertices_matches_torch_restatement(oracle):
    inp, _, proj = make_inputs("sphere", 3, 8, 8, seed=2)
    T = oracle.camera(inp["distances"], inp["elevations"], inp["azimuths"], inp["biases"])
    fvc, fvi, fn = oracle.prepare_vertices(inp["vertices"], inp["faces"], T, proj)
    v = torch.from_numpy(inp["vertices"]); Tt = torch.from_numpy(T); f = torch.from_numpy(inp["faces"]).long()
    vc = torch.nn.functional.pad(v, (0, 1), value=1.0) @ Tt
    pp = vc * torch.from_numpy(proj).view(1, 1, 3)
    vi = pp[..., :2] / pp[..., 2:3]
    fc = vc[:, f.reshape(-1)].reshape(3, -1, 3, 3)
    n = torch.cross(fc[:, :, 1] - fc[:, :, 0], fc[:, :, 2] - fc[:, :, 0], dim=2)
    n = n / (n.norm(dim=2, keepdim=True) + 1e-10)
    np.testing.assert_allclose(fvc, fc.numpy(), atol=2e-6)
    np.testing.assert_allclose(fvi, vi[:, f.reshape(-1)].reshape(3, -1, 3, 2).numpy(), atol=2e-6)
    np.testing.assert_allclose(fn, n.numpy(), atol=2e-5)
    assert (fvc[..., 2] < 0).all()                         # camera looks down -z
