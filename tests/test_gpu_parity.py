"""Parity proper: the HIP path (through the C ABI, via the DiffRender mirror) against the CPU oracle on the same seeded
inputs.  Bar (BASELINE.json north_star): face_idx bit-exact; RGBA within 1e-4 (absolute: image values are O(1)); every input gradient
within 1e-4 OF THE GRADIENT'S OWN MAXIMUM (max|got - ref| <= 1e-4 * max|ref|, no floor: tests/parity_bar.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, TEMPLATES
from parity_bar import grad_close, rel_errors

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEAVES = ("vertices", "textures", "lights", "bg", "azimuths", "elevations", "distances", "biases")


def _setup(pkg, name, B, S, ratio=1, ell=1, seed=0, no_mask=True, imn=True):
    dev = torch.device("cuda:0")
    dr = pkg.DiffRender(os.path.join(TEMPLATES, name + ".npz"), S, ratio=ratio, init_ellipsoid=ell, emit_imnormal=imn)
    H, W = dr.render_height, dr.image_size
    att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, B, H, W, seed=seed)
    datt = {k: (v.to(dev).requires_grad_(k in LEAVES) if torch.is_tensor(v) else v) for k, v in att.items()}
    inp = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in att.items()}
    inp["faces"] = dr.faces.numpy().astype(np.int32)
    inp["face_uvs"] = dr.face_uvs.numpy()[0]
    return dr, att, datt, gt, inp, dr.cam_proj.numpy().reshape(3), H, W, dev


def _close(got, ref, tol=1e-4):
    """FORWARD values (image channels, normals: O(1) quantities): absolute bar, as north_star states it for RGBA."""
    scale = max(1.0, float(np.abs(ref).max()))
    err = float(np.abs(got - ref).max())
    assert err <= tol * scale, (err, scale)


def _gclose(got, ref, tol=1e-4, what=""):
    """GRADIENTS: max|got - ref| <= tol * max|ref|, NO floor of 1 (tests/parity_bar.py says why): a zero, a mis-scaled or a partly
    missing gradient fails whatever the gradient's magnitude."""
    grad_close(got, ref, rtol=tol, what=what)


@pytest.mark.parametrize("name,B,S,ratio,no_mask,seed", [
    ("sphere", 4, 64, 1, True, 0),          # BASELINE config 1
    ("sphere", 4, 64, 1, False, 1),
    ("smpl_uv_642", 3, 32, 2, True, 2),     # Market shape: H = 2W
    ("ellipsoid", 2, 96, 1, True, 3),
    ("sphere", 2, 50, 1, True, 4),          # ragged: not a multiple of the 16x16 block
    ("smpl_uv", 2, 96, 1, True, 5),         # 13 776 faces: 216 mask words (multi-group walk), 16-px screen bins
    ("sphere2", 2, 40, 1, False, 6),        # 5 120 faces, white background
    ("smpl_uv_642", 48, 128, 1, True, 0),   # BASELINE config 2 at FULL size: the bench's batch, pixel for pixel
    ("smpl_uv_642", 48, 64, 2, True, 7),    # BASELINE config 4 (Market 128x64) at full size
    ("ellipsoid", 48, 256, 1, True, 8),     # BASELINE config 3's render at FULL size (B=48, 256x256, texture 512x256)
])
def test_render_loss_backward_matches_oracle(pkg, oracle, name, B, S, ratio, no_mask, seed):
    dr, att, datt, gt, inp, proj, H, W, dev = _setup(pkg, name, B, S, ratio=ratio, seed=seed, no_mask=no_mask)
    rgbs, out = dr.render(no_mask=no_mask, **datt)
    assert rgbs.shape == (B, 4, H, W) and rgbs.stride() == (H * W * 4, 1, W * 4, 4)      # NCHW view of NHWC memory
    wfn = torch.from_numpy(np.random.default_rng(seed).normal(size=(B, dr.num_faces, 3)).astype(np.float32) * 1e-3)
    loss = dr.recon_data(rgbs, gt.to(dev), no_mask=no_mask) + (out["face_normals"] * wfn.to(dev)).sum()
    loss.backward()
    torch.cuda.synchronize()

    rgba_o, fidx_o, fn_o, imn_o = oracle.render_forward(inp, H, W, no_mask, proj)
    loss_o, dpred = oracle.recon_data(rgba_o.transpose(0, 3, 1, 2), gt.numpy(), image_weight=dr.image_weight, want_grad=True)
    g_o = oracle.render_backward(inp, H, W, no_mask, proj, np.ascontiguousarray(dpred.transpose(0, 2, 3, 1)), wfn.numpy())

    fidx = dr.last_face_idx.cpu().numpy()
    assert (fidx == fidx_o).all(), "face_idx mismatches: %d" % int((fidx != fidx_o).sum())
    assert (fidx >= 0).mean() > 0.03
    _close(rgbs.detach().permute(0, 2, 3, 1).cpu().numpy(), rgba_o)
    _close(out["face_normals"].detach().cpu().numpy(), fn_o, 1e-6)
    _close(out["imnormal"].cpu().numpy(), imn_o, 1e-6)
    assert abs(float(loss) - (loss_o + float((fn_o * wfn.numpy()).sum()))) < 2e-5
    for k in LEAVES:
        if k == "bg" and not no_mask:
            assert datt[k].grad is None
            continue
        _gclose(datt[k].grad.cpu().numpy(), g_o[k], what=k)
        assert np.abs(g_o[k]).max() > 0


@pytest.mark.parametrize("label,name,B,S,ratio,no_mask,seed", [
    ("config 1", "sphere", 4, 64, 1, True, 0),
    ("config 2 (full size)", "smpl_uv_642", 48, 128, 1, True, 0),
    ("config 5 (one image)", "smpl_uv", 1, 512, 1, True, 0),
    ("config 3's render (full size)", "ellipsoid", 48, 256, 1, True, 8),
    ("config 4, Market 128x64 (full size)", "smpl_uv_642", 48, 64, 2, True, 7),
    ("white background", "sphere", 3, 50, 1, False, 4),
])
def test_fused_backward_under_a_unit_scale_upstream_gradient_matches_oracle(pkg, oracle, label, name, B, S, ratio, no_mask, seed):
    """The product path's backward (pixel_bwd -> gather_bwd -> vertex_bwd through mm_render_backward) driven by an O(1) upstream gradient --
    (rgbs * w).sum() + (face_normals * wfn).sum(), w and wfn ~ N(0,1) -- instead of the batch-mean loss, whose gradients are 1e-8 ... 1e-2
    large: every one of the eight input gradients is O(1) ... O(1e4) here and must agree with the oracle's to 1e-4 of its own maximum.
    (A case the fp32 oracle itself cannot hold against its float64 form would be decided by the float64 backward and pass as 'cond':
    none does -- profiles/r05_parity_relative.md.)"""
    dr, att, datt, gt, inp, proj, H, W, dev = _setup(pkg, name, B, S, ratio=ratio, seed=seed, no_mask=no_mask)
    rng = np.random.default_rng(seed + 77)
    w = rng.normal(size=(B, H, W, 4)).astype(np.float32)
    wfn = rng.normal(size=(B, dr.num_faces, 3)).astype(np.float32)
    rgbs, out = dr.render(no_mask=no_mask, **datt)
    ((rgbs.permute(0, 2, 3, 1) * torch.from_numpy(w).to(dev)).sum() + (out["face_normals"] * torch.from_numpy(wfn).to(dev)).sum()).backward()
    torch.cuda.synchronize()
    rgba_o, fidx_o, fn_o, imn_o = oracle.render_forward(inp, H, W, no_mask, proj)
    assert (dr.last_face_idx.cpu().numpy() == fidx_o).all()
    g_o = oracle.render_backward(inp, H, W, no_mask, proj, w, wfn)
    g64 = {}

    def ref64(k):
        if not g64:
            g64.update(oracle.render_backward(inp, H, W, no_mask, proj, w.astype(np.float64), wfn.astype(np.float64), dtype=np.float64))
        return g64[k]
    verdicts = {}
    for k in LEAVES:
        if k == "bg" and not no_mask:
            assert datt[k].grad is None
            continue
        assert float(np.abs(g_o[k]).max()) > 0.5, (k, "the upstream gradient was meant to make every input gradient O(1) or larger")
        verdicts[k] = grad_close(datt[k].grad, g_o[k], rtol=1e-4, what="%s, %s" % (label, k), ref64=lambda k=k: ref64(k))
    assert all(v == "ok" for v in verdicts.values()), verdicts      # (a 'cond' here would be news: say so instead of passing silently)


def test_the_gradient_bar_rejects_zero_scaled_and_partly_missing_gradients(pkg, oracle):
    """Mutation check of the bar itself at BASELINE config 2 (the bench's batch), for every one of the eight inputs: fed zeros, the reference
    scaled by 1 + 1e-3, or the true HIP gradient with ONE image's share removed, the comparison must FAIL -- and it must pass the true HIP
    gradient.  The bar of rounds 1-4 (1e-4 * max(1, max|ref|)) is evaluated beside it: it accepts the zero gradient for four of the eight
    inputs of this batch, which is why it was replaced."""
    dr, att, datt, gt, inp, proj, H, W, dev = _setup(pkg, "smpl_uv_642", 48, 128, seed=0)
    rgbs, out = dr.render(no_mask=True, **datt)
    dr.recon_data(rgbs, gt.to(dev), no_mask=True).backward()
    torch.cuda.synchronize()
    rgba_o, fidx_o, _, _ = oracle.render_forward(inp, H, W, True, proj)
    _, dpred = oracle.recon_data(rgba_o.transpose(0, 3, 1, 2), gt.numpy(), image_weight=dr.image_weight, want_grad=True)
    g_o = oracle.render_backward(inp, H, W, True, proj, np.ascontiguousarray(dpred.transpose(0, 2, 3, 1)), None)
    old_bar_accepts_zero = {}
    for k in LEAVES:
        ref = g_o[k]; got = datt[k].grad.cpu().numpy()
        assert grad_close(got, ref, what=k) == "ok"
        with pytest.raises(AssertionError):
            grad_close(np.zeros_like(ref), ref, what=k + " (zeros)")
        with pytest.raises(AssertionError):
            grad_close(ref * np.float32(1.001), ref, what=k + " (scaled)")
        worst = int(np.abs(ref.reshape(48, -1)).max(1).argmax())
        part = got.copy(); part[worst] = 0
        with pytest.raises(AssertionError):
            grad_close(part, ref, what=k + " (one image missing)")
        old_bar_accepts_zero[k] = float(np.abs(ref).max()) <= 1e-4 * max(1.0, float(np.abs(ref).max()))
    # the whole texture / background / light / azimuth gradient of this batch-mean loss is smaller than 1e-4: the old bar could not fail for them
    assert all(old_bar_accepts_zero[k] for k in ("textures", "bg", "lights", "azimuths")), old_bar_accepts_zero


@pytest.mark.parametrize("name,B,S,ratio,no_mask,seed,dist", [
    ("sphere", 4, 64, 1, True, 0, None),
    ("smpl_uv_642", 3, 32, 2, False, 2, None),      # H = 2W, white background
    ("sphere", 2, 50, 1, True, 4, None),            # ragged
    ("smpl_uv_642", 48, 128, 1, True, 0, None),     # BASELINE config 2, full size
    ("smpl_uv_642", 6, 128, 1, True, 3, 8.0),       # far camera: the whole mesh in a handful of tiles -> cooperative heavy-tile walk
    ("ellipsoid", 5, 200, 1, True, 9, None),
    ("smpl_uv", 2, 512, 1, True, 12, None),         # BASELINE config 5's shape (13 776 faces, 32-pixel bins, faces of several sweep chunks): the hint's four-lane sweep there
])
def test_the_two_walk_kernel_shapes_agree_bit_for_bit(pkg, name, B, S, ratio, no_mask, seed, dist):
    """MM_OPT_WALK_BLOCK (four tiles per 256-thread workgroup, heavy tiles walked by the four waves together) and MM_OPT_WALK_WAVE (one
    tile per one-wave workgroup) evaluate the same expressions and combine them with exact, commutative LDS atomics: every forward
    output must be identical, and so must the backward they feed.  So must MM_OPT_MANY_IN_FLIGHT's selection (below)."""
    N = pkg._native
    res = {}
    for tag, opt in (("block", N.OPT_WALK_BLOCK), ("wave", N.OPT_WALK_WAVE), ("hint", N.OPT_MANY_IN_FLIGHT)):
        dr, att, datt, gt, inp, proj, H, W, dev = _setup(pkg, name, B, S, ratio=ratio, seed=seed, no_mask=no_mask)
        if dist is not None:
            with torch.no_grad():
                datt["distances"].fill_(dist)
        dr.options = opt
        rgbs, out = dr.render(no_mask=no_mask, **datt)
        dr.recon_data(rgbs, gt.to(dev), no_mask=no_mask).backward()
        res[tag] = (rgbs.detach().clone(), dr.last_face_idx.clone(), out["face_normals"].detach().clone(), out["imnormal"].clone(),
                    {k: datt[k].grad.clone() for k in LEAVES if datt[k].grad is not None})
    a = res["block"]
    assert float((a[1] >= 0).float().mean()) > 0.01
    # "hint" = MM_OPT_MANY_IN_FLIGHT (round 6): the caller's word that several calls share the chip -- the large-batch shapes of the forward walk AND of
    # the backward's face sweep (four lanes per item) at any batch size; a hint, so nothing may change
    for other in ("wave", "hint"):
        b = res[other]
        assert torch.equal(a[1], b[1]) and torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]), other
        for k in a[4]:
            if S >= 512 and k == "vertices":
                # Screen bins larger than a tile (the compacting walk): the cooperative heavy-tile walk of the 256-thread shape flags a SUPERSET of the
                # faces some pixel took into its silhouette product (mark_taken, mm_raster_walk.h), so a few faces are swept over their inflated box
                # instead of their own: the same exact integer sums, cut into different sweep items, each rounded to float once -- measured 6e-10 of
                # the gradient's maximum at BASELINE config 5 (profiles/tools/shape_grad_diff.py).  Two one-wave forms (wave, hint) flag the same faces.
                assert float((a[4][k] - b[4][k]).abs().max()) <= 1e-8 * float(a[4][k].abs().max()), (other, k)
                assert torch.equal(res["wave"][4][k], b[4][k]), (other, k)
            else:
                assert torch.equal(a[4][k], b[4][k]), (other, k)     # integer fixed-point sums: the backward is bitwise reproducible


@pytest.mark.parametrize("knum,boxlen,sigmainv,dist", [
    (3, 0.02, 7000.0, None),      # knum far below the candidates per pixel: the "first knum faces in index order" rule
    (30, 0.02, 7000.0, 9.0),      # far camera: the whole mesh in a few tiles, > 30 candidates per silhouette pixel
    (7, 0.08, 900.0, None),       # wide soft margin, flat falloff: many more faces per pixel, tiles with > 64 candidates
    (80, 0.3, 60.0, None),        # knum > 64 (more than one staged batch per pixel) with a margin wide enough to reach it
])
def test_soft_mask_truncation_and_margins_match_oracle(pkg, oracle, knum, boxlen, sigmainv, dist):
    """dibr_rasterization's knum / boxlen / sigmainv away from their defaults: forward and backward."""
    dr, att, datt, gt, inp, proj, H, W, dev = _setup(pkg, "sphere", 3, 64, seed=11)
    dr.knum, dr.boxlen, dr.sigmainv = knum, boxlen, sigmainv
    if dist is not None:
        with torch.no_grad():
            datt["distances"].fill_(dist)
        inp["distances"] = np.full_like(inp["distances"], dist)
    rgbs, out = dr.render(no_mask=True, **datt)
    dr.recon_data(rgbs, gt.to(dev), no_mask=True).backward()
    kw = dict(knum=knum, boxlen=boxlen, sigmainv=sigmainv)
    rgba_o, fidx_o, fn_o, imn_o = oracle.render_forward(inp, H, W, True, proj, **kw)
    loss_o, dpred = oracle.recon_data(rgba_o.transpose(0, 3, 1, 2), gt.numpy(), image_weight=dr.image_weight, want_grad=True)
    g_o = oracle.render_backward(inp, H, W, True, proj, np.ascontiguousarray(dpred.transpose(0, 2, 3, 1)), None, **kw)
    assert (dr.last_face_idx.cpu().numpy() == fidx_o).all()
    alpha = rgba_o[..., 3]
    assert ((alpha > 0.01) & (alpha < 0.99)).mean() > 0.005          # there is a silhouette band to get wrong
    _close(rgbs.detach().permute(0, 2, 3, 1).cpu().numpy(), rgba_o)
    for k in LEAVES:
        _gclose(datt[k].grad.cpu().numpy(), g_o[k], what=k)


@pytest.mark.parametrize("name,B,S,dist", [("sphere", 6, 128, 1.45), ("smpl_uv_642", 4, 200, 1.6)])
def test_close_camera_huge_face_boxes_match_oracle(pkg, oracle, name, B, S, dist):
    """Camera almost inside the mesh: perspective blows single faces up to thousands of box pixels.  The backward sweeps such faces
    in chunks on dedicated waves (plan kernel -> chunk list -> partial sums added by the vertex backward); same bar as everywhere."""
    dr, att, datt, gt, inp, proj, H, W, dev = _setup(pkg, name, B, S, seed=31)
    with torch.no_grad():
        datt["distances"].fill_(dist)
    inp["distances"] = np.full_like(inp["distances"], dist)
    rgbs, out = dr.render(no_mask=True, **datt)
    dr.recon_data(rgbs, gt.to(dev), no_mask=True).backward()
    rgba_o, fidx_o, fn_o, imn_o = oracle.render_forward(inp, H, W, True, proj)
    loss_o, dpred = oracle.recon_data(rgba_o.transpose(0, 3, 1, 2), gt.numpy(), image_weight=dr.image_weight, want_grad=True)
    g_o = oracle.render_backward(inp, H, W, True, proj, np.ascontiguousarray(dpred.transpose(0, 2, 3, 1)), None)
    assert (dr.last_face_idx.cpu().numpy() == fidx_o).all()
    # faces with more than 512 box pixels exist (what the chunk path is for)
    fvi = oracle.prepare_vertices(inp["vertices"], inp["faces"], oracle.camera(inp["distances"], inp["elevations"], inp["azimuths"], inp["biases"]), proj)[1]
    ext = (fvi.max(2) - fvi.min(2)) * np.asarray([W, H], np.float32) / 2.0
    assert int(((ext[..., 0] * ext[..., 1]) > 512).sum()) > 10
    _close(rgbs.detach().permute(0, 2, 3, 1).cpu().numpy(), rgba_o)
    for k in LEAVES:
        _gclose(datt[k].grad.cpu().numpy(), g_o[k], what=k)


def test_texture_record_pool_overflow_is_loud_and_a_larger_workspace_holds_it(pkg, oracle):
    """Every covered pixel's bilinear footprint on the corner shared by four 32x32-texel tiles: four texture-gradient records per pixel,
    more than the 9/8 per pixel the minimum workspace's record array holds.  The images that run out get NaN texture gradients (never a
    short sum) and mm_render_status reports the dropped records; every other gradient is unaffected; with the array enlarged through
    workspace_bytes the same batch matches the oracle."""
    def run(extra, check):
        dr, att, datt, gt, inp, proj, H, W, dev = _setup(pkg, "sphere", 3, 64, seed=12)
        Ht, Wt = att["textures"].shape[2:]
        dr.face_uvs = torch.empty_like(dr.face_uvs)
        dr.face_uvs[..., 0] = 32.0 / Wt                           # texel coordinates (31.5, 31.5)
        dr.face_uvs[..., 1] = 1.0 - 32.0 / Ht
        inp["face_uvs"] = dr.face_uvs.numpy()[0]
        dr.extra_texture_records_per_pixel, dr.check_texture_records = extra, check
        with torch.no_grad():
            datt["distances"].fill_(1.9)
        inp["distances"] = np.full_like(inp["distances"], 1.9)
        rgbs, out = dr.render(no_mask=True, **datt)
        loss = dr.recon_data(rgbs, gt.to(dev), no_mask=True)
        return dr, datt, gt, inp, proj, H, W, loss

    dr, datt, gt, inp, proj, H, W, loss = run(0.0, False)
    loss.backward()
    torch.cuda.synchronize()
    assert (dr.last_face_idx >= 0).float().mean() > 0.4            # 4 x 0.4 HW records against room for 9/8 HW
    rgba_o, fidx_o, fn_o, imn_o = oracle.render_forward(inp, H, W, True, proj)
    loss_o, dpred = oracle.recon_data(rgba_o.transpose(0, 3, 1, 2), gt.numpy(), image_weight=dr.image_weight, want_grad=True)
    g_o = oracle.render_backward(inp, H, W, True, proj, np.ascontiguousarray(dpred.transpose(0, 2, 3, 1)), None)
    assert torch.isnan(datt["textures"].grad).all()                # loud, in every texel of every image that lost records
    for k in LEAVES:
        if k != "textures":
            _gclose(datt[k].grad.cpu().numpy(), g_o[k], what=k)
    # ... and loud WITHOUT a diagnostic switch or a synchronisation of the caller's: the backward added the dropped records to the object's pinned
    # status word (MMRenderDesc.status_flag), and the next render of this object -- any node flavour -- raises
    assert dr.poll_dropped_records(reset=False) > 0
    with pytest.raises(RuntimeError, match="dropped"):
        dr.render(no_mask=True, **{k: (v.detach() if torch.is_tensor(v) else v) for k, v in datt.items()})
    assert dr.poll_dropped_records() == 0                          # (reported once)

    dr, datt, gt, inp, proj, H, W, loss = run(0.0, True)           # the class API's diagnostic switch raises instead
    with pytest.raises(RuntimeError, match="texture-record pool"):
        loss.backward()

    dr, datt, gt, inp, proj, H, W, loss = run(3.0, True)           # 4 1/8 records per pixel: enough for any image
    loss.backward()
    assert np.abs(g_o["textures"]).max() > 0
    for k in LEAVES:
        _gclose(datt[k].grad.cpu().numpy(), g_o[k], what=k)


@pytest.mark.parametrize("bit", ["OPT_CULL_STRICT", "OPT_SOFT_SKIP_CULLED", "OPT_BBOX_HALF_OPEN", "OPT_BBOX_MIN_CLOSED_MAX_OPEN", "OPT_BARY_ONE_MINUS", "OPT_SH_ORDER_XYZ",
                                 "ALL"])
def test_appendix_c_switches_match_the_oracle(pkg, oracle, bit):
    """the choices recalled from kaolin's sources (cull >= / >, soft mask over culled faces, closed / [min, max) / open
    bbox, copysign(eps) / (1 - w1 - w2) barycentrics, SH band order) are switches of MMRenderDesc.options, mirrored bit for bit by
    the oracle: whoever can run real kaolin pins the path by flipping a bit.  Every switch alone and all together:
      (1) forward + backward at the full bar on the usual seeded input;
      (2) forward (face_idx bit-exact, RGBA 1e-4) on an input built to make the switch BITE -- vertices snapped to a coarse grid and
          an axis-aligned camera put pixel centres exactly on edges and box borders and give exactly edge-on faces (normal z == 0):
          gradients are not compared there (1 / area of a zero-area face), the switches only touch the forward's decisions."""
    N = pkg._native
    names = ["OPT_CULL_STRICT", "OPT_SOFT_SKIP_CULLED", "OPT_BBOX_HALF_OPEN", "OPT_BBOX_MIN_CLOSED_MAX_OPEN", "OPT_BARY_ONE_MINUS", "OPT_SH_ORDER_XYZ"]
    bits = sum(getattr(N, n) for n in names) if bit == "ALL" else getattr(N, bit)
    assert bits == (sum(getattr(oracle, n) for n in names) if bit == "ALL" else getattr(oracle, bit))
    # (1)
    dr, att, datt, gt, inp, proj, H, W, dev = _setup(pkg, "sphere", 4, 64, seed=17)
    with torch.no_grad():
        base_rgbs, _ = dr.render(no_mask=True, **datt)
    dr.options = bits
    rgbs, out = dr.render(no_mask=True, **datt)
    dr.recon_data(rgbs, gt.to(dev), no_mask=True).backward()
    with oracle.options(bits):
        rgba_o, fidx_o, fn_o, imn_o = oracle.render_forward(inp, H, W, True, proj)
        loss_o, dpred = oracle.recon_data(rgba_o.transpose(0, 3, 1, 2), gt.numpy(), image_weight=dr.image_weight, want_grad=True)
        g_o = oracle.render_backward(inp, H, W, True, proj, np.ascontiguousarray(dpred.transpose(0, 2, 3, 1)), None)
    assert (dr.last_face_idx.cpu().numpy() == fidx_o).all()
    _close(rgbs.detach().permute(0, 2, 3, 1).cpu().numpy(), rgba_o)
    for k in LEAVES:
        _gclose(datt[k].grad.cpu().numpy(), g_o[k], what=k)
    if bit in ("OPT_SOFT_SKIP_CULLED", "OPT_SH_ORDER_XYZ", "ALL"):             # these change every silhouette pixel / every lit pixel
        assert float((base_rgbs - rgbs.detach()).abs().max()) > 1e-3
    # (2)
    dr2, att2, datt2, gt2, inp2, proj2, H, W, dev = _setup(pkg, "sphere", 4, 64, seed=17)
    with torch.no_grad():
        datt2["vertices"].copy_(torch.round(datt2["vertices"] * 16) / 16)
        datt2["azimuths"].fill_(0.0); datt2["elevations"].fill_(0.0); datt2["biases"].zero_(); datt2["distances"].fill_(2.5)
        for k in ("vertices", "azimuths", "elevations", "biases", "distances"):
            inp2[k] = datt2[k].detach().cpu().numpy().copy()
        base2, _ = dr2.render(no_mask=True, **datt2)
        base_idx = dr2.last_face_idx.clone()
        dr2.options = bits
        r2, _ = dr2.render(no_mask=True, **datt2)
    with oracle.options(bits):
        rgba2_o, fidx2_o, _, _ = oracle.render_forward(inp2, H, W, True, proj2)
    got_idx = dr2.last_face_idx.cpu().numpy()
    assert (got_idx == fidx2_o).all(), int((got_idx != fidx2_o).sum())
    _close(r2.permute(0, 2, 3, 1).cpu().numpy(), rgba2_o)
    assert (not torch.equal(base_idx, dr2.last_face_idx)) or float((base2 - r2).abs().max()) > 1e-6, bit   # the switch changed something


PIN_COMBOS = [("OPT_BARY_ONE_MINUS", "OPT_BBOX_MIN_CLOSED_MAX_OPEN"), ("OPT_SOFT_SKIP_CULLED",), ("OPT_SH_ORDER_XYZ",),
              ("OPT_BARY_ONE_MINUS", "OPT_BBOX_MIN_CLOSED_MAX_OPEN", "OPT_SOFT_SKIP_CULLED", "OPT_SH_ORDER_XYZ")]


@pytest.mark.parametrize("names", PIN_COMBOS, ids=["+".join(n[4:] for n in c) for c in PIN_COMBOS])
@pytest.mark.parametrize("shape", ["config2", "config5_one_image"])
def test_the_option_combinations_a_kaolin_fixture_may_select_hold_the_bar_at_baseline_sizes(pkg, oracle, names, shape):
    """Ready for the pin (verdict r05 item 5).  The two defaults the judge's reading of kaolin 0.12 disputes -- barycentrics as w1, w2 over
    (sum + eps) with w0 = 1 - w1 - w2, and a [min, max) box test -- and the two bits whose effect is not small (the soft mask skipping culled
    faces, the SH band order), alone and all together: the full bar (face_idx bit-exact, RGBA 1e-4, every gradient 1e-4 of its own maximum)
    at BASELINE config 2 at full size and on one image of config 5 (13 776 faces, 512x512).  Whichever combination a kaolin fixture selects
    (tests/test_kaolin_pinning.py) is already tested at these sizes; bench.py times the same combinations (`options_ab` in its line)."""
    N = pkg._native
    bits = sum(getattr(N, n) for n in names)
    assert bits == sum(getattr(oracle, n) for n in names)
    if shape == "config2":
        dr, att, datt, gt, inp, proj, H, W, dev = _setup(pkg, "smpl_uv_642", 48, 128, seed=0)
    else:
        dr, att, datt, gt, inp, proj, H, W, dev = _setup(pkg, "smpl_uv", 1, 512, seed=12, imn=False)
    dr.options = bits
    rgbs, out = dr.render(no_mask=True, **datt)
    loss = dr.recon_data(rgbs, gt.to(dev), no_mask=True)
    loss.backward()
    with oracle.options(bits):
        rgba_o, fidx_o, _, _ = oracle.render_forward(inp, H, W, True, proj)
        loss_o, g_o = oracle.step(inp, gt.numpy(), H, W, True, proj, image_weight=dr.image_weight)
    got_idx = dr.last_face_idx.cpu().numpy()
    assert np.array_equal(got_idx, fidx_o), int((got_idx != fidx_o).sum())
    _close(rgbs.detach().permute(0, 2, 3, 1).cpu().numpy(), rgba_o)
    assert abs(float(loss) - loss_o) < 1e-5
    for k in LEAVES:
        _gclose(datt[k].grad.cpu().numpy(), g_o[k], what=k)


@pytest.mark.parametrize("name,B,S,ratio,no_mask,sigmainv,boxlen,seed,dist", [
    ("sphere", 2, 96, 2, True, 900.0, 0.02, 794003348, None),           # a covered pixel whose texture row coordinate is 58.99994
    ("smpl_uv_642", 6, 96, 1, True, 900.0, 0.02, 339905349, 1.808205592613872),
    ("sphere", 3, 128, 2, False, 200.0, 0.02, 723017325, None),
    ("ellipsoid", 3, 128, 2, False, 200.0, 0.15, 769837633, None),
])
def test_fuzz_regressions_pixel_pass_recomputes_the_forward_bit_for_bit(pkg, oracle, name, B, S, ratio, no_mask, sigmainv, boxlen, seed, dist):
    """Found by profiles/tools/fuzz_parity.py (4 of 860 random cases): the backward's pixel pass recomputes uv -> texel cell and the
    pre-clamp colour; compiled with other floating-point flags than the forward it landed, for a pixel within an ulp of a texel-cell border
    (or of 0 / 1), in the neighbouring bilinear cell (on the other side of torch.clamp): a different one-sided derivative, up to 1e-3 on one
    vertex gradient.  The pixel pass is now compiled like the forward (csrc/mm_backward.h) and recomputes it bit for bit."""
    dr = pkg.DiffRender(os.path.join(TEMPLATES, name + ".npz"), S, ratio=ratio)
    dr.sigmainv, dr.boxlen = sigmainv, boxlen
    dev = torch.device("cuda:0")
    H, W = dr.render_height, dr.image_size
    att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, B, H, W, seed=seed)
    if dist is not None:
        att["distances"] = torch.full_like(att["distances"], dist)
    datt = {k: (v.to(dev).requires_grad_(k in LEAVES) if torch.is_tensor(v) else v) for k, v in att.items()}
    inp = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in att.items()}
    inp["faces"] = dr.faces.numpy().astype(np.int32); inp["face_uvs"] = dr.face_uvs.numpy()[0]
    proj = dr.cam_proj.numpy().reshape(3)
    rgbs, out = dr.render(no_mask=no_mask, **datt)
    dr.recon_data(rgbs, gt.to(dev), no_mask=no_mask).backward()
    kw = dict(sigmainv=sigmainv, boxlen=boxlen)
    rgba_o, fidx_o, fn_o, imn_o = oracle.render_forward(inp, H, W, no_mask, proj, **kw)
    loss_o, dpred = oracle.recon_data(rgba_o.transpose(0, 3, 1, 2), gt.numpy(), image_weight=dr.image_weight, want_grad=True)
    g_o = oracle.render_backward(inp, H, W, no_mask, proj, np.ascontiguousarray(dpred.transpose(0, 2, 3, 1)), None, **kw)
    assert (dr.last_face_idx.cpu().numpy() == fidx_o).all()
    assert np.array_equal(rgbs.detach().permute(0, 2, 3, 1).cpu().numpy()[..., :3], rgba_o[..., :3])       # the colour channels are bit-exact
    for k in LEAVES:
        if k == "bg" and not no_mask:
            continue
        _gclose(datt[k].grad.cpu().numpy(), g_o[k], what=k)


def test_fuzz_regression_an_image_that_shows_nothing_still_has_its_tiny_geometry_gradient(pkg, oracle):
    """Found by profiles/tools/fuzz_parity.py under round 5's scale-aware bar (case 82 of seed 6106): an 8x8 screen, the camera 25 units away, white
    background -- no pixel is covered and every silhouette factor 1 - exp(-sigma d^2) rounds to exactly 1, so the image is alpha = 0 to the last bit.  The
    derivative terms of those factors are 1e-9 small but they are the WHOLE geometry gradient of this batch; the backward used to skip pixels whose stored
    product is exactly 1 and returned exact zeros (invisible to the old absolute bar).  Same bar as everywhere: 1e-4 of each gradient's own maximum."""
    dr = pkg.DiffRender(os.path.join(TEMPLATES, "ellipsoid.npz"), 8)
    dr.boxlen, dr.sigmainv = 0.05, 7000.0
    N = pkg._native
    dr.options = N.OPT_BARY_ONE_MINUS | N.OPT_BBOX_MIN_CLOSED_MAX_OPEN | N.OPT_WALK_BATCH        # (the case's option bits: 640 | 2048)
    assert dr.options == 640 | 2048
    dev = torch.device("cuda:0")
    B, H, W = 8, 8, 8
    att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, B, H, W, seed=295243571)
    att["distances"] = torch.full_like(att["distances"], 25.444954239929586)
    datt = {k: (v.to(dev).requires_grad_(k in LEAVES) if torch.is_tensor(v) else v) for k, v in att.items()}
    inp = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in att.items()}
    inp["faces"] = dr.faces.numpy().astype(np.int32); inp["face_uvs"] = dr.face_uvs.numpy()[0]
    proj = dr.cam_proj.numpy().reshape(3)
    loss, rgbs, out = dr.render_recon(gt.to(dev), no_mask=False, **datt)
    loss.backward()
    kw = dict(boxlen=0.05, sigmainv=7000.0)
    with oracle.options(640):
        rgba_o, fidx_o, _, _ = oracle.render_forward(inp, H, W, False, proj, **kw)
        _, dpred = oracle.recon_data(rgba_o.transpose(0, 3, 1, 2), gt.numpy(), image_weight=dr.image_weight, want_grad=True)
        g_o = oracle.render_backward(inp, H, W, False, proj, np.ascontiguousarray(dpred.transpose(0, 2, 3, 1)), None, **kw)
    assert (fidx_o == -1).all() and float(rgba_o[..., 3].max()) == 0.0                         # the image shows nothing at all
    assert (dr.last_face_idx.cpu().numpy() == fidx_o).all()
    assert np.array_equal(rgbs.detach().permute(0, 2, 3, 1).cpu().numpy(), rgba_o)
    for k in ("vertices", "azimuths", "elevations", "distances", "biases"):
        assert 0 < float(np.abs(g_o[k]).max()) < 1e-8, k                                       # tiny, and not zero
        _gclose(datt[k].grad.cpu().numpy(), g_o[k], what=k)
    for k in ("textures", "lights"):
        assert float(np.abs(g_o[k]).max()) == 0.0 and float(datt[k].grad.abs().max()) == 0.0


def test_fuzz_regression_saturated_silhouettes_keep_their_tiny_geometry_gradient_to_the_full_bar(pkg, oracle):
    """Found by profiles/tools/fuzz_parity.py in round 6 (case 254 of seed 8809): 13 776 faces on a 16x16 screen 26.6 units away, sigmainv = 200 -- every
    silhouette pixel is saturated and none is covered: the geometry gradients are 3e-11 (vertices) ... 1e-13 (azimuths) under an upstream gradient of 1.7e-2.  Rounds 2-5 kept the
    gather's per-item fixed-point sums at a unit of 2^-40 of the image's K4 bound whatever the item's size: these gradients came out 1e-3 ... 4e-3 of their own
    maximum off (and exact zeros below 2^-41 of the bound: round 5's NEGL class).  The unit now follows the chunk size (2^-55 of the bound for a 128-pixel
    chunk, csrc/mm_backward.hip: face_sum_scale): the full bar holds for every one of them."""
    dr = pkg.DiffRender(os.path.join(TEMPLATES, "smpl_uv.npz"), 16)
    dr.knum, dr.boxlen, dr.sigmainv = 30, 0.05, 200.0
    dr.options = 128
    dev = torch.device("cuda:0")
    B, H, W = 3, 16, 16
    att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, B, H, W, seed=906465233)
    att["distances"] = torch.full_like(att["distances"], 26.642802256650747)
    datt = {k: (v.to(dev).requires_grad_(k in LEAVES) if torch.is_tensor(v) else v) for k, v in att.items()}
    inp = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in att.items()}
    inp["faces"] = dr.faces.numpy().astype(np.int32); inp["face_uvs"] = dr.face_uvs.numpy()[0]
    proj = dr.cam_proj.numpy().reshape(3)
    loss, rgbs, out = dr.render_recon(gt.to(dev), no_mask=False, contour=0.5, **datt)
    loss.backward()
    kw = dict(knum=30, boxlen=0.05, sigmainv=200.0)
    with oracle.options(128):
        rgba_o, fidx_o, _, _ = oracle.render_forward(inp, H, W, False, proj, **kw)
        _, dpred = oracle.recon_data(rgba_o.transpose(0, 3, 1, 2), gt.numpy(), image_weight=dr.image_weight, contour=0.5, want_grad=True)
        g_o = oracle.render_backward(inp, H, W, False, proj, np.ascontiguousarray(dpred.transpose(0, 2, 3, 1)), None, **kw)
    assert (dr.last_face_idx.cpu().numpy() == fidx_o).all()
    _close(rgbs.detach().permute(0, 2, 3, 1).cpu().numpy(), rgba_o)
    assert (fidx_o == -1).all() and float(rgba_o[..., 3].max()) == 1.0                         # nothing covered, saturated silhouette pixels
    up = float(np.abs(dpred).max())                                                             # the upstream gradient dL/d(pixel): 1.7e-2
    for k in ("vertices", "azimuths", "elevations", "distances", "biases"):
        assert 0 < float(np.abs(g_o[k]).max()) < 1e-7 * up, k                                   # tiny beside what drives them, and not zero
        _gclose(datt[k].grad.cpu().numpy(), g_o[k], what=k)
    for k in ("textures", "lights"):
        assert float(np.abs(g_o[k]).max()) == 0.0 and float(datt[k].grad.abs().max()) == 0.0


def test_backward_twice_after_one_forward(pkg):
    """retain_graph: the backward leaves its scratch counters the way it found them (the library clears them in-kernel)."""
    dr, att, datt, gt, inp, proj, H, W, dev = _setup(pkg, "smpl_uv_642", 4, 96, seed=21)
    rgbs, _ = dr.render(no_mask=True, **datt)
    loss = dr.recon_data(rgbs, gt.to(dev), no_mask=True)
    loss.backward(retain_graph=True)
    first = {k: datt[k].grad.clone() for k in LEAVES}
    loss.backward()
    for k in LEAVES:
        _gclose(datt[k].grad.cpu().numpy(), 2.0 * first[k].cpu().numpy(), 1e-6)
        assert float(first[k].abs().max()) > 0


def test_backward_is_bitwise_reproducible(pkg):
    """No floating-point atomic anywhere in the backward (integer fixed point in LDS, fixed-order sums): the same step twice gives the
    same bits for all eight gradients, at the headline size."""
    grads = []
    for _ in range(2):
        dr, att, datt, gt, inp, proj, H, W, dev = _setup(pkg, "smpl_uv_642", 48, 128, seed=0, imn=False)
        rgbs, _ = dr.render(no_mask=True, **datt)
        dr.recon_data(rgbs, gt.to(dev), no_mask=True).backward()
        grads.append({k: datt[k].grad.clone() for k in LEAVES})
    for k in LEAVES:
        assert torch.equal(grads[0][k], grads[1][k]), k
        assert float(grads[0][k].abs().max()) > 0


def test_render_recon_is_render_plus_recon_data(pkg):
    """DiffRender.render_recon = render + recon_data with the loss folded into the render kernels: same value, same gradients."""
    got = []
    for fused in (False, True):
        dr, att, datt, gt, inp, proj, H, W, dev = _setup(pkg, "smpl_uv_642", 6, 96, seed=31)
        if fused:
            loss, rgbs, out = dr.render_recon(gt.to(dev), no_mask=True, **datt)
            assert not rgbs.requires_grad
        else:
            rgbs, out = dr.render(no_mask=True, **datt)
            loss = dr.recon_data(rgbs, gt.to(dev), no_mask=True)
        (2.5 * loss).backward()
        got.append((float(loss), rgbs.detach().clone(), {k: datt[k].grad.clone() for k in LEAVES}))
    assert abs(got[0][0] - got[1][0]) < 2e-6 and torch.equal(got[0][1], got[1][1])
    for k in LEAVES:
        _gclose(got[1][2][k].cpu().numpy(), got[0][2][k].cpu().numpy(), 2e-5)
        assert float(got[1][2][k].abs().max()) > 0


def test_captured_step_replays_the_eager_step_bit_for_bit(pkg):
    """RenderLossStep.capture(): the whole C-ABI step (mm_render_forward + mm_render_backward, fused loss) captured as ONE HIP graph -- the library
    neither allocates nor synchronises, so the capture is plain (bench.py --mode hipgraph).  Replays give the eager run's bits: loss, image,
    face_idx and all eight gradients; the input slots may be refilled between replays."""
    import importlib
    stepmod = importlib.import_module("3d-magic-mirror_amd.step")
    dr, att, datt, gt, inp, proj, H, W, dev = _setup(pkg, "smpl_uv_642", 6, 96, seed=41)
    plain = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in datt.items()}
    eager = stepmod.RenderLossStep(dr, plain, gt.to(dev), no_mask=True, fused=True)
    eager.run(); torch.cuda.synchronize()
    ref = (float(eager.loss), eager.rgba.clone(), eager.face_idx.clone(), {k: v.clone() for k, v in eager.grads.items() if v is not None})
    cap = stepmod.RenderLossStep(dr, plain, gt.to(dev), no_mask=True, fused=True)
    cap.capture()
    for rep in range(3):
        cap.replay(); torch.cuda.synchronize()
        assert float(cap.loss) == ref[0] and torch.equal(cap.rgba, ref[1]) and torch.equal(cap.face_idx, ref[2])
        for k, g in ref[3].items():
            assert torch.equal(cap.grads[k], g), k
    # other inputs through the same slots: the graph reads the slots' memory
    _, _, d2, gt2, _, _, _, _, _ = _setup(pkg, "smpl_uv_642", 6, 96, seed=42)
    with torch.no_grad():
        for k in LEAVES:
            cap.inp[k].copy_(d2[k].detach().reshape(cap.inp[k].shape))
        cap.gt.copy_(gt2.to(dev))
    cap.replay(); torch.cuda.synchronize()
    e2 = stepmod.RenderLossStep(dr, {k: (v.detach() if torch.is_tensor(v) else v) for k, v in d2.items()}, gt2.to(dev), no_mask=True, fused=True)
    e2.run(); torch.cuda.synchronize()
    assert float(cap.loss) == float(e2.loss) and float(cap.loss) != ref[0]
    for k in LEAVES:
        assert torch.equal(cap.grads[k], e2.grads[k]), k


def test_fused_backward_does_not_read_the_forward_image_back(pkg):
    """include/mm_render.h, fused_gt: the fused backward re-forms the prediction per pixel (bit for bit) instead of reading `rgba` back,
    so a caller may overwrite the image between forward and backward: same bits for all eight gradients either way."""
    grads = []
    for scribble in (False, True):
        dr, att, datt, gt, inp, proj, H, W, dev = _setup(pkg, "smpl_uv_642", 5, 96, seed=12)
        loss, rgbs, out = dr.render_recon(gt.to(dev), no_mask=True, **datt)
        if scribble:
            rgbs.detach().uniform_(-3.0, 3.0)                    # (a view of the storage the forward wrote)
        loss.backward()
        grads.append({k: datt[k].grad.clone() for k in LEAVES})
    for k in LEAVES:
        assert torch.equal(grads[0][k], grads[1][k]), k
        assert float(grads[0][k].abs().max()) > 0


def test_large_batches_take_the_per_image_vertex_backward_and_agree_with_small_ones(pkg):
    """B >= 128 (small templates): the vertex backward runs as one workgroup per image (vertex_image_bwd_kernel: face-major -> LDS -> vertex-major,
    no ticket); smaller batches run the 8-lanes-per-vertex grid.  An image's gradients do not depend on the batch it is rendered in: one call of
    136 images against the same images in two calls of 68, every input gradient to 2e-6 of its largest entry; and the large call is bitwise
    reproducible."""
    dr, att, datt, gt, inp, proj, H, W, dev = _setup(pkg, "sphere", 136, 32, seed=91)
    w = torch.linspace(-1.0, 1.0, 136 * 4 * H * W, device=dev).reshape(136, 4, H, W) * 1e-2
    wf = torch.linspace(1.0, -1.0, 136 * dr.num_faces * 3, device=dev).reshape(136, dr.num_faces, 3) * 1e-3

    def grads(sl):
        lv = {k: datt[k].detach()[sl].clone().requires_grad_(True) for k in LEAVES}
        a = dict(datt); a.update(lv)
        rgbs, out = dr.render(no_mask=True, **a)
        ((rgbs * w[sl]).sum() + (out["face_normals"] * wf[sl]).sum()).backward()
        torch.cuda.synchronize()
        return {k: lv[k].grad.clone() for k in LEAVES}
    big = grads(slice(0, 136))
    again = grads(slice(0, 136))
    halves = [grads(slice(0, 68)), grads(slice(68, 136))]
    for k in LEAVES:
        assert torch.equal(big[k], again[k]), k
        ref = torch.cat([halves[0][k], halves[1][k]], 0)
        assert float(big[k].abs().max()) > 0
        assert float((big[k] - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max())), (k, float((big[k] - ref).abs().max()))


def test_geometry_only_render_is_the_render_without_the_image(pkg):
    """DiffRender.render_geometry (MMRenderDesc.geometry_only), for the call site that discards the image (trainer.py:367): face_normals
    are render's bit for bit, and the gradient a loss on them sends to vertices and camera is what the full render's backward gives when
    only face_normals is differentiated; twice after one forward (retain_graph) as well."""
    dr, att, datt, gt, inp, proj, H, W, dev = _setup(pkg, "smpl_uv_642", 5, 64, seed=81)
    w = torch.linspace(-1.0, 1.0, 5 * dr.num_faces * 3, device=dev).reshape(5, dr.num_faces, 3)
    _, out = dr.render(no_mask=True, **datt)
    (out["face_normals"] * w).sum().backward()
    ref_fn = out["face_normals"].detach().clone()
    ref = {k: datt[k].grad.clone() for k in ("vertices", "azimuths", "elevations", "distances", "biases")}
    lv = {k: datt[k].detach().clone().requires_grad_(True) for k in LEAVES}
    a = dict(datt); a.update(lv)
    out_g = dr.render_geometry(**a)
    assert out_g["imnormal"] is None and torch.equal(out_g["face_normals"].detach(), ref_fn)
    tot = (out_g["face_normals"] * w).sum()
    tot.backward(retain_graph=True)
    for k, r in ref.items():
        _gclose(lv[k].grad.cpu().numpy(), r.cpu().numpy(), 1e-6)
    assert lv["textures"].grad is None and lv["lights"].grad is None and lv["bg"].grad is None
    first = {k: lv[k].grad.clone() for k in ref}
    tot.backward()                                               # a second backward on the same workspace: the arrival counter was left clean
    for k in ref:
        assert torch.equal(lv[k].grad, first[k] + first[k]), k


def test_lane_exchange_primitives_on_this_gpu():
    """mm_device.h builds its 64x64 bit transposes and prefix scans from DPP lane selects and gfx950's v_permlane16/32_swap (no LDS-crossbar
    shuffles); profiles/tools/xchg_test.hip checks every stride, the transpose and the scan against their definitions on the device."""
    import shutil, subprocess, tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    assert os.path.exists(hipcc), "this test compiles its checker on the GPU box (ROCm image: /opt/rocm/bin/hipcc); without a compiler it FAILS rather than skipping"
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "xchg_test")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "3d-magic-mirror_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "profiles", "tools", "xchg_test.hip"), "-o", exe], check=True, capture_output=True, timeout=300)
        out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "bad 0" in out.stdout, out.stdout + out.stderr


def test_cpp_and_python_host_paths_agree(pkg):
    """The optional C++ autograd nodes (lib/mm_torch_ext.so) and the Python torch.autograd.Functions issue the same ABI calls: same bits,
    for render + recon_data and for render_recon."""
    N = pkg._native
    if N.torch_ext() is None:
        pytest.skip("mm_torch_ext is not built")
    ext, got = N._EXT, []
    try:
        for use_ext in (True, False):
            N._EXT = ext if use_ext else None
            for fused in (False, True):
                dr, att, datt, gt, inp, proj, H, W, dev = _setup(pkg, "smpl_uv_642", 5, 80, seed=41)
                if fused:
                    loss, rgbs, out = dr.render_recon(gt.to(dev), no_mask=True, **datt)
                else:
                    rgbs, out = dr.render(no_mask=True, **datt)
                    loss = dr.recon_data(rgbs, gt.to(dev), no_mask=True) + 1e-3 * out["face_normals"].sum()
                loss.backward()
                got.append((loss.detach().clone(), rgbs.detach().clone(), out["imnormal"].clone(), dr.last_face_idx.clone(),
                            {k: datt[k].grad.clone() for k in LEAVES}))
    finally:
        N._EXT = ext
    for a, b in ((got[0], got[2]), (got[1], got[3])):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
        for k in LEAVES:
            assert torch.equal(a[4][k], b[4][k]), k


def test_recon_data_matches_reference_golden(pkg):
    z = np.load(os.path.join(GOLDEN, "losses.npz"))
    dev = torch.device("cuda:0")
    dr = pkg.DiffRender(os.path.join(TEMPLATES, "sphere.npz"), 64, image_weight=0.1)
    for contour in (0.0, 0.5):
        pred = torch.from_numpy(z["rd_pred_nhwc"]).to(dev).permute(0, 3, 1, 2).requires_grad_(True)
        loss = dr.recon_data(pred, torch.from_numpy(z["rd_gt"]).to(dev), no_mask=True, contour=contour)
        (loss * 3.0).backward()
        assert abs(float(loss) - float(z["recon_data_c%g" % contour])) < 2e-6
        got = pred.grad.permute(0, 2, 3, 1).cpu().numpy() / 3.0
        np.testing.assert_allclose(got, z["recon_data_c%g__d_pred_nhwc" % contour], rtol=1e-4, atol=1e-9)
    # NCHW-contiguous prediction goes through the strided path too
    pred = torch.from_numpy(z["rd_pred_nhwc"]).permute(0, 3, 1, 2).contiguous().to(dev).requires_grad_(True)
    loss = dr.recon_data(pred, torch.from_numpy(z["rd_gt"]).to(dev))
    loss.backward()
    assert abs(float(loss) - float(z["recon_data_c0"])) < 2e-6
    np.testing.assert_allclose(pred.grad.permute(0, 2, 3, 1).cpu().numpy(), z["recon_data_c0__d_pred_nhwc"], rtol=1e-4, atol=1e-9)


def test_forward_is_deterministic_and_headline_size_properties(pkg):
    """BASELINE config 2 (smpl_uv_642, B=48, 128x128): size-independent properties instead of an oracle run."""
    dr, att, datt, gt, inp, proj, H, W, dev = _setup(pkg, "smpl_uv_642", 48, 128, seed=0, imn=False)
    with torch.no_grad():
        r1, o1 = dr.render(no_mask=True, **datt)
        f1 = dr.last_face_idx.clone()
        r2, _ = dr.render(no_mask=True, **datt)
        f2 = dr.last_face_idx
    assert torch.equal(f1, f2) and torch.equal(r1, r2)                      # no atomics in the forward
    assert o1["imnormal"] is None
    a = r1[:, 3]
    assert float(r1.min()) >= 0 and float(r1.max()) <= 1
    assert bool((a[f1 >= 0] == 1).all())                                      # covered pixels have alpha exactly 1
    assert bool((a[f1 < 0] < 1).any()) and bool((a[f1 < 0] >= 0).all())       # uncovered ones carry the soft silhouette in [0, 1]
    cov = (f1 >= 0).float().mean().item()
    assert 0.1 < cov < 0.5
    # every winning face is front facing
    fn = o1["face_normals"]
    b_idx = torch.arange(48, device=dev).view(-1, 1, 1).expand_as(f1)[f1 >= 0]
    assert bool((fn[b_idx, f1[f1 >= 0].long(), 2] >= 0).all())
    # batch-permutation equivariance: rendering a permuted batch permutes the output
    perm = torch.randperm(48, device=dev)
    patt = {k: (v.detach()[perm] if torch.is_tensor(v) else v) for k, v in datt.items()}
    with torch.no_grad():
        rp, _ = dr.render(no_mask=True, **patt)
    assert torch.equal(rp, r1[perm])
    # translation of all vertices along the view ray changes nothing but depth order; mask IoU with itself is 1
    loss = dr.recon_data(r1, torch.cat([r1[:, :3], (a > 0.5).float().unsqueeze(1)], 1))
    assert float(loss) < 0.2


def test_gradients_reach_all_inputs_at_headline_size(pkg):
    dr, att, datt, gt, inp, proj, H, W, dev = _setup(pkg, "smpl_uv_642", 48, 128, seed=1, imn=False)
    rgbs, _ = dr.render(no_mask=True, **datt)
    dr.recon_data(rgbs, gt.to(dev), no_mask=True).backward()
    for k in LEAVES:
        g = datt[k].grad
        assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0, k
    # texture gradient is zero wherever no visible pixel samples (sum of |grad| > 0 only on a subset)
    assert float((datt["textures"].grad != 0).float().mean()) < 0.9


@pytest.mark.parametrize("dist", [None, 8.0])
def test_batches_of_192_images_take_the_large_batch_shapes_and_hold_the_bar(pkg, oracle, dist):
    """B = 192 at 128x128 crosses both large-batch thresholds of round 6: raster_fwd takes the one-tile-per-workgroup walk for 8-pixel bins
    (MM_WAVE_SHAPE_MIN_TILES = 49 152 tiles per launch, mm_raster_common.h: walk_block_mode) and gather_bwd sweeps with four lanes per item
    (MM_FL4_MIN_B = 128, mm_backward.hip).  The whole batch against the oracle at the full bar, the default dispatch against the forced
    256-thread shape bit for bit, and against the same images rendered 48 at a time (the small-batch kernels: eight lanes per item, the ticketed
    vertex backward), whose gradients differ from the large call's only in the last fixed-point step of the per-face sums."""
    B, S = 192, 128
    N = pkg._native
    res = {}
    for tag, opt in (("default", 0), ("block", N.OPT_WALK_BLOCK)):
        dr, att, datt, gt, inp, proj, H, W, dev = _setup(pkg, "smpl_uv_642", B, S, seed=23)
        if dist is not None:
            with torch.no_grad():
                datt["distances"].fill_(dist)
            inp["distances"] = np.full_like(inp["distances"], dist)
        dr.options = opt
        rgbs, out = dr.render(no_mask=True, **datt)
        dr.recon_data(rgbs, gt.to(dev), no_mask=True).backward()
        torch.cuda.synchronize()
        res[tag] = (rgbs.detach().clone(), dr.last_face_idx.clone(), {k: datt[k].grad.clone() for k in LEAVES})
    a, b = res["default"], res["block"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for k in LEAVES:
        assert torch.equal(a[2][k], b[2][k]), k
    # the oracle on the whole batch
    rgba_o, fidx_o, _, _ = oracle.render_forward(inp, H, W, True, proj)
    assert np.array_equal(a[1].cpu().numpy(), fidx_o)
    _close(a[0].permute(0, 2, 3, 1).cpu().numpy(), rgba_o)
    loss_o, g_o = oracle.step(inp, gt.numpy(), H, W, True, proj, image_weight=dr.image_weight)
    for k in LEAVES:
        _gclose(a[2][k].cpu().numpy(), g_o[k], what=k)
    # the same images 48 at a time: recon_data is a mean over the batch, so a quarter's gradients are 4x the large call's
    for q in range(4):
        sl = slice(48 * q, 48 * q + 48)
        part = {kk: (v[sl].detach().clone().requires_grad_(kk in LEAVES) if torch.is_tensor(v) and v.shape[:1] == (B,) else v) for kk, v in datt.items()}
        r1, _ = dr.render(no_mask=True, **part)
        assert torch.equal(dr.last_face_idx, a[1][sl]) and torch.equal(r1.detach(), a[0][sl])
        dr.recon_data(r1, gt[sl].to(dev), no_mask=True).backward()
        for kk in LEAVES:
            if datt[kk].shape[:1] != (B,):
                continue
            _gclose(a[2][kk][sl].cpu().numpy() * 4, part[kk].grad.cpu().numpy(), 2e-5, what=kk)


def test_stress_size_batch_independence_and_four_images_against_oracle(pkg, oracle):
    """BASELINE config 5 (smpl_uv: 13 776 faces, B=16, 512x512, texture 1024x512) at FULL size.  Size-independent property: an
    image's result does not depend on the batch it is rendered in (bitwise, forward and backward: every accumulation of the backward is
    an integer add) -- and FOUR images of the batch (the nearest and the farthest camera among them: the lightest and the heaviest tiles)
    are checked against the oracle one by one."""
    B, S = 16, 512
    dr, att, datt, gt, inp, proj, H, W, dev = _setup(pkg, "smpl_uv", B, S, seed=12, imn=False)
    rgbs, out = dr.render(no_mask=True, **datt)
    fidx = dr.last_face_idx.clone()
    dr.recon_data(rgbs, gt.to(dev), no_mask=True).backward()
    cov = (fidx >= 0).float().mean().item()
    assert 0.05 < cov < 0.6 and bool((rgbs[:, 3][fidx >= 0] == 1).all())
    dist = att["distances"].numpy()
    picks = sorted({int(dist.argmin()), int(dist.argmax()), 5, 11})
    assert len(picks) >= 3
    for k in picks:
        # image k alone
        one = {kk: (v[k:k + 1].detach().clone().requires_grad_(kk in LEAVES) if torch.is_tensor(v) else v) for kk, v in datt.items()}
        r1, o1 = dr.render(no_mask=True, **one)
        assert torch.equal(dr.last_face_idx[0], fidx[k]) and torch.equal(r1[0], rgbs[k].detach())
        # recon_data means over the batch: the full batch's gradient of image k is 1/B of the single-image gradient for the L1 term
        # and for the IoU term alike (both are means of per-image terms)
        dr.recon_data(r1, gt[k:k + 1].to(dev), no_mask=True).backward()
        for kk in LEAVES:
            _gclose(datt[kk].grad[k].cpu().numpy() * B, one[kk].grad[0].cpu().numpy(), 2e-5)
        # the oracle on that one image
        inp1 = {kk: (v[k:k + 1] if isinstance(v, np.ndarray) and v.shape[:1] == (B,) else v) for kk, v in inp.items()}
        rgba_o, fidx_o, _, _ = oracle.render_forward(inp1, H, W, True, proj)
        assert np.array_equal(dr.last_face_idx[0].cpu().numpy(), fidx_o[0]), k
        _close(r1[0].detach().permute(1, 2, 0).cpu().numpy(), rgba_o[0])
        loss_o, g_o = oracle.step(inp1, gt[k:k + 1].numpy(), H, W, True, proj, image_weight=dr.image_weight)
        for kk in LEAVES:
            _gclose(one[kk].grad.cpu().numpy(), g_o[kk], what=kk)


@pytest.mark.parametrize("use_ext", [True, False])
def test_unused_fused_loss_contributes_no_gradient(pkg, use_ext):
    """render_recon's loss output is ONE of the node's differentiable outputs.  Differentiating something that does not involve it --
    a regulariser through attributes['face_normals'] alone -- must give exactly what the un-fused render gives: the missing gradient
    of the loss is zero, not one (both host paths; also under a float64 default dtype, which must not leak into the 4-byte scalar)."""
    N = pkg._native
    if use_ext and N.torch_ext() is None:
        pytest.skip("mm_torch_ext is not built")
    ext, got = N.torch_ext(), []
    old = torch.get_default_dtype()
    try:
        N._EXT = ext if use_ext else None
        torch.set_default_dtype(torch.float64)
        for fused in (False, True):
            dr, att, datt, gt, inp, proj, H, W, dev = _setup(pkg, "smpl_uv_642", 4, 64, seed=5)
            if fused:
                loss, rgbs, out = dr.render_recon(gt.to(dev), no_mask=True, **datt)
            else:
                rgbs, out = dr.render(no_mask=True, **datt)
            w = torch.linspace(-1.0, 1.0, out["face_normals"].numel(), device=dev, dtype=torch.float32).reshape(out["face_normals"].shape)
            (out["face_normals"] * w).sum().backward(retain_graph=True)
            g_reg = {k: (None if datt[k].grad is None else datt[k].grad.clone()) for k in LEAVES}
            if fused:                                            # the loss afterwards, on the retained graph: accumulates on top
                loss.backward()
                g_both = {k: datt[k].grad.clone() for k in LEAVES}
            got.append((g_reg, g_both if fused else None))
    finally:
        N._EXT = ext
        torch.set_default_dtype(old)
    for k in LEAVES:
        a, b = got[0][0][k], got[1][0][k]
        assert (a is None) == (b is None) or (a is None and float(b.abs().max()) == 0) or (b is None and float(a.abs().max()) == 0), k
        if a is not None and b is not None:
            assert torch.equal(a, b), k                          # same kernels, same bits: nothing of the loss leaked in
    assert float(got[0][0]["vertices"].abs().max()) > 0
    assert float((got[1][1]["textures"] - 0).abs().max()) > 0   # and the loss's own gradient arrives when it IS differentiated
    assert not torch.equal(got[1][1]["vertices"], got[1][0]["vertices"])


def test_error_behaviour(pkg):
    dev = torch.device("cuda:0")
    dr = pkg.DiffRender(os.path.join(TEMPLATES, "sphere.npz"), 32)
    att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, 2, 32, 32)
    with pytest.raises(RuntimeError):
        dr.render(no_mask=True, **att)                                     # host tensors: no CPU fallback
    datt = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}
    bad = dict(datt); bad["bg"] = None
    with pytest.raises(TypeError):
        dr.render(no_mask=True, **bad)
    rgbs, _ = dr.render(no_mask=False, **bad)                               # bg is not needed without no_mask
    assert rgbs.shape == (2, 4, 32, 32)
    bad = dict(datt); bad["vertices"] = datt["vertices"][:, :10]
    with pytest.raises(RuntimeError):
        dr.render(**bad)
    with pytest.raises(KeyError):
        dr.render(**{k: v for k, v in datt.items() if k != "lights"})


@pytest.mark.parametrize("name,B,S,seed", [("sphere", 4, 64, 0), ("smpl_uv_642", 3, 50, 7),
                                           ("smpl_uv_642", 48, 128, 0)])        # the last one IS the bench step
def test_fused_step_matches_oracle_and_unfused(pkg, oracle, name, B, S, seed):
    """RenderLossStep: recon_data folded into the render kernels (MMRenderDesc.fused_*) vs the four-call sequence vs oracle."""
    import importlib
    stepmod = importlib.import_module("3d-magic-mirror_amd.step")
    dr, att, datt, gt, inp, proj, H, W, dev = _setup(pkg, name, B, S, seed=seed, imn=False)
    plain = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in datt.items()}
    fused = stepmod.RenderLossStep(dr, plain, gt.to(dev), no_mask=True, fused=True, loss_scale=0.5)
    unfused = stepmod.RenderLossStep(dr, plain, gt.to(dev), no_mask=True, fused=False, loss_scale=0.5)
    fused.run(); unfused.run()
    torch.cuda.synchronize()
    loss_o, g_o = oracle.step(inp, gt.numpy(), H, W, True, proj, image_weight=dr.image_weight)
    assert abs(float(fused.loss) - loss_o) < 2e-5 and abs(float(unfused.loss) - loss_o) < 2e-5
    assert torch.equal(fused.face_idx, unfused.face_idx) and torch.equal(fused.rgba, unfused.rgba)
    for k in LEAVES:
        _gclose(fused.grads[k].cpu().numpy() / 0.5, g_o[k], what='fused ' + k)
        _gclose(unfused.grads[k].cpu().numpy() / 0.5, g_o[k], what='unfused ' + k)


def test_chamfer_matches_bruteforce(pkg):
    """recon_att(chamfer=True): pytorch3d semantics (absent here) restated as a torch brute force."""
    import importlib
    ch = importlib.import_module("3d-magic-mirror_amd.chamfer")
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    x = torch.randn(5, 642, 3, generator=g).to(dev).requires_grad_(True)
    y = (torch.randn(5, 700, 3, generator=g) * 0.9).to(dev).requires_grad_(True)
    loss, nrm = ch.chamfer_distance(x, y)
    assert nrm is None
    loss.backward()
    x2, y2 = x.detach().clone().requires_grad_(True), y.detach().clone().requires_grad_(True)
    d = torch.cdist(x2.double(), y2.double()).pow(2)
    ref = d.min(2)[0].mean(1).mean(0) + d.min(1)[0].mean(1).mean(0)
    ref.backward()
    assert abs(float(loss) - float(ref)) < 1e-5
    torch.testing.assert_close(x.grad, x2.grad, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(y.grad, y2.grad, rtol=1e-4, atol=1e-6)
    dist, idx = ch.nearest_neighbour(x, y)
    assert torch.equal(idx, d.min(2)[1])
    # through the class: chamfer=True replaces the vertex term only
    dr = pkg.DiffRender(os.path.join(TEMPLATES, "sphere.npz"), 32)
    att, _ = pkg.synthetic.synthetic_batch(dr.vertices_init, 3, 32, 32, seed=1)
    att2, _ = pkg.synthetic.synthetic_batch(dr.vertices_init, 3, 32, 32, seed=2)
    A = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}
    A2 = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att2.items()}
    with_c = dr.recon_att(A, A2, L1=False, chamfer=True)
    without = dr.recon_att(A, A2, L1=False, chamfer=False)
    assert float(with_c[0]) == float(without[0]) and float(with_c[2]) == float(without[2])
    dd = torch.cdist(A["vertices"].double(), A2["vertices"].double()).pow(2)
    assert abs(float(with_c[1]) - float(dd.min(2)[0].mean(1).mean(0) + dd.min(1)[0].mean(1).mean(0))) < 1e-6


def test_concurrent_steps_on_four_streams_match_the_serial_result(pkg):
    """The bench's mode: independent steps in flight on four HIP streams.  Every step must still produce what it produces alone
    (the last-workgroup tickets and in-kernel counter clears must not interfere across streams or across repetitions)."""
    import importlib
    stepmod = importlib.import_module("3d-magic-mirror_amd.step")
    dev = torch.device("cuda:0")
    dr = pkg.DiffRender(os.path.join(TEMPLATES, "smpl_uv_642.npz"), 128, emit_imnormal=False)
    steps, refs = [], []
    for seed in range(4):
        att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, 12, 128, 128, seed=seed)
        datt = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}
        st = stepmod.RenderLossStep(dr, datt, gt.to(dev), no_mask=True, fused=True)
        st.run(); torch.cuda.synchronize()
        refs.append((float(st.loss), {k: v.clone() for k, v in st.grads.items() if v is not None}, st.face_idx.clone(), st.rgba.clone()))
        steps.append(st)
    streams = [torch.cuda.Stream(dev) for _ in steps]
    for rep in range(25):
        for st, s in zip(steps, streams):
            st.run(s)
    torch.cuda.synchronize()
    for st, (loss, grads, fidx, rgba) in zip(steps, refs):
        assert float(st.loss) == loss            # the fused loss is a sum of exact integer (fixed-point) sums: bitwise repeatable under overlap
        assert torch.equal(st.face_idx, fidx) and torch.equal(st.rgba, rgba)
        for k, g in grads.items():
            _gclose(st.grads[k].cpu().numpy(), g.cpu().numpy(), 2e-6)


@pytest.mark.parametrize("name,B,S,ratio", [("sphere", 3, 64, 1), ("smpl_uv_642", 4, 128, 2), ("sphere", 2, 36, 1)])
def test_fused_loss_carries_the_contour_term(pkg, oracle, name, B, S, ratio):
    """render_recon(..., contour=c) == render(...) + recon_data(..., contour=c) (networks.py:379-388; trainer.py:441 passes opt.lambda_contour):
    value and all eight input gradients, through the class API and through the C-ABI step object.  36x36: tiles that straddle the image edge."""
    import importlib
    stepmod = importlib.import_module("3d-magic-mirror_amd.step")
    dev = torch.device("cuda:0")
    dr = pkg.DiffRender(os.path.join(TEMPLATES, name + ".npz"), S, ratio=ratio, emit_imnormal=True)
    H, W = dr.render_height, dr.image_size
    att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, B, H, W, seed=21)
    gt = gt.to(dev)
    leaves = ("vertices", "textures", "lights", "bg", "azimuths", "elevations", "distances", "biases")

    def fresh():
        return {k: (v.to(dev).clone().requires_grad_(k in leaves) if torch.is_tensor(v) else v) for k, v in att.items()}
    for c in (0.5, 3.0):
        a1 = fresh()
        rgbs, _ = dr.render(no_mask=True, **a1)
        l1 = dr.recon_data(rgbs, gt, no_mask=True, contour=c)
        l1.backward()
        a2 = fresh()
        l2, rgbs2, _ = dr.render_recon(gt, no_mask=True, contour=c, **a2)
        l2.backward()
        assert torch.equal(rgbs2, rgbs.detach())
        assert abs(float(l1) - float(l2)) < 2e-6 * max(1.0, abs(float(l1)))
        # ... and the oracle's recon_data (pinned to the reference's own value and gradient with contour 0.5: losses.npz) on the rendered image
        lo, dpo = oracle.recon_data(rgbs.detach().cpu().numpy(), gt.cpu().numpy(), image_weight=dr.image_weight, contour=c, want_grad=True)
        assert abs(float(l2) - float(lo)) < 2e-6 * max(1.0, abs(float(lo)))
        l0 = dr.recon_data(rgbs.detach(), gt, no_mask=True, contour=0)
        assert abs(float(l1) - float(l0)) > 1e-5                 # the term is there
        for k in leaves:
            g1, g2 = a1[k].grad, a2[k].grad
            scale = float(g1.abs().max()) + 1e-12
            assert float((g1 - g2).abs().max()) <= 2e-5 * scale + 1e-9, (k, c, float((g1 - g2).abs().max()), scale)
    # the step object the bench drives: fused with contour == its own un-fused form
    datt = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}
    sf = stepmod.RenderLossStep(dr, datt, gt, no_mask=True, contour=0.5, fused=True, emit_imnormal=True)
    su = stepmod.RenderLossStep(dr, datt, gt, no_mask=True, contour=0.5, fused=False, emit_imnormal=True)
    assert sf.fused and not su.fused
    sf.run(); su.run(); torch.cuda.synchronize()
    assert abs(float(sf.loss) - float(su.loss)) < 2e-6 * max(1.0, abs(float(su.loss)))
    for k in ("vertices", "textures", "lights", "azimuths"):
        g1, g2 = su.grads[k], sf.grads[k]
        assert float((g1 - g2).abs().max()) <= 2e-5 * (float(g1.abs().max()) + 1e-12) + 1e-9, k


def test_fused_contour_needs_sizes_that_are_multiples_of_four(pkg):
    dev = torch.device("cuda:0")
    dr = pkg.DiffRender(os.path.join(TEMPLATES, "sphere.npz"), 30, emit_imnormal=False)
    att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, 2, 30, 30, seed=2)
    datt = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in att.items()}
    with pytest.raises(ValueError, match="multiples of 4"):
        dr.render_recon(gt.to(dev), no_mask=True, contour=0.5, **datt)
    loss, _, _ = dr.render_recon(gt.to(dev), no_mask=True, contour=0, **datt)     # contour = 0: any size
    rgbs, _ = dr.render(no_mask=True, **datt)
    assert abs(float(loss) - float(dr.recon_data(rgbs, gt.to(dev), no_mask=True))) < 1e-6


@pytest.mark.parametrize("B,H,W", [(3, 30, 30), (2, 64, 48), (2, 37, 21), (1, 8, 8)])
def test_unfused_contour_backward_matches_the_oracle_and_is_bitwise_reproducible(pkg, oracle, B, H, W):
    """recon_data(contour > 0) on its own (networks.py:379-388) at sizes that are NOT multiples of 4 too (the two nearest-neighbour resamplings then
    select pixels outside a pixel's own 4x4 block): value and dL/dpred against the oracle (pinned to the reference's own numbers in losses.npz), and
    the backward twice -- it gathers per selected pixel in a fixed order, no floating-point atomics."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(H * 100 + W)
    pred = torch.rand(B, 4, H, W, generator=g)
    gt = torch.rand(B, 4, H, W, generator=g)
    gt[:, 3] = (gt[:, 3] > 0.5).float()
    pred[:, 3] = torch.where(torch.rand(B, H, W, generator=g) > 0.5, pred[:, 3], torch.ones(B, H, W))   # flat regions: |0| gradients
    dr = pkg.DiffRender(os.path.join(TEMPLATES, "sphere.npz"), 32)
    grads = []
    for _ in range(3):
        p = pred.to(dev).clone().requires_grad_(True)
        loss = dr.recon_data(p, gt.to(dev), no_mask=True, contour=0.7)
        loss.backward()
        grads.append(p.grad.clone())
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])
    lo, dpo = oracle.recon_data(pred.numpy(), gt.numpy(), image_weight=dr.image_weight, contour=0.7, want_grad=True)
    assert abs(float(loss) - lo) < 2e-6 * max(1.0, abs(lo))
    np.testing.assert_allclose(grads[0].cpu().numpy(), dpo, rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize("use_ext", [True, False])
def test_camera_scalars_of_shape_B_by_1_get_gradients_of_their_own_shape(pkg, use_ext):
    """azimuths / elevations / distances arrive as (B,1) from some callers (a Linear head's output): the kernels see (B), the gradients must come
    back as (B,1) -- autograd rejects a (B) gradient for a (B,1) input (advisor r05) -- with the bits of the (B) call; render and render_geometry."""
    N = pkg._native
    if use_ext and N.torch_ext() is None:
        pytest.skip("mm_torch_ext is not built")
    ext = N.torch_ext()
    try:
        N._EXT = ext if use_ext else None
        got = []
        for col in (False, True):
            dr, att, datt, gt, inp, proj, H, W, dev = _setup(pkg, "smpl_uv_642", 3, 64, seed=9)
            if col:
                for k in ("azimuths", "elevations", "distances"):
                    datt[k] = datt[k].detach().reshape(-1, 1).requires_grad_(True)
            rgbs, out = dr.render(no_mask=True, **datt)
            geo = dr.render_geometry(**{k: v for k, v in datt.items()})
            (dr.recon_data(rgbs, gt.to(dev), no_mask=True) + 1e-3 * geo["face_normals"].sum()).backward()
            for k in ("azimuths", "elevations", "distances"):
                assert datt[k].grad.shape == datt[k].shape, (k, datt[k].grad.shape)
            got.append({k: datt[k].grad.reshape(-1).clone() for k in ("azimuths", "elevations", "distances", "vertices")})
        for k in got[0]:
            assert torch.equal(got[0][k], got[1][k]), k
    finally:
        N._EXT = ext


def test_a_geometry_only_render_asks_for_the_vertex_stages_workspace_alone(pkg):
    import ctypes
    N = pkg._native
    dr, att, datt, gt, inp, proj, H, W, dev = _setup(pkg, "smpl_uv_642", 48, 128, seed=0)
    d = dr._desc(dr._static(dev), 48, True, datt["vertices"], datt["textures"], datt["lights"], datt["bg"], datt["azimuths"], datt["elevations"],
                 datt["distances"], datt["biases"], None, None, None, None)
    full = N.lib().mm_query_workspace(ctypes.byref(d))
    d.geometry_only = 1
    geo = N.lib().mm_query_workspace(ctypes.byref(d))
    assert 0 < geo < full // 10, (geo, full)                     # 3 MB of face records and counters against 71 MB
    # and the class API's geometry render still gives render's normals and its own gradients with it (bitwise: test_geometry_only_render_...)
    a = dr.render_geometry(**datt)
    r, o = dr.render(no_mask=True, **datt)
    assert torch.equal(a["face_normals"], o["face_normals"])
