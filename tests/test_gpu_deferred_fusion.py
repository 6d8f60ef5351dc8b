"""DEFERRED FUSION (MMRenderDesc.fused_totals, csrc/mm_torch_ext.cpp): the un-modified trainer's `pred, att = render(...)` ... `recon_data(pred, gt)`
(trainer.py:276,441) with the fused backward.  The contract under test: every observable number -- the loss, the image, all eight input gradients --
has the BITS of the two separate passes (mm_recon_data_backward writing dL/d image, mm_render_backward reading it), whether or not the deferral
applies; and it applies exactly when `pred` is the untouched image of a render."""
import os

import numpy as np
import pytest
import torch

from conftest import TEMPLATES

pytestmark = pytest.mark.gpu
LEAVES = ("vertices", "textures", "lights", "bg", "azimuths", "elevations", "distances", "biases")


def _setup(pkg, name, B, S, ratio=1, seed=0, defer=True, imn=True):
    dev = torch.device("cuda:0")
    dr = pkg.DiffRender(os.path.join(TEMPLATES, name + ".npz"), S, ratio=ratio, emit_imnormal=imn)
    dr.defer_recon_fusion = defer
    H, W = dr.render_height, dr.image_size
    att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, B, H, W, seed=seed)
    datt = {k: (v.to(dev).requires_grad_(k in LEAVES) if torch.is_tensor(v) else v) for k, v in att.items()}
    return dr, datt, gt.to(dev), dev


def _ext(pkg):
    ext = pkg._native.torch_ext()
    if ext is None or not hasattr(ext, "deferrable"):
        pytest.skip("mm_torch_ext is not built: the Python nodes never defer")
    return ext


def _grads(datt, no_mask=True):
    return {k: (None if datt[k].grad is None else datt[k].grad.clone()) for k in LEAVES if no_mask or k != "bg"}


def _same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        assert (a[k] is None) == (b[k] is None), k
        if a[k] is not None:
            assert torch.equal(a[k], b[k]), (k, float((a[k] - b[k]).abs().max()))
            assert float(a[k].abs().max()) > 0, k


@pytest.mark.parametrize("name,B,S,ratio,no_mask,seed", [("smpl_uv_642", 6, 96, 1, True, 31), ("sphere", 3, 50, 1.4, False, 32),
                                                         ("smpl_uv_642", 48, 128, 1, True, 0)])
def test_recon_data_on_a_renders_image_is_deferred_and_has_the_separate_passes_bits(pkg, name, B, S, ratio, no_mask, seed):
    ext = _ext(pkg)
    got = []
    for defer in (False, True):
        dr, datt, gt, dev = _setup(pkg, name, B, S, ratio=ratio, seed=seed, defer=defer)
        rgbs, out = dr.render(no_mask=no_mask, **datt)
        assert ext.deferrable(rgbs) == defer
        loss = dr.recon_data(rgbs, gt, no_mask=no_mask)
        assert not ext.deferrable(rgbs) or not defer           # taken: a second recon_data on the same image runs on its own
        (2.5 * loss).backward()
        got.append((loss.detach().clone(), rgbs.detach().clone(), _grads(datt, no_mask)))
    assert torch.equal(got[0][0], got[1][0]) and torch.equal(got[0][1], got[1][1])
    _same(got[0][2], got[1][2])


def test_deferred_recon_data_with_other_consumers_of_the_image_and_of_the_normals(pkg):
    """The image also feeds something else (a discriminator in the trainer), and a regulariser hangs on attributes['face_normals']: the image's
    other gradient arrives as grad_rgba and is ADDED inside the pixel pass -- autograd's own sum, bit for bit."""
    _ext(pkg)
    got = []
    for defer in (False, True):
        dr, datt, gt, dev = _setup(pkg, "smpl_uv_642", 5, 80, seed=7, defer=defer)
        rgbs, out = dr.render(no_mask=True, **datt)
        w = torch.linspace(-1.0, 1.0, rgbs.numel(), device=dev).reshape(rgbs.shape)
        wn = torch.linspace(0.5, -0.5, out["face_normals"].numel(), device=dev).reshape(out["face_normals"].shape)
        loss = dr.recon_data(rgbs, gt, no_mask=True) + 1e-5 * (rgbs * w).sum() + 1e-3 * (out["face_normals"] * wn).sum()
        loss.backward()
        got.append((loss.detach().clone(), _grads(datt)))
    assert torch.equal(got[0][0], got[1][0])
    _same(got[0][1], got[1][1])


def test_only_the_untouched_image_of_a_render_defers_and_everything_else_gives_the_same_numbers(pkg):
    ext = _ext(pkg)
    ref = None
    cases = ("plain", "off", "clone", "inplace", "contour", "second", "scaled")
    for case in cases:
        dr, datt, gt, dev = _setup(pkg, "smpl_uv_642", 4, 64, seed=11, defer=case != "off")
        rgbs, out = dr.render(no_mask=True, **datt)
        pred, contour = rgbs, 0
        if case == "clone":
            pred = rgbs.clone()
        elif case == "inplace":
            with torch.no_grad():
                rgbs.mul_(1.0)                                   # same values, but the image is no longer "as render wrote it"
        elif case == "scaled":
            pred = rgbs * 1.0
        elif case == "contour":
            contour = 0.5
        expect = case in ("plain", "second", "contour")          # (the image qualifies; a contour term makes recon_data itself decline)
        assert ext.deferrable(pred) == expect, case
        loss = dr.recon_data(pred, gt, no_mask=True, contour=contour)
        assert ext.deferrable(pred) == (case == "contour"), case  # taken by the recon_data above -- unless it declined
        if case == "second":
            assert not ext.deferrable(pred)
            loss = 0.5 * loss + 0.5 * dr.recon_data(pred, gt, no_mask=True)      # the second one runs on its own: the two gradients meet in grad_rgba
        loss.backward()
        g = _grads(datt)
        if case == "plain":
            ref = (loss.detach().clone(), g)
        elif case in ("off", "clone", "inplace", "scaled"):
            assert torch.equal(loss.detach(), ref[0]), case
            _same(ref[1], g)
        elif case == "second":
            for k in g:                                          # (0.5 a + 0.5 a: equal up to the rounding of the halves' sum)
                assert float((g[k] - ref[1][k]).abs().max()) <= 1e-6 * float(ref[1][k].abs().max()), (case, k)


def test_a_deferred_loss_that_is_not_differentiated_contributes_nothing_and_backward_twice_works(pkg):
    _ext(pkg)
    dr, datt, gt, dev = _setup(pkg, "smpl_uv_642", 4, 64, seed=5)
    rgbs, out = dr.render(no_mask=True, **datt)
    loss = dr.recon_data(rgbs, gt, no_mask=True)
    wn = torch.linspace(-1.0, 1.0, out["face_normals"].numel(), device=dev).reshape(out["face_normals"].shape)
    reg = (out["face_normals"] * wn).sum()
    reg.backward(retain_graph=True)                              # the loss takes no part: its token has no gradient, the render's backward is the plain one
    g_reg = _grads(datt)
    assert g_reg["textures"] is None or float(g_reg["textures"].abs().max()) == 0
    dr2, datt2, gt2, _ = _setup(pkg, "smpl_uv_642", 4, 64, seed=5, defer=False)
    rgbs2, out2 = dr2.render(no_mask=True, **datt2)
    (out2["face_normals"] * wn).sum().backward()
    for k in ("vertices", "azimuths", "elevations", "distances", "biases"):
        assert torch.equal(g_reg[k], datt2[k].grad), k
    for k in LEAVES:
        if datt[k].grad is not None:
            datt[k].grad = None
    loss.backward(retain_graph=True)
    g1 = _grads(datt)
    for k in LEAVES:
        datt[k].grad = None
    loss.backward()
    _same(g1, _grads(datt))
    assert float(g1["textures"].abs().max()) > 0


def test_the_python_nodes_and_the_deferring_cpp_nodes_agree_bit_for_bit(pkg):
    N = pkg._native
    ext = _ext(pkg)
    got = []
    try:
        for use_ext in (True, False):
            N._EXT = ext if use_ext else None
            dr, datt, gt, dev = _setup(pkg, "smpl_uv_642", 5, 80, seed=41)
            rgbs, out = dr.render(no_mask=True, **datt)
            loss = dr.recon_data(rgbs, gt, no_mask=True) + 1e-3 * out["face_normals"].sum()
            loss.backward()
            got.append((loss.detach().clone(), rgbs.detach().clone(), _grads(datt)))
    finally:
        N._EXT = ext
    assert torch.equal(got[0][0], got[1][0]) and torch.equal(got[0][1], got[1][1])
    _same(got[0][2], got[1][2])


def test_deferred_totals_at_the_c_abi(pkg):
    """mm_render_backward with fused_gt + fused_totals (no autograd involved) against mm_recon_data_backward -> mm_render_backward, and its argument
    checks: totals without a target, with a contour weight, or in a forward call are refused."""
    import ctypes
    import importlib
    N = pkg._native
    stepmod = importlib.import_module("3d-magic-mirror_amd.step")
    dr, datt, gt, dev = _setup(pkg, "smpl_uv_642", 4, 64, seed=3)
    plain = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in datt.items()}
    ref = stepmod.RenderLossStep(dr, plain, gt, no_mask=True, fused=False)
    ref.run(); torch.cuda.synchronize()
    st = stepmod.RenderLossStep(dr, plain, gt, no_mask=True, fused=False)
    st.run_deferred(); torch.cuda.synchronize()
    assert torch.equal(st.loss, ref.loss) and torch.equal(st.rgba, ref.rgba)
    for k, g in ref.grads.items():
        if g is not None:
            assert torch.equal(st.grads[k], g), k
    d = st.render_desc()
    d.fused_totals = st.recon_totals_ptr()
    g = st.grads_struct()
    assert N.lib().mm_render_backward(ctypes.byref(d), ctypes.byref(g), None) == -1        # MM_ERR_NULL_POINTER: totals without fused_gt
    d.fused_gt = N.ptr(gt); d.fused_contour = 0.5
    assert N.lib().mm_render_backward(ctypes.byref(d), ctypes.byref(g), None) == -5        # MM_ERR_UNSUPPORTED: the contour term is not deferred
    d.fused_contour = 0.0
    assert N.lib().mm_render_forward(ctypes.byref(d), None) == -5                          # ... and a forward has nothing to defer


def test_deferred_steps_free_everything_they_held(pkg):
    """The render node keeps the recon_data's target and workspace alive for its backward (the mailbox) -- and lets go of them with the graph: no
    reference cycle through the token (the token tensor is made at recon_data time, not kept by the node).  Device memory in use is flat over steps
    that bring a fresh target each."""
    _ext(pkg)
    dr, datt, gt, dev = _setup(pkg, "smpl_uv_642", 8, 64, seed=3)
    used = []
    for step in range(24):
        for k in LEAVES:
            datt[k].grad = None
        g = gt.clone() + 0.0                                     # a new target tensor every step, as a data loader hands them out
        rgbs, out = dr.render(no_mask=True, **datt)
        loss = dr.recon_data(rgbs, g, no_mask=True)
        loss.backward()
        del rgbs, out, loss, g
        torch.cuda.synchronize()
        used.append(torch.cuda.memory_allocated(dev))
    assert used[-1] == used[8], (used[8], used[-1])
    # ... and without a backward (an evaluation loop that forgot no_grad): the graph dies with its last reference
    for step in range(12):
        g = gt.clone() + 0.0
        rgbs, out = dr.render(no_mask=True, **datt)
        loss = dr.recon_data(rgbs, g, no_mask=True)
        del rgbs, out, loss, g
        torch.cuda.synchronize()
        used.append(torch.cuda.memory_allocated(dev))
    assert used[-1] == used[-6], (used[-6], used[-1])
