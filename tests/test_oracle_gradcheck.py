"""fp64 central-difference check of the oracle's full backward (render -> recon_data) for every differentiable input.
The oracle's gradient is derived, not recalled, so it is checked against its own forward."""
import numpy as np
import pytest

from conftest import make_inputs

KEYS = ("vertices", "textures", "lights", "bg", "azimuths", "elevations", "distances", "biases")


def _loss(oracle, inp, gt, proj, H, W, no_mask, wfn, contour):
    rgba, fidx, fn, _ = oracle.render_forward(inp, H, W, no_mask, proj, dtype=np.float64)
    pred = rgba.transpose(0, 3, 1, 2)
    return oracle.recon_data(pred, gt, image_weight=1.0, contour=contour, dtype=np.float64) + float((fn * wfn).sum()), rgba


@pytest.mark.parametrize("no_mask,contour,seed", [(True, 0.0, 0), (False, 0.0, 1), (True, 0.3, 2)])
def test_full_backward_matches_central_differences(oracle, no_mask, contour, seed):
    B, H, W = 2, 32, 32
    inp, gt, proj = make_inputs("sphere", B, H, W, seed=seed)
    inp = {k: (v.astype(np.float64) if (hasattr(v, "dtype") and v.dtype == np.float32) else v) for k, v in inp.items()}
    inp["distances"] = np.asarray([2.6, 3.4])             # close enough that faces span several pixels
    gt = gt.astype(np.float64); proj = proj.astype(np.float64)
    rng = np.random.default_rng(seed)
    wfn = rng.normal(size=(B, inp["faces"].shape[0], 3)) * 1e-3
    base, rgba = _loss(oracle, inp, gt, proj, H, W, no_mask, wfn, contour)
    _, dpred = oracle.recon_data(rgba.transpose(0, 3, 1, 2), gt, image_weight=1.0, contour=contour, want_grad=True, dtype=np.float64)
    g = oracle.render_backward(inp, H, W, no_mask, proj, np.ascontiguousarray(dpred.transpose(0, 2, 3, 1)), wfn, dtype=np.float64)
    h = 1e-6
    for k in KEYS:
        if k == "bg" and not no_mask:
            continue
        d = rng.normal(size=inp[k].shape)
        d /= np.linalg.norm(d)
        ip, im = dict(inp), dict(inp)
        ip[k] = inp[k] + h * d; im[k] = inp[k] - h * d
        fd = (_loss(oracle, ip, gt, proj, H, W, no_mask, wfn, contour)[0] - _loss(oracle, im, gt, proj, H, W, no_mask, wfn, contour)[0]) / (2 * h)
        an = float((g[k] * d).sum())
        assert abs(fd - an) <= 2e-5 * max(1.0, abs(fd)) + 1e-7, (k, fd, an)
        if k in ("vertices", "distances", "azimuths", "biases", "lights", "textures"):
            assert abs(an) > 1e-7, (k, an)                # the check is not vacuous


def test_f32_backward_tracks_f64(oracle):
    B, H, W = 2, 32, 32
    inp, gt, proj = make_inputs("sphere", B, H, W, seed=4)
    rgba, fidx, fn, _ = oracle.render_forward(inp, H, W, True, proj)
    _, dpred = oracle.recon_data(rgba.transpose(0, 3, 1, 2), gt, image_weight=1.0, want_grad=True)
    drgba = np.ascontiguousarray(dpred.transpose(0, 2, 3, 1))
    g32 = oracle.render_backward(inp, H, W, True, proj, drgba)
    inp64 = {k: (v.astype(np.float64) if (hasattr(v, "dtype") and v.dtype == np.float32) else v) for k, v in inp.items()}
    r64, f64, _, _ = oracle.render_forward(inp64, H, W, True, proj.astype(np.float64), dtype=np.float64)
    if (f64 == fidx).all():                               # identical visibility -> gradients must agree closely
        g64 = oracle.render_backward(inp64, H, W, True, proj.astype(np.float64), drgba.astype(np.float64), dtype=np.float64)
        for k in KEYS:
            scale = max(1e-6, float(np.abs(g64[k]).max()))
            assert float(np.abs(g32[k] - g64[k]).max()) <= 2e-3 * scale, k
