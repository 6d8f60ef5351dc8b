"""Texture-flow sampling: mm_texture_flow_* against the reference's own formulation
(/root/reference/network/model_res.py:597-612, makeup == 0) evaluated by torch on the CPU in fp64:
    cat([t, t.flip([2])], 2),  t = F.grid_sample(img, flow.permute(0,2,3,1), mode='bicubic', align_corners=True)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _reference(img, flow):
    t = F.grid_sample(img, flow.permute(0, 2, 3, 1), mode='bicubic', align_corners=True)
    return torch.cat([t, t.flip([2])], dim=2)


@pytest.mark.parametrize("B,C,H,W,Ho,Wo,spread,seed", [
    (3, 3, 32, 32, 32, 32, 0.9, 0),       # flow inside the image
    (2, 3, 40, 24, 64, 20, 1.3, 1),       # flow leaves the image (zeros padding), non-square, upsampling
    (48, 3, 128, 128, 128, 128, 1.0, 2),  # the reference's CUB size: (48,3,256,128) textures
    (1, 4, 5, 7, 3, 130, 1.1, 3),         # tiny image, ragged widths, 4 channels
])
def test_texture_flow_matches_torch_grid_sample(pkg, B, C, H, W, Ho, Wo, spread, seed):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, C, H, W, generator=g)
    # smooth identity-like flow plus noise, as the decoder's tanh output looks
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, Ho), torch.linspace(-1, 1, Wo), indexing="ij")
    flow = (torch.stack([xs, ys], 0)[None] * spread + 0.15 * torch.randn(B, 2, Ho, Wo, generator=g)).contiguous()
    wgt = torch.randn(B, C, 2 * Ho, Wo, generator=g)

    img_h, flow_h = img.double().requires_grad_(True), flow.double().requires_grad_(True)
    ref = _reference(img_h, flow_h)
    (ref * wgt.double()).sum().backward()

    img_d, flow_d = img.to(DEV).requires_grad_(True), flow.to(DEV).requires_grad_(True)
    out = pkg.sample_texture(img_d, flow_d)
    assert out.shape == (B, C, 2 * Ho, Wo)
    assert torch.equal(out[:, :, :Ho], out[:, :, Ho:].flip([2]))                 # back == mirrored front, exactly
    (out * wgt.to(DEV)).sum().backward()

    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=0, atol=2e-5)
    for got, want, nm in ((flow_d.grad, flow_h.grad, "flow"), (img_d.grad, img_h.grad, "image")):
        scale = max(1.0, float(want.abs().max()))
        err = float((got.cpu().double() - want).abs().max())
        assert err <= 1e-4 * scale, (nm, err, scale)
    # flow-only gradient (the trainer's case: the image is data) takes the no-atomics path
    flow_d2 = flow.to(DEV).requires_grad_(True)
    (pkg.sample_texture(img.to(DEV), flow_d2) * wgt.to(DEV)).sum().backward()
    assert torch.equal(flow_d2.grad, flow_d.grad)


def test_texture_flow_feeds_the_renderer(pkg):
    """End to end: flow -> texture -> render -> loss -> d loss / d flow is finite and non-zero where the mesh is visible."""
    import os
    from conftest import TEMPLATES
    dr = pkg.DiffRender(os.path.join(TEMPLATES, "sphere.npz"), 64)
    att, gt = pkg.synthetic.synthetic_batch(dr.vertices_init, 2, 64, 64, seed=5)
    datt = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in att.items()}
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, 64), torch.linspace(-1, 1, 64), indexing="ij")
    flow = (torch.stack([xs, ys], 0)[None].repeat(2, 1, 1, 1) * 0.9).to(DEV).requires_grad_(True)
    datt["textures"] = pkg.sample_texture(gt[:, :3].to(DEV), flow)
    assert datt["textures"].shape == att["textures"].shape
    rgbs, _ = dr.render(no_mask=True, **datt)
    dr.recon_data(rgbs, gt.to(DEV), no_mask=True).backward()
    assert torch.isfinite(flow.grad).all() and float(flow.grad.abs().max()) > 0


def test_texture_flow_validation(pkg):
    import ctypes
    N = pkg._native
    d = N.MMTexFlowDesc()
    assert N.lib().mm_texture_flow_forward(ctypes.byref(d), None) == -2
    d.B, d.C, d.H, d.W, d.Ho, d.Wo = 1, 3, 8, 8, 8, 8
    assert N.lib().mm_texture_flow_forward(ctypes.byref(d), None) == -1
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.sample_texture(torch.zeros(1, 3, 8, 8), torch.zeros(1, 2, 8, 8))
