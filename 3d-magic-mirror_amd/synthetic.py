"""Seeded synthetic attribute batches for the render path.

Value ranges follow what the reference's encoders emit (network/model_res.py:206-216,333-337,392-395,610-611):
the same draws feed the CPU oracle and the HIP path in tests and in bench.py.
"""
import torch


def synthetic_batch(vertices_init, B, H, W, seed=0, device="cpu", with_bg=True):
    """Returns (attributes dict, gt_data (B,4,H,W)).  Textures are (B,3,2H,W), mirrored top/bottom."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    V = vertices_init.shape[0]

    def U(*shape, lo=0.0, hi=1.0):
        return torch.rand(*shape, generator=g) * (hi - lo) + lo

    delta = 0.05 * torch.randn(B, V, 3, generator=g)
    delta = delta - delta.mean(dim=1, keepdim=True)
    half = U(B, 3, H, W)
    scale = torch.tensor([0.5] + [0.1] * 8)
    att = {
        "azimuths": U(B, lo=-180.0, hi=180.0),
        "elevations": U(B, lo=0.0, hi=30.0),
        "distances": U(B, lo=2.0, hi=7.0),
        "biases": U(B, 2, lo=-0.3, hi=0.3),
        "vertices": vertices_init.reshape(1, V, 3).float() + delta,
        "delta_vertices": delta,
        "textures": torch.cat([half, half.flip(2)], dim=2),
        "lights": torch.tensor([3.0] + [0.0] * 8) + scale * U(B, 9, lo=-1.0, hi=1.0),
        "bg": U(B, 3, H, W) if with_bg else None,
        "img_feats": None,
    }
    rgb = U(B, 3, H, W)
    ys = (torch.arange(H).float() + 0.5 - H / 2.0) / (0.4 * H)
    xs = (torch.arange(W).float() + 0.5 - W / 2.0) / (0.35 * W)
    ell = ((ys[:, None] ** 2 + xs[None, :] ** 2) <= 1.0).float()
    gt = torch.cat([rgb, ell.expand(B, 1, H, W)], dim=1).contiguous()
    att = {k: (v.to(device).contiguous() if torch.is_tensor(v) else v) for k, v in att.items()}
    return att, gt.to(device)
