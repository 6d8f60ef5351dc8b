"""Chamfer distance with pytorch3d's semantics (pytorch3d.loss.chamfer_distance, defaults: point_reduction='mean',
batch_reduction='mean', no normals), which the reference imports at /root/reference/networks.py:19 and calls at :342,356
and trainer.py:445,469,483.  pytorch3d is absent here (and has no ROCm wheel): parity is against a torch brute force.

The O(N*M) nearest-neighbour search runs in the HIP kernel behind `mm_chamfer_nearest` (both directions, one launch); the loss is then a differentiable gather
(gradients flow to both clouds exactly as through knn_points' returned distances)."""
import ctypes

import torch

from . import _native as N


def nearest_neighbour(x, y):
    """x (B,N,3), y (B,M,3) device tensors -> (squared distance (B,N), index (B,N) int64) of the nearest y for every x."""
    N.require_device(x, y)
    xc, yc = x.detach().float().contiguous(), y.detach().float().contiguous()
    B, n, _ = xc.shape
    m = yc.shape[1]
    dist = torch.empty((B, n), device=x.device, dtype=torch.float32)
    idx = torch.empty((B, n), device=x.device, dtype=torch.int32)
    N.check(N.lib().mm_nearest_neighbour(B, n, m, N.ptr(xc), N.ptr(yc), N.ptr(dist), N.ptr(idx), N.current_stream(x.device)),
            "mm_nearest_neighbour")
    return dist, idx.long()


def nearest_both(x, y):
    """Both directions in ONE launch (mm_chamfer_nearest): (index (B,N) of the nearest y for every x, index (B,M) of the nearest x for every y)."""
    N.require_device(x, y)
    xc, yc = x.detach().float().contiguous(), y.detach().float().contiguous()
    B, n, _ = xc.shape
    m = yc.shape[1]
    dist = torch.empty((B, n + m), device=x.device, dtype=torch.float32)
    idx = torch.empty((B * (n + m),), device=x.device, dtype=torch.int32)
    dx, dy = dist.view(-1)[:B * n], dist.view(-1)[B * n:]
    ix, iy = idx[:B * n], idx[B * n:]
    N.check(N.lib().mm_chamfer_nearest(B, n, m, N.ptr(xc), N.ptr(yc), N.ptr(dx), N.ptr(ix), N.ptr(dy), N.ptr(iy), N.current_stream(x.device)),
            "mm_chamfer_nearest")
    return ix.view(B, n).long(), iy.view(B, m).long()


def chamfer_distance(x, y):
    """Returns (loss, None) like pytorch3d: mean_b [ mean_i min_j |x_i - y_j|^2 + mean_j min_i |x_i - y_j|^2 ]."""
    if x.dim() != 3 or y.dim() != 3 or x.shape[0] != y.shape[0] or x.shape[2] != 3 or y.shape[2] != 3:
        raise ValueError("chamfer_distance expects (B,N,3) and (B,M,3)")
    ix, iy = nearest_both(x, y)
    cham_x = (x - torch.gather(y, 1, ix.unsqueeze(-1).expand(-1, -1, 3))).pow(2).sum(-1)     # (B,N)
    cham_y = (y - torch.gather(x, 1, iy.unsqueeze(-1).expand(-1, -1, 3))).pow(2).sum(-1)     # (B,M)
    loss = cham_x.mean(1).mean(0) + cham_y.mean(1).mean(0)
    return loss, None
