"""CPU restatement (plain torch ops, fp32 or fp64) of the reference's mesh regularisers and attribute losses
(/root/reference/networks.py:326-491).  TEST INFRASTRUCTURE ONLY: nothing in the product imports this module; tests use it as
the checker of the HIP kernels (mm_mesh_reg_*), and tests/test_abi_and_host.py pins it against outputs and gradients of the
reference itself (tests/golden/losses.npz, minted by tests/golden/make_golden.py under kaolin stubs).

Every function takes the template tables it needs explicitly (``t`` = any object with the DiffRender attributes
``flip_index, sign_init, edges, edge2faces, vertices_laplacian_matrix, ratio, lambda_lpl, lambda_flat``).
"""
import math

import torch


def recon_att(pred_att, target_att, L1=False, azim=1, shape_loss=None):
    """networks.py:326-362 with chamfer=False (or ``shape_loss`` supplied by the caller)."""
    def angle2xy(angle):
        angle = angle * math.pi / 180.0
        return torch.stack([torch.cos(angle), torch.sin(angle)], 1)

    dist = (lambda a, b: torch.abs(a - b).mean()) if L1 else (lambda a, b: torch.pow(a - b, 2).mean())
    loss_azim = dist(angle2xy(pred_att['azimuths']), angle2xy(target_att['azimuths']))
    loss_elev = dist(angle2xy(pred_att['elevations']), angle2xy(target_att['elevations']))
    loss_dist = dist(pred_att['distances'], target_att['distances'])
    loss_bias = dist(pred_att['biases'], target_att['biases'])
    loss_cam = azim * loss_azim + loss_elev + loss_dist
    loss_shape = dist(pred_att['vertices'], target_att['vertices']) if shape_loss is None else shape_loss
    loss_texture = dist(pred_att['textures'], target_att['textures'])
    loss_light = 0.1 * dist(pred_att['lights'], target_att['lights'])
    return loss_cam, loss_shape, loss_texture, loss_light, loss_bias


def recon_flip(t, att, L1):
    """networks.py:392-410.  L1=True raises like the reference ((B,V,3) * (B,V), :409)."""
    Na = att['delta_vertices']
    flip = t.flip_index.to(Na.device)
    Nf = Na.index_select(1, flip)
    Nf[..., 2] *= -1
    loss_norm = torch.abs(Na - Nf) if L1 else (Na - Nf).norm(dim=2)
    sign_init = t.sign_init.to(device=Na.device, dtype=Na.dtype)
    mask_a = torch.nn.functional.relu(torch.sign(Na[:, :, 2]) * sign_init)
    mask_f = mask_a.index_select(1, flip)
    return torch.mean(loss_norm * mask_f)


def laplacian_term(t, delta_vertices):
    L = t.vertices_laplacian_matrix.to(device=delta_vertices.device, dtype=delta_vertices.dtype)
    return torch.mean(torch.matmul(L, delta_vertices) ** 2) * delta_vertices.shape[1] * 3


def flat_term(t, face_normals):
    e2f = t.edge2faces.to(face_normals.device)
    cos = torch.sum(face_normals[:, e2f[:, 0]] * face_normals[:, e2f[:, 1]], dim=2)
    return torch.mean((cos - 1) ** 2) * e2f.shape[0]


def calc_reg_loss(t, att):
    """networks.py:412-451."""
    return t.lambda_lpl * laplacian_term(t, att['delta_vertices']) + t.lambda_flat * flat_term(t, att['face_normals'])


def calc_reg_edge(t, pred):
    """networks.py:453-461."""
    edges = t.edges.to(pred.device)
    edge_length = torch.norm(pred[:, edges[:, 0]] - pred[:, edges[:, 1]], p=2, dim=2)
    bias_length = edge_length - torch.mean(edge_length, dim=1, keepdim=True)
    return 0.1 * torch.mean(torch.norm(bias_length, p=2, dim=1))


def calc_reg_depth(t, pred):
    """networks.py:463-466."""
    return torch.mean(pred[:, :, 2] ** 2)


def _depth_weighted(t, pred, w, eps):
    s = t.sign_init.to(pred.device)
    return torch.mean((s >= 0) * (pred[:, :, 2] - eps) ** 2 * w + (s < 0) * (pred[:, :, 2] + eps) ** 2 * w)


def calc_reg_depthR(t, pred, temp=2, eps=0.001):
    """networks.py:468-475."""
    x, y = pred[:, :, 0].detach(), pred[:, :, 1].detach()
    return _depth_weighted(t, pred, torch.exp(temp * (x ** 2 + (y / t.ratio) ** 2)), eps)


def calc_reg_depthC(t, pred, eps=0.001):
    """networks.py:477-485."""
    x, y = pred[:, :, 0].detach(), pred[:, :, 1].detach()
    return _depth_weighted(t, pred, x ** 2 + (y / t.ratio) ** 2, eps)


def calc_reg_deform(t, pred):
    """networks.py:487-491."""
    batchsize = pred.shape[0]
    return torch.mean(torch.norm(pred.reshape(-1, pred.size(2)), p=2, dim=1).reshape(batchsize, -1))
