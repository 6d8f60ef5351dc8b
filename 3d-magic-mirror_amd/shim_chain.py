"""The un-fused compatibility path as ONE function: the operators of the kaolin-shaped import boundary (shim/kaolin over ops.py) called in
the order of the reference's ``DiffRender.render`` (/root/reference/networks.py:278-317) -- what the reference runs if a maintainer only
switches ``sys.path`` to the shim and keeps ``networks.py`` as it is.  bench.py times it (``value_shim``) next to the fused path;
tests/test_gpu_shim_ops.py holds the same chain against the fused render operator by operator.

The camera (smr_utils.py:257-311, pure torch in the reference) is restated with torch ops: spherical angles -> position, look-at
rotation [x y z] with x = normalize(up x z), y = z x x, translation -cam . R.
"""
import math

import torch

from . import ops


def camera_transform(distances, elevations, azimuths, biases):
    """(B,4,3) [R; t] of smr_utils.generate_transformation_matrix(camera_position_from_spherical_angles(d, e, a, degrees=True), [bx,by,0], up=+y)."""
    e, a = elevations * (math.pi / 180.0), azimuths * (math.pi / 180.0)
    cam = torch.stack([distances * torch.cos(e) * torch.sin(a), distances * torch.sin(e), distances * torch.cos(e) * torch.cos(a)], dim=-1)
    at = torch.cat([biases, torch.zeros_like(biases[:, :1])], dim=1)
    up = torch.tensor([[0.0, 1.0, 0.0]], device=cam.device, dtype=cam.dtype).expand_as(cam)
    z = torch.nn.functional.normalize(cam - at, dim=1, eps=1e-5)
    x = torch.nn.functional.normalize(torch.cross(up, z, dim=1), dim=1, eps=1e-5)
    y = torch.cross(z, x, dim=1)
    rot = torch.stack([x, y, z], dim=2)                          # columns
    trans = -torch.bmm(cam.unsqueeze(1), rot)
    return torch.cat([rot, trans], dim=1)


def render(dr, no_mask=False, **attributes):
    """rgbs (B,4,H,W), face_normals, face_idx through the un-fused operators, in the reference's order."""
    dev = attributes["azimuths"].device
    B = attributes["azimuths"].shape[0]
    faces = dr.faces.to(dev)                                     # (the reference re-uploads both on every call, networks.py:272-273)
    face_uvs = dr.face_uvs.to(dev)
    T = camera_transform(attributes["distances"], attributes["elevations"], attributes["azimuths"], attributes["biases"])
    fvc, fvi, fn = ops.prepare_vertices(attributes["vertices"], faces, dr.cam_proj.to(dev), camera_transform=T)
    nrm = ops.face_normals(fvc, unit=True).unsqueeze(-2).repeat(1, 1, 3, 1)
    feats = [torch.ones((B, dr.num_faces, 3, 1), device=dev), face_uvs.repeat(B, 1, 1, 1), nrm]
    (texmask, texcoord, imnormal), soft, fidx = ops.dibr_rasterization(dr.render_height, dr.image_size, fvc[:, :, :, -1], fvi, feats, fn[:, :, -1])
    texcolor = ops.texture_mapping(texcoord, attributes["textures"], mode="bilinear")
    coef = ops.spherical_harmonic_lighting(imnormal, attributes["lights"])
    if no_mask:
        image = (texcolor * texmask + attributes["bg"].permute(0, 2, 3, 1) * (1 - texmask)) * coef.unsqueeze(-1)
    else:
        image = texcolor * texmask * coef.unsqueeze(-1) + torch.ones_like(texcolor) * (1 - texmask)
    rgbs = torch.cat([torch.clamp(image, 0, 1), soft[..., None]], -1).permute(0, 3, 1, 2)
    return rgbs, fn, fidx


def recon_data(dr, pred, gt):
    """networks.py:364-378 (contour = 0) with the shim's mask_iou."""
    gm = gt[:, 3:4]
    l1 = ((pred[:, :3] * gm + (1 - gm)) - (gt[:, :3] * gm + (1 - gm))).abs().mean()
    return dr.image_weight * l1 + ops.mask_iou(pred[:, 3], gt[:, 3])
