"""Wavefront OBJ input/output for the template meshes either side of the render path.

Mirrors the fields the reference reads from ``kaolin.io.obj.import_mesh(path, with_materials=True)``
(call site /root/reference/networks.py:176, fields used at :181,:196-200) and the writer
``smr_utils.save_mesh`` (/root/reference/smr_utils.py:188-196).  Semantics restated in SURVEY.md Appendix B.
"""
from collections import namedtuple

import numpy as np
import torch

ObjMesh = namedtuple("ObjMesh", ["vertices", "faces", "uvs", "face_uvs_idx", "vertex_normals", "face_normals_idx"])


def _parse_obj_text(lines):
    v, vt, vn = [], [], []
    fv, ft, fnm = [], [], []
    for raw in lines:
        line = raw.strip()
        if not line or line[0] == "#":
            continue
        tok = line.split()
        key = tok[0]
        if key == "v":
            v.append((float(tok[1]), float(tok[2]), float(tok[3])))
        elif key == "vt":
            vt.append((float(tok[1]), float(tok[2])))
        elif key == "vn":
            vn.append((float(tok[1]), float(tok[2]), float(tok[3])))
        elif key == "f":
            corners = tok[1:]
            if len(corners) != 3:
                raise ValueError("only triangle meshes are supported (face with %d corners)" % len(corners))
            iv, it, inn = [], [], []
            for c in corners:
                parts = c.split("/")
                iv.append(int(parts[0]))
                it.append(int(parts[1]) if len(parts) > 1 and parts[1] != "" else 0)
                inn.append(int(parts[2]) if len(parts) > 2 and parts[2] != "" else 0)
            fv.append(iv); ft.append(it); fnm.append(inn)
        # mtllib / usemtl / o / g / s: materials carry no texture map in any template; ignored
    return v, vt, vn, fv, ft, fnm


def _rebase(idx, count):
    """OBJ indices are 1-based; negative indices are relative to the end."""
    a = np.asarray(idx, dtype=np.int64).reshape(-1, 3)
    if a.size == 0:
        return a
    a = np.where(a < 0, a + count + 1, a)
    return a - 1


def import_mesh(path, with_materials=True):
    """Read a triangle OBJ.  Returns an ``ObjMesh`` of torch tensors:

    vertices (V,3) float32, faces (F,3) int64 0-based, uvs (Nvt,2) float32 raw (OBJ convention, v=0 bottom row),
    face_uvs_idx (F,3) int64 0-based (-1 where a corner has no ``vt``).  ``with_materials`` is accepted for
    signature compatibility with the reference call and does not change what is returned.
    """
    with open(path, "r") as fp:
        v, vt, vn, fv, ft, fnm = _parse_obj_text(fp)
    vertices = torch.tensor(np.asarray(v, dtype=np.float64).reshape(-1, 3), dtype=torch.float32)
    uvs = torch.tensor(np.asarray(vt, dtype=np.float64).reshape(-1, 2), dtype=torch.float32)
    normals = torch.tensor(np.asarray(vn, dtype=np.float64).reshape(-1, 3), dtype=torch.float32)
    faces = torch.from_numpy(_rebase(fv, len(v)))
    face_uvs_idx = torch.from_numpy(_rebase(ft, len(vt)))
    face_nrm_idx = torch.from_numpy(_rebase(fnm, len(vn)))
    if faces.numel() and (faces.min() < 0 or faces.max() >= len(v)):
        raise ValueError("face index out of range in %s" % path)
    return ObjMesh(vertices, faces, uvs, face_uvs_idx, normals, face_nrm_idx)


def save_mesh(obj_mesh_name, v, faces, vt=None):
    """Write ``v`` lines, then ``vt`` lines, then 1-based ``f a b c`` (no uv indices), %f formatting."""
    v = torch.as_tensor(v).detach().cpu().numpy()
    faces = torch.as_tensor(faces).detach().cpu().numpy()
    out = ["v %f %f %f\n" % (p[0], p[1], p[2]) for p in v]
    if vt is not None:
        vt = torch.as_tensor(vt).detach().cpu().numpy()
        out += ["vt %f %f\n" % (t[0], t[1]) for t in vt]
    out += ["f %d %d %d\n" % (f[0] + 1, f[1] + 1, f[2] + 1) for f in faces]
    with open(obj_mesh_name, "w") as fp:
        fp.writelines(out)


def load_template(path):
    """Template loader used by DiffRender: ``.obj`` through :func:`import_mesh`, ``.npz`` (the committed fixtures of
    the reference's template meshes, tests/golden/templates/) through numpy."""
    if str(path).endswith(".npz"):
        z = np.load(path)
        e3 = torch.zeros((0, 3))
        return ObjMesh(torch.from_numpy(z["vertices"].astype(np.float32)), torch.from_numpy(z["faces"].astype(np.int64)),
                       torch.from_numpy(z["uvs"].astype(np.float32)), torch.from_numpy(z["face_uvs_idx"].astype(np.int64)),
                       e3, torch.zeros((0, 3), dtype=torch.int64))
    return import_mesh(path, with_materials=True)
