"""Host side of the mesh regularisers: static template tables and the autograd wrapper over
``mm_mesh_reg_forward / backward`` (csrc/mm_reg.hip).  Replaces what /root/reference/networks.py:392-491 computes with ~60
small torch launches per attribute set by one launch per direction.  No CPU path: device tensors only."""
import ctypes

import numpy as np
import torch

from . import _native as N

LAPLACIAN, FLAT, EDGE, DEPTH, DEPTHR, DEPTHC, DEFORM, FLIP = range(8)
NTERMS = 8


def _csr(keys, items, n):
    """Rows ``keys`` (ascending after a stable sort) -> (offsets (n+1), items in row order)."""
    order = np.argsort(keys, kind="stable")
    offsets = np.zeros(n + 1, dtype=np.int32)
    np.add.at(offsets, np.asarray(keys, dtype=np.int64) + 1, 1)
    return np.cumsum(offsets).astype(np.int32), np.asarray(items)[order]


def build_tables(dr, device):
    """Static tables of a template, on ``device`` (cached by the caller)."""
    V, F = dr.num_vertices, dr.num_faces
    Lm = dr.vertices_laplacian_matrix.detach().cpu().to(torch.float32).numpy()
    t = {}
    for name, M in (("lap", Lm), ("lapT", Lm.T)):
        r, c = np.nonzero(M)
        off, cols = _csr(r, c.astype(np.int32), V)
        _, vals = _csr(r, M[r, c].astype(np.float32), V)
        t[name + "_offsets"], t[name + "_cols"], t[name + "_vals"] = off, cols.astype(np.int32), vals.astype(np.float32)
    edges = dr.edges.detach().cpu().numpy().astype(np.int32)
    e2f = dr.edge2faces.detach().cpu().numpy().astype(np.int32)
    E = edges.shape[0]
    item = (np.arange(E, dtype=np.int32)[:, None] * 2 + np.arange(2, dtype=np.int32)[None]).reshape(-1)
    t["edges"], t["edge2faces"] = edges, e2f
    t["ve_offsets"], t["ve_items"] = _csr(edges.reshape(-1), item, V)
    t["fe_offsets"], t["fe_items"] = _csr(e2f.reshape(-1), item, F)
    flip = dr.flip_index.detach().cpu().numpy().astype(np.int32)
    t["flip_index"] = flip
    t["flipT_offsets"], t["flipT_items"] = _csr(flip, np.arange(V, dtype=np.int32), V)
    t["sign_init"] = dr.sign_init.detach().cpu().to(torch.float32).numpy()
    out = {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in t.items()}
    out["E"] = E
    return out


def _desc(dr, tab, terms, temp, eps, vertices, delta, fn, losses, ws):
    d = N.MMMeshRegDesc()
    ref = vertices if vertices is not None else (delta if delta is not None else fn)
    d.B, d.V, d.F, d.E = ref.shape[0], dr.num_vertices, dr.num_faces, tab["E"]
    d.terms = terms
    for k in ("lap_offsets", "lap_cols", "lap_vals", "lapT_offsets", "lapT_cols", "lapT_vals", "edges", "edge2faces", "ve_offsets",
              "ve_items", "fe_offsets", "fe_items", "flip_index", "flipT_offsets", "flipT_items", "sign_init"):
        setattr(d, k, N.ptr(tab[k]))
    d.vertices, d.delta_vertices, d.face_normals = N.ptr(vertices), N.ptr(delta), N.ptr(fn)
    d.ratio, d.temp, d.eps = float(dr.ratio), float(temp), float(eps)
    d.losses = N.ptr(losses)
    if ws is not None:
        d.workspace, d.workspace_bytes = N.ptr(ws), ws.numel()
    return d


class MeshRegFn(torch.autograd.Function):
    """losses (8,) = mesh_reg(terms; vertices, delta_vertices, face_normals); unrequested terms are 0."""

    @staticmethod
    def forward(ctx, dr, terms, temp, eps, vertices, delta, fn):
        given = [t for t in (vertices, delta, fn) if t is not None]
        N.require_device(*given)
        dev = given[0].device
        f32 = lambda t: N.as_f32(t, dev)
        vertices, delta, fn = f32(vertices), f32(delta), f32(fn)
        B = given[0].shape[0]
        for t, n in ((vertices, dr.num_vertices), (delta, dr.num_vertices), (fn, dr.num_faces)):
            if t is not None and tuple(t.shape) != (B, n, 3):
                raise RuntimeError("mesh regulariser input must be (%d,%d,3), got %s" % (B, n, tuple(t.shape)))
        tab = dr._reg_tables(dev)
        losses = torch.empty(NTERMS, device=dev, dtype=torch.float32)
        d = _desc(dr, tab, terms, temp, eps, vertices, delta, fn, losses, None)
        ws = torch.zeros(N.lib().mm_mesh_reg_query_workspace(ctypes.byref(d)), device=dev, dtype=torch.uint8)   # zero-filled: ABI contract
        d.workspace, d.workspace_bytes = N.ptr(ws), ws.numel()
        N.check(N.lib().mm_mesh_reg_forward(ctypes.byref(d), N.current_stream(dev)), "mm_mesh_reg_forward")
        ctx.dr, ctx.cfg = dr, (terms, temp, eps)
        ctx.has = (vertices is not None, delta is not None, fn is not None)
        ctx.save_for_backward(*[t if t is not None else torch.empty(0, device=dev) for t in (vertices, delta, fn)], ws)
        return losses

    @staticmethod
    def backward(ctx, g):
        vertices, delta, fn, ws = ctx.saved_tensors
        vertices, delta, fn = [t if h else None for t, h in zip((vertices, delta, fn), ctx.has)]
        dr = ctx.dr
        terms, temp, eps = ctx.cfg
        dev = ws.device
        tab = dr._reg_tables(dev)
        d = _desc(dr, tab, terms, temp, eps, vertices, delta, fn, None, ws)
        w = g.detach().to(device=dev, dtype=torch.float32).contiguous()
        need = ctx.needs_input_grad[4:7]
        gv = torch.empty_like(vertices) if vertices is not None and need[0] else None
        gd = torch.empty_like(delta) if delta is not None and need[1] else None
        gf = torch.empty_like(fn) if fn is not None and need[2] else None
        if gv is None and gd is None and gf is None:
            return (None,) * 7
        gr = N.MMMeshRegGrads(N.ptr(w), N.ptr(gv), N.ptr(gd), N.ptr(gf))
        N.check(N.lib().mm_mesh_reg_backward(ctypes.byref(d), ctypes.byref(gr), N.current_stream(dev)), "mm_mesh_reg_backward")
        return None, None, None, None, gv, gd, gf


def mask(*terms):
    m = 0
    for t in terms:
        m |= 1 << t
    return m
