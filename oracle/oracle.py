"""ctypes binding of the CPU oracle (oracle/libmm_oracle.so).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never
from the product package.  PARITY UNPINNED at the kaolin boundary (see mm_oracle.inc header).
All arrays are numpy, C-contiguous; ``dtype`` selects the float32 (parity / baseline) or float64 (finite-difference)
instantiation.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libmm_oracle.so")
    srcs = [os.path.join(_HERE, n) for n in ("mm_oracle.c", "mm_oracle.inc", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libmm_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.mmo_num_threads.restype = ctypes.c_int
        for sfx, ct in (("f32", ctypes.c_float), ("f64", ctypes.c_double)):
            getattr(_LIB, "mmo_recon_data_" + sfx).restype = ct
    return _LIB


OPT_CULL_STRICT, OPT_SOFT_SKIP_CULLED, OPT_BBOX_HALF_OPEN, OPT_BARY_ONE_MINUS, OPT_SH_ORDER_XYZ = 1 << 4, 1 << 5, 1 << 6, 1 << 7, 1 << 8
OPT_BBOX_MIN_CLOSED_MAX_OPEN = 1 << 9


class options(object):
    """``with oracle.options(bits): ...`` -- the SURVEY Appendix C switches (include/mm_render.h MM_OPT_*), mirrored by the HIP path."""

    def __init__(self, bits):
        self.bits = int(bits)

    def __enter__(self):
        self.old = lib().mmo_get_options()
        lib().mmo_set_options(self.bits)
        return self

    def __exit__(self, *exc):
        lib().mmo_set_options(self.old)
        return False


def num_threads():
    return lib().mmo_num_threads()


def set_threads(n):
    lib().mmo_set_threads(int(n))


def _sfx(dtype):
    return "f32" if np.dtype(dtype) == np.float32 else "f64"


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


def _real(dtype):
    return ctypes.c_float if np.dtype(dtype) == np.float32 else ctypes.c_double


class _Cfg32(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("V", ctypes.c_int), ("F", ctypes.c_int),
                ("Ht", ctypes.c_int), ("Wt", ctypes.c_int), ("no_mask", ctypes.c_int), ("knum", ctypes.c_int),
                ("proj", ctypes.c_float * 3), ("sigmainv", ctypes.c_float), ("boxlen", ctypes.c_float),
                ("mult", ctypes.c_float), ("eps", ctypes.c_float)]


class _Cfg64(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("V", ctypes.c_int), ("F", ctypes.c_int),
                ("Ht", ctypes.c_int), ("Wt", ctypes.c_int), ("no_mask", ctypes.c_int), ("knum", ctypes.c_int),
                ("proj", ctypes.c_double * 3), ("sigmainv", ctypes.c_double), ("boxlen", ctypes.c_double),
                ("mult", ctypes.c_double), ("eps", ctypes.c_double)]


def camera(dist, elev, azim, bias, dtype=np.float32):
    B = len(dist)
    T = np.zeros((B, 4, 3), dtype)
    getattr(lib(), "mmo_camera_" + _sfx(dtype))(B, _p(_c(dist, dtype)), _p(_c(elev, dtype)), _p(_c(azim, dtype)),
                                                 _p(_c(bias, dtype)), _p(T))
    return T


def camera_backward(dist, elev, azim, bias, dT, dtype=np.float32):
    B = len(dist)
    dd, de, da, db = np.zeros(B, dtype), np.zeros(B, dtype), np.zeros(B, dtype), np.zeros((B, 2), dtype)
    getattr(lib(), "mmo_camera_backward_" + _sfx(dtype))(B, _p(_c(dist, dtype)), _p(_c(elev, dtype)), _p(_c(azim, dtype)),
                                                          _p(_c(bias, dtype)), _p(_c(dT, dtype)), _p(dd), _p(de), _p(da), _p(db))
    return dd, de, da, db


def prepare_vertices(vertices, faces, T, proj, dtype=np.float32):
    vertices = _c(vertices, dtype); faces = _c(faces, np.int32); T = _c(T, dtype); proj = _c(np.asarray(proj).reshape(3), dtype)
    B, V, _ = vertices.shape; F = faces.shape[0]
    fvc, fvi, fn = np.zeros((B, F, 3, 3), dtype), np.zeros((B, F, 3, 2), dtype), np.zeros((B, F, 3), dtype)
    getattr(lib(), "mmo_prepare_vertices_" + _sfx(dtype))(B, V, F, _p(vertices), _p(faces), _p(T), _p(proj), _p(fvc), _p(fvi), _p(fn))
    return fvc, fvi, fn


def prepare_vertices_backward(vertices, faces, T, proj, dfvc, dfvi, dfn, dtype=np.float32):
    vertices = _c(vertices, dtype); faces = _c(faces, np.int32); T = _c(T, dtype); proj = _c(np.asarray(proj).reshape(3), dtype)
    B, V, _ = vertices.shape; F = faces.shape[0]
    dv, dT = np.zeros((B, V, 3), dtype), np.zeros((B, 4, 3), dtype)
    getattr(lib(), "mmo_prepare_vertices_backward_" + _sfx(dtype))(B, V, F, _p(vertices), _p(faces), _p(T), _p(proj),
                                                                    _p(_c(dfvc, dtype)), _p(_c(dfvi, dtype)), _p(_c(dfn, dtype)), _p(dv), _p(dT))
    return dv, dT


def rasterize(H, W, fz, fvi, feats, valid, mult=1000.0, eps=1e-8, dtype=np.float32):
    fz = _c(fz, dtype); fvi = _c(fvi, dtype); feats = _c(feats, dtype); valid = _c(valid, np.uint8)
    B, F, _ = fz.shape; D = feats.shape[-1]
    fidx = np.zeros((B, H, W), np.int32); w = np.zeros((B, H, W, 3), dtype); out = np.zeros((B, H, W, D), dtype)
    r = _real(dtype)
    getattr(lib(), "mmo_rasterize_" + _sfx(dtype))(B, H, W, F, D, _p(fz), _p(fvi), _p(feats), _p(valid), r(mult), r(eps), _p(fidx), _p(w), _p(out))
    return fidx, w, out


def rasterize_backward(dinterp, face_idx, fvi, feats, mult=1000.0, eps=1e-8, dtype=np.float32):
    dinterp = _c(dinterp, dtype); face_idx = _c(face_idx, np.int32); fvi = _c(fvi, dtype); feats = _c(feats, dtype)
    B, H, W, D = dinterp.shape; F = fvi.shape[1]
    dfvi, dfeats = np.zeros((B, F, 3, 2), dtype), np.zeros((B, F, 3, D), dtype)
    r = _real(dtype)
    getattr(lib(), "mmo_rasterize_backward_" + _sfx(dtype))(B, H, W, F, D, _p(dinterp), _p(face_idx), _p(fvi), _p(feats), r(mult), r(eps), _p(dfvi), _p(dfeats))
    return dfvi, dfeats


def soft_mask(H, W, fvi, face_idx, sigmainv=7000.0, boxlen=0.02, knum=30, mult=1000.0, dtype=np.float32, aux=True, valid=None):
    """``valid`` (B,F) uint8: the faces the colour pass rasterises; only read under OPT_SOFT_SKIP_CULLED."""
    fvi = _c(fvi, dtype); face_idx = _c(face_idx, np.int32)
    valid = None if valid is None else _c(valid, np.uint8)
    lib().mmo_set_soft_valid(_p(valid))
    B, F = fvi.shape[:2]
    soft = np.zeros((B, H, W), dtype)
    prob = np.zeros((B, H, W, knum), dtype) if aux else None
    idx = np.zeros((B, H, W, knum), np.int32) if aux else None
    typ = np.zeros((B, H, W, knum), np.uint8) if aux else None
    r = _real(dtype)
    getattr(lib(), "mmo_soft_mask_" + _sfx(dtype))(B, H, W, F, _p(fvi), _p(face_idx), r(sigmainv), r(boxlen), knum, r(mult), _p(soft), _p(prob), _p(idx), _p(typ))
    lib().mmo_set_soft_valid(None)
    return soft, prob, idx, typ


def soft_mask_backward(dsoft, face_idx, fvi, prob, idx, typ, sigmainv=7000.0, mult=1000.0, dtype=np.float32):
    dsoft = _c(dsoft, dtype); fvi = _c(fvi, dtype); face_idx = _c(face_idx, np.int32)
    B, H, W = dsoft.shape; F = fvi.shape[1]; knum = prob.shape[-1]
    dfvi = np.zeros((B, F, 3, 2), dtype)
    r = _real(dtype)
    getattr(lib(), "mmo_soft_mask_backward_" + _sfx(dtype))(B, H, W, F, _p(dsoft), _p(face_idx), _p(fvi), _p(_c(prob, dtype)), _p(_c(idx, np.int32)),
                                                             _p(_c(typ, np.uint8)), r(sigmainv), knum, r(mult), _p(dfvi))
    return dfvi


def texture_mapping(uv, tex, dtype=np.float32):
    uv = _c(uv, dtype); tex = _c(tex, dtype)
    B, N, _ = uv.shape; C, Ht, Wt = tex.shape[1:]
    out = np.zeros((B, N, C), dtype)
    getattr(lib(), "mmo_texture_mapping_" + _sfx(dtype))(B, N, C, Ht, Wt, _p(uv), _p(tex), _p(out))
    return out


def texture_mapping_backward(uv, tex, dout, dtype=np.float32):
    uv = _c(uv, dtype); tex = _c(tex, dtype); dout = _c(dout, dtype)
    B, N, _ = uv.shape; C, Ht, Wt = tex.shape[1:]
    duv, dtex = np.zeros((B, N, 2), dtype), np.zeros_like(tex)
    getattr(lib(), "mmo_texture_mapping_backward_" + _sfx(dtype))(B, N, C, Ht, Wt, _p(uv), _p(tex), _p(dout), _p(duv), _p(dtex))
    return duv, dtex


def sh_lighting(nrm, lights, dtype=np.float32):
    nrm = _c(nrm, dtype); lights = _c(lights, dtype)
    B, N, _ = nrm.shape
    coef = np.zeros((B, N), dtype)
    getattr(lib(), "mmo_sh_lighting_" + _sfx(dtype))(B, N, _p(nrm), _p(lights), _p(coef))
    return coef


def sh_lighting_backward(nrm, lights, dcoef, dtype=np.float32):
    nrm = _c(nrm, dtype); lights = _c(lights, dtype); dcoef = _c(dcoef, dtype)
    B, N, _ = nrm.shape
    dn, dl = np.zeros((B, N, 3), dtype), np.zeros((B, 9), dtype)
    getattr(lib(), "mmo_sh_lighting_backward_" + _sfx(dtype))(B, N, _p(nrm), _p(lights), _p(dcoef), _p(dn), _p(dl))
    return dn, dl


def _cfg(inp, H, W, no_mask, proj, dtype, sigmainv=7000.0, boxlen=0.02, knum=30, mult=1000.0, eps=1e-8):
    cls = _Cfg32 if np.dtype(dtype) == np.float32 else _Cfg64
    c = cls()
    c.B, c.V = inp["vertices"].shape[:2]
    c.H, c.W, c.F = H, W, inp["faces"].shape[0]
    c.Ht, c.Wt = inp["textures"].shape[2:]
    c.no_mask, c.knum = int(bool(no_mask)), knum
    for i in range(3):
        c.proj[i] = float(np.asarray(proj).reshape(3)[i])
    c.sigmainv, c.boxlen, c.mult, c.eps = sigmainv, boxlen, mult, eps
    return c


_KEYS = ("vertices", "faces", "face_uvs", "textures", "lights", "bg", "azimuths", "elevations", "distances", "biases")


def _inputs(inp, dtype):
    a = {k: (None if inp.get(k) is None else _c(inp[k], np.int32 if k == "faces" else dtype)) for k in _KEYS}
    a["face_uvs"] = a["face_uvs"].reshape(-1, 3, 2)
    return a


def render_forward(inp, H, W, no_mask, proj, dtype=np.float32, **kw):
    """inp: dict with the keys of _KEYS (numpy).  Returns rgba (B,H,W,4), face_idx (B,H,W) int32, face_normals, imnormal."""
    a = _inputs(inp, dtype); c = _cfg(a, H, W, no_mask, proj, dtype, **kw)
    B, F = c.B, c.F
    rgba = np.zeros((B, H, W, 4), dtype); fidx = np.zeros((B, H, W), np.int32)
    fn = np.zeros((B, F, 3), dtype); imn = np.zeros((B, H, W, 3), dtype)
    getattr(lib(), "mmo_render_forward_" + _sfx(dtype))(ctypes.byref(c), _p(a["vertices"]), _p(a["faces"]), _p(a["face_uvs"]),
        _p(a["textures"]), _p(a["lights"]), _p(a["bg"]), _p(a["azimuths"]), _p(a["elevations"]), _p(a["distances"]), _p(a["biases"]),
        _p(rgba), _p(fidx), _p(fn), _p(imn))
    return rgba, fidx, fn, imn


def render_backward(inp, H, W, no_mask, proj, drgba, dface_normals=None, dtype=np.float32, **kw):
    a = _inputs(inp, dtype); c = _cfg(a, H, W, no_mask, proj, dtype, **kw)
    B = c.B
    g = {"vertices": np.zeros_like(a["vertices"]), "textures": np.zeros_like(a["textures"]), "lights": np.zeros((B, 9), dtype),
         "bg": None if a["bg"] is None else np.zeros_like(a["bg"]), "azimuths": np.zeros(B, dtype), "elevations": np.zeros(B, dtype),
         "distances": np.zeros(B, dtype), "biases": np.zeros((B, 2), dtype)}
    getattr(lib(), "mmo_render_backward_" + _sfx(dtype))(ctypes.byref(c), _p(a["vertices"]), _p(a["faces"]), _p(a["face_uvs"]),
        _p(a["textures"]), _p(a["lights"]), _p(a["bg"]), _p(a["azimuths"]), _p(a["elevations"]), _p(a["distances"]), _p(a["biases"]),
        _p(_c(drgba, dtype)), _p(_c(dface_normals, dtype)), _p(g["vertices"]), _p(g["textures"]), _p(g["lights"]), _p(g["bg"]),
        _p(g["azimuths"]), _p(g["elevations"]), _p(g["distances"]), _p(g["biases"]))
    return g


def recon_data(pred, gt, image_weight=0.1, contour=0.0, want_grad=False, gscale=1.0, dtype=np.float32):
    """pred: (B,4,H,W) numpy array of ANY strides (e.g. a transposed view of NHWC storage); gt (B,4,H,W).
    Returns loss, or (loss, dpred) with dpred laid out like pred."""
    assert pred.dtype == np.dtype(dtype)
    gt = _c(gt, dtype)
    B, _, H, W = pred.shape
    item = pred.dtype.itemsize
    strides = (ctypes.c_int64 * 4)(*[s // item for s in pred.strides])
    dpred = None
    if want_grad:
        dpred = np.lib.stride_tricks.as_strided(np.zeros(pred.size, dtype), pred.shape, pred.strides)
        # as_strided over a fresh buffer only works for permutations of a dense layout; check
        assert sum(s // item * (n - 1) for s, n in zip(pred.strides, pred.shape)) < pred.size
    r = _real(dtype)
    loss = getattr(lib(), "mmo_recon_data_" + _sfx(dtype))(B, H, W, _p(pred), strides, _p(gt), r(image_weight), r(contour),
                                                           _p(dpred), r(gscale))
    return (float(loss), dpred) if want_grad else float(loss)


def step(inp, gt, H, W, no_mask, proj, image_weight=0.1, dtype=np.float32, **kw):
    """render -> recon_data -> backward in one call (the unit the CPU baseline times).  Returns (loss, grads dict)."""
    a = _inputs(inp, dtype); c = _cfg(a, H, W, no_mask, proj, dtype, **kw)
    B = c.B
    g = {"vertices": np.zeros_like(a["vertices"]), "textures": np.zeros_like(a["textures"]), "lights": np.zeros((B, 9), dtype),
         "bg": None if a["bg"] is None else np.zeros_like(a["bg"]), "azimuths": np.zeros(B, dtype), "elevations": np.zeros(B, dtype),
         "distances": np.zeros(B, dtype), "biases": np.zeros((B, 2), dtype)}
    fn = getattr(lib(), "mmo_step_" + _sfx(dtype))
    fn.restype = _real(dtype)
    r = _real(dtype)
    loss = fn(ctypes.byref(c), _p(a["vertices"]), _p(a["faces"]), _p(a["face_uvs"]), _p(a["textures"]), _p(a["lights"]), _p(a["bg"]),
              _p(a["azimuths"]), _p(a["elevations"]), _p(a["distances"]), _p(a["biases"]), _p(_c(gt, dtype)), r(image_weight),
              _p(g["vertices"]), _p(g["textures"]), _p(g["lights"]), _p(g["bg"]), _p(g["azimuths"]), _p(g["elevations"]),
              _p(g["distances"]), _p(g["biases"]))
    return float(loss), g
