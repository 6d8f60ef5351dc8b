"""The kaolin / pytorch3d operators the reference imports (/root/reference/networks.py:6-19, trainer.py:31-40) as stand-alone
differentiable functions over the gfx950 C ABI.  Names, arguments, defaults, return values and error
behaviour follow kaolin v0.12.0 / pytorch3d 0.7 as restated in SURVEY.md 8(a)/8(b); ``3d-magic-mirror_amd/shim`` re-exports them
under kaolin's and pytorch3d's own module paths so that ``networks.py`` / ``trainer.py`` import and run unmodified.

Every function that touches per-sample data runs a hand-written HIP kernel through ``lib/libmm_render.so`` (csrc/mm_ops.hip,
csrc/mm_dibr.hip) and raises for tensors that are not in device memory -- there is no CPU or eager-torch fallback.  torch is used
for device memory, the current stream and autograd plumbing (``cat`` / ``split`` of feature lists, views).  Host-side,
one-time template preparation (``index_vertices_by_faces`` on the static uv table, ``uniform_laplacian``, ``import_mesh``)
is torch/numpy indexing exactly as in kaolin itself.
"""
import ctypes

import numpy as np
import torch

from . import _native as N
from . import obj_io, template
from .chamfer import chamfer_distance as _chamfer

# ---- kaolin.render.camera ---------------------------------------------------------------------------------------------
generate_perspective_projection = template.generate_perspective_projection

# ---- kaolin.io.obj -----------------------------------------------------------------------------------------------------
import_mesh = obj_io.import_mesh

# ---- kaolin.ops.mesh (host-side template preparation) ------------------------------------------------------------------
index_vertices_by_faces = template.index_vertices_by_faces
uniform_laplacian = template.uniform_laplacian


def _f32(t, dev):
    return N.as_f32(t, dev)


_STATUS = {}              # per DEVICE: one pinned int32 the device adds vertex ids outside the cloud to (mm_build_vertex_corner_csr_device); polled, never waited for
_CSR_MAX_V = 12288        # MM_CSR_MAX_V (csrc/mm_ops.hip)


def _status_word(dev):
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _STATUS:
        _STATUS[key] = torch.zeros(1, dtype=torch.int32).pin_memory()
    return _STATUS[key]


def poll_reported_faces(device=None, synchronize=False):
    """DEFERRED ERROR of ``prepare_vertices``.  The call itself never waits for the device, so it cannot validate ``faces`` against V on the spot:
    the device-side CSR builder counts vertex ids outside [0, V) into a pinned status word (one per device) -- such ids are clamped in the forward
    and left out of the gradient gather, never dereferenced -- and the RuntimeError is raised by whoever looks at the word next: the next
    ``prepare_vertices`` on that device, the backward of the offending call, or this function.  ``synchronize=True`` waits for the device first,
    i.e. reports everything enqueued so far (use it once after set-up, or in tests); without it the poll costs nothing.  Returns None if clean."""
    devs = list(_STATUS) if device is None else [(torch.device(device).type, torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device())]
    for key in devs:
        w = _STATUS.get(key)
        if w is None:
            continue
        if synchronize:
            torch.cuda.synchronize(key[1])
        n = int(w[0])
        if n != 0:
            w[0] = 0
            recent = ", ".join("#%d (V=%d, F=%d)" % c for c in _RECENT.get(key, []))
            _RECENT[key] = []
            raise RuntimeError("prepare_vertices: faces held %d vertex ids outside [0, V) (device %s:%d).  The report is DEFERRED (no call waits for the "
                               "device): it surfaces in whichever prepare_vertices call, backward or poll on this device looks next -- possibly not the "
                               "offender's own.  The offender is one of the calls enqueued there since the last clean look: %s" % (n, key[0], key[1], recent or "(none on record)"))
        elif synchronize:
            _RECENT[key] = []                                    # everything enqueued so far has run and was clean
    return None


_RECENT = {}         # device key -> [(serial, V, F)] of the prepare_vertices calls since the last clean look (at most 16 kept): named in the deferred error
_SERIAL = [0]


def _note_call(dev, V, F):
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    _SERIAL[0] += 1
    r = _RECENT.setdefault(key, [])
    r.append((_SERIAL[0], int(V), int(F)))
    del r[:-16]


def _raise_on_reported_faces(dev):
    """An EARLIER prepare_vertices on this device saw vertex ids outside its cloud (poll_reported_faces)."""
    poll_reported_faces(dev)


def _faces_tables(faces, V, dev):
    """int32 device copy of ``faces`` and its vertex->corner CSR (needed by the gather-formulated backward), built ON THE DEVICE by one small
    launch, every call.  The reference re-creates the tensor on every render (``faces = self.faces.to(device)``, networks.py:272): a cache
    keyed by address goes stale between two templates of one shape (seen once as a wrong-topology render), and validating an entry against
    the tensor's CONTENTS is a device -> host read, i.e. a synchronisation of the whole stream in the middle of every render (it made this
    path host-bound at 2.2 ms per step).  No cache, no read-back: nothing to go stale, nothing to wait for."""
    fi = faces.detach().to(device=dev, dtype=torch.int32).contiguous()
    F = fi.shape[0]
    off = torch.empty(V + 1, device=dev, dtype=torch.int32)
    items = torch.empty(3 * F, device=dev, dtype=torch.int32)
    if V <= _CSR_MAX_V:
        N.check(N.lib().mm_build_vertex_corner_csr_device(V, F, N.ptr(fi), N.ptr(off), N.ptr(items), _status_word(dev).data_ptr(), N.current_stream(dev)),
                "mm_build_vertex_corner_csr_device")
    else:                                                        # (huge clouds: the host builder, with its read-back)
        fh = faces.detach().to("cpu", torch.int64)
        if fh.numel() and (int(fh.max()) >= V or int(fh.min()) < 0):
            raise RuntimeError("faces index vertex %d but vertices has %d" % (int(fh.max()), V))
        o, it = template.vertex_corner_adjacency(V, fh)
        off.copy_(o.to(torch.int32)); items.copy_(it.to(torch.int32))
    return fi, off, items


# ---- kaolin.render.mesh.prepare_vertices -------------------------------------------------------------------------------
class _PrepareFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vertices, transform, faces_i32, vc_off, vc_items, proj):
        dev = vertices.device
        vertices, transform = _f32(vertices, dev), _f32(transform, dev)
        B, V, _ = vertices.shape
        F = faces_i32.shape[0]
        fvc = torch.empty((B, F, 3, 3), device=dev, dtype=torch.float32)
        fvi = torch.empty((B, F, 3, 2), device=dev, dtype=torch.float32)
        fn = torch.empty((B, F, 3), device=dev, dtype=torch.float32)
        d = N.MMPrepareDesc()
        d.B, d.V, d.F = B, V, F
        d.proj_device = N.ptr(proj)
        d.faces, d.vc_offsets, d.vc_items = N.ptr(faces_i32), N.ptr(vc_off), N.ptr(vc_items)
        d.vertices, d.transform = N.ptr(vertices), N.ptr(transform)
        d.face_vertices_camera, d.face_vertices_image, d.face_normals = N.ptr(fvc), N.ptr(fvi), N.ptr(fn)
        N.check(N.lib().mm_prepare_vertices_forward(ctypes.byref(d), N.current_stream(dev)), "mm_prepare_vertices_forward")
        ctx.save_for_backward(vertices, transform, faces_i32, vc_off, vc_items, proj)
        return fvc, fvi, fn

    @staticmethod
    def backward(ctx, g_fvc, g_fvi, g_fn):
        vertices, transform, faces_i32, vc_off, vc_items, proj = ctx.saved_tensors
        dev = vertices.device
        _raise_on_reported_faces(dev)                            # the forward's CSR builder has long finished by the time a backward runs on the host
        B, V, _ = vertices.shape
        c = lambda g: None if g is None else g.to(torch.float32).contiguous()
        g_fvc, g_fvi, g_fn = c(g_fvc), c(g_fvi), c(g_fn)
        d = N.MMPrepareDesc()
        d.B, d.V, d.F = B, V, faces_i32.shape[0]
        d.proj_device = N.ptr(proj)
        d.faces, d.vc_offsets, d.vc_items = N.ptr(faces_i32), N.ptr(vc_off), N.ptr(vc_items)
        d.vertices, d.transform = N.ptr(vertices), N.ptr(transform)
        gv = torch.empty_like(vertices)
        gT = torch.empty_like(transform) if ctx.needs_input_grad[1] else None
        ws = torch.empty(N.lib().mm_prepare_vertices_query_workspace(ctypes.byref(d)), device=dev, dtype=torch.uint8)
        d.workspace, d.workspace_bytes = N.ptr(ws), ws.numel()
        g = N.MMPrepareGrads(N.ptr(g_fvc), N.ptr(g_fvi), N.ptr(g_fn), N.ptr(gv), N.ptr(gT))
        N.check(N.lib().mm_prepare_vertices_backward(ctypes.byref(d), ctypes.byref(g), N.current_stream(dev)), "mm_prepare_vertices_backward")
        return gv, gT, None, None, None, None


def prepare_vertices(vertices, faces, camera_proj, camera_rot=None, camera_trans=None, camera_transform=None):
    """kaolin.render.mesh.prepare_vertices: (face_vertices_camera (B,F,3,3), face_vertices_image (B,F,3,2), face_normals (B,F,3)).

    ``camera_transform`` (B,4,3) is what the reference passes (networks.py:284-287).  With ``camera_rot`` (B,3,3) /
    ``camera_trans`` (B,3) instead, kaolin's ``rotate_translate_points`` ((p - t) @ R^T) is folded into the same transform.
    NOTE -- deferred error: vertex ids of ``faces`` outside [0, V) are reported by a LATER look at the device's status word (the next
    ``prepare_vertices``, a backward -- possibly ANOTHER call's, from the autograd thread -- or ``poll_reported_faces``), not by this call; the
    message names the calls enqueued since the last clean look (serial number, V, F) so that the offender can be told apart."""
    N.require_device(vertices)
    dev = vertices.device
    if camera_transform is None:
        if camera_rot is None or camera_trans is None:
            raise AssertionError("camera_transform or camera_trans and camera_rot must be defined")
        rt = camera_rot.to(dev).transpose(1, 2)
        camera_transform = torch.cat([rt, -(camera_trans.to(dev).reshape(-1, 1, 3) @ rt)], dim=1)
    if camera_transform.dim() != 3 or camera_transform.shape[1:] != (4, 3) or camera_transform.shape[0] != vertices.shape[0]:
        raise RuntimeError("camera_transform must be (B,4,3), got %s" % (tuple(camera_transform.shape),))
    if faces.dim() != 2 or faces.shape[1] != 3:
        raise RuntimeError("faces must be (F,3), got %s" % (tuple(faces.shape),))
    _raise_on_reported_faces(dev)                                # (deferred: see poll_reported_faces)
    _note_call(dev, vertices.shape[1], faces.shape[0])
    faces_i32, off, items = _faces_tables(faces, int(vertices.shape[1]), dev)
    proj = camera_proj.detach().to(device=dev, dtype=torch.float32).reshape(-1).contiguous()       # stays on the device: read by the kernels
    if proj.numel() != 3:
        raise RuntimeError("camera_proj must have 3 entries")
    return _PrepareFn.apply(vertices, camera_transform.to(dev), faces_i32, off, items, proj)


# ---- kaolin.ops.mesh.face_normals --------------------------------------------------------------------------------------
class _FaceNormalsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fv, unit):
        dev = fv.device
        fv = _f32(fv, dev)
        out = torch.empty(fv.shape[:-2] + (3,), device=dev, dtype=torch.float32)
        n = fv.numel() // 9
        N.check(N.lib().mm_face_normals_forward(n, int(unit), N.ptr(fv), N.ptr(out), N.current_stream(dev)), "mm_face_normals_forward")
        ctx.save_for_backward(fv)
        ctx.unit = int(unit)
        return out

    @staticmethod
    def backward(ctx, g):
        (fv,) = ctx.saved_tensors
        g = g.to(torch.float32).contiguous()
        gfv = torch.empty_like(fv)
        N.check(N.lib().mm_face_normals_backward(fv.numel() // 9, ctx.unit, N.ptr(fv), N.ptr(g), N.ptr(gfv), N.current_stream(fv.device)),
                "mm_face_normals_backward")
        return gfv, None


def face_normals(face_vertices, unit=False):
    """kaolin.ops.mesh.face_normals: (B,F,3,3) -> (B,F,3), cross(v1-v0, v2-v0) [/ (length + 1e-10) if unit]."""
    N.require_device(face_vertices)
    if face_vertices.shape[-2:] != (3, 3):
        raise RuntimeError("face_vertices must be (...,3,3), got %s" % (tuple(face_vertices.shape),))
    return _FaceNormalsFn.apply(face_vertices, bool(unit))


# ---- kaolin.render.mesh.dibr_rasterization -----------------------------------------------------------------------------
class _DibrFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, H, W, fz, fvi, feats, fnz, sigmainv, boxlen, knum, multiplier, eps):
        dev = fvi.device
        fz, fvi, feats, fnz = _f32(fz, dev), _f32(fvi, dev), _f32(feats, dev), _f32(fnz, dev)
        B, F = fvi.shape[:2]
        D = feats.shape[-1]
        interp = torch.empty((B, H, W, D), device=dev, dtype=torch.float32)
        soft = torch.empty((B, H, W), device=dev, dtype=torch.float32)
        fidx = torch.empty((B, H, W), device=dev, dtype=torch.int64)
        d = N.MMDibrDesc()
        d.B, d.H, d.W, d.F, d.D, d.knum = B, H, W, F, D, int(knum)
        d.sigmainv, d.boxlen, d.multiplier, d.eps = float(sigmainv), float(boxlen), float(multiplier), float(eps)
        d.face_vertices_z, d.face_vertices_image, d.face_features, d.face_normals_z = N.ptr(fz), N.ptr(fvi), N.ptr(feats), N.ptr(fnz)
        d.interpolated_features, d.soft_mask, d.face_idx = N.ptr(interp), N.ptr(soft), N.ptr(fidx)
        ws = torch.empty(N.lib().mm_dibr_query_workspace(ctypes.byref(d)), device=dev, dtype=torch.uint8)
        d.workspace, d.workspace_bytes = N.ptr(ws), ws.numel()
        N.check(N.lib().mm_dibr_rasterization_forward(ctypes.byref(d), N.current_stream(dev)), "mm_dibr_rasterization_forward")
        ctx.save_for_backward(fz, fvi, feats, fnz, ws)
        ctx.cfg = (H, W, int(knum), float(sigmainv), float(boxlen), float(multiplier), float(eps))
        ctx.mark_non_differentiable(fidx)
        return interp, soft, fidx

    @staticmethod
    def backward(ctx, g_interp, g_soft, _g_idx):
        fz, fvi, feats, fnz, ws = ctx.saved_tensors
        dev = fvi.device
        H, W, knum, sigmainv, boxlen, multiplier, eps = ctx.cfg
        B, F = fvi.shape[:2]
        c = lambda g: None if g is None else g.to(torch.float32).contiguous()
        g_interp, g_soft = c(g_interp), c(g_soft)
        d = N.MMDibrDesc()
        d.B, d.H, d.W, d.F, d.D, d.knum = B, H, W, F, feats.shape[-1], knum
        d.sigmainv, d.boxlen, d.multiplier, d.eps = sigmainv, boxlen, multiplier, eps
        d.face_vertices_z, d.face_vertices_image, d.face_features, d.face_normals_z = N.ptr(fz), N.ptr(fvi), N.ptr(feats), N.ptr(fnz)
        d.workspace, d.workspace_bytes = N.ptr(ws), ws.numel()
        gfvi = torch.empty_like(fvi)
        gfeat = torch.empty_like(feats) if ctx.needs_input_grad[4] else None
        g = N.MMDibrGrads(N.ptr(g_interp), N.ptr(g_soft), N.ptr(gfvi), N.ptr(gfeat))
        N.check(N.lib().mm_dibr_rasterization_backward(ctypes.byref(d), ctypes.byref(g), N.current_stream(dev)), "mm_dibr_rasterization_backward")
        return None, None, None, gfvi, gfeat, None, None, None, None, None, None


def dibr_rasterization(height, width, face_vertices_z, face_vertices_image, face_features, face_normals_z,
                       sigmainv=7000, boxlen=0.02, knum=30, multiplier=None, eps=None, rast_backend='cuda'):
    """kaolin.render.mesh.dibr_rasterization -> (interpolated_features, soft_mask, face_idx).

    ``face_features`` (B,F,3,D) or a list / tuple of such tensors (then a tuple of interpolated tensors comes back, as
    upstream); faces with ``face_normals_z >= 0`` are rasterised; the soft mask covers every face.  ``multiplier`` defaults
    to 1000 and ``eps`` to 1e-8 like upstream.  ``rast_backend`` is accepted for signature compatibility ('cuda' is HIP here;
    there is no nvdiffrast)."""
    if rast_backend not in ('cuda', 'hip'):
        raise ValueError("rast_backend '%s' is not available in the MI355X build" % rast_backend)
    N.require_device(face_vertices_z, face_vertices_image, face_normals_z)
    multiplier = 1000 if multiplier is None else multiplier
    eps = 1e-8 if eps is None else eps
    is_list = isinstance(face_features, (list, tuple))
    feats = torch.cat(list(face_features), dim=-1) if is_list else face_features
    N.require_device(feats)
    B, F = face_vertices_image.shape[:2]
    if face_vertices_image.shape != (B, F, 3, 2) or face_vertices_z.shape != (B, F, 3) or face_normals_z.shape != (B, F) or \
            feats.dim() != 4 or feats.shape[:3] != (B, F, 3):
        raise RuntimeError("dibr_rasterization: expected face_vertices_z (B,F,3), face_vertices_image (B,F,3,2), face_features "
                           "(B,F,3,D), face_normals_z (B,F)")
    interp, soft, fidx = _DibrFn.apply(int(height), int(width), face_vertices_z, face_vertices_image, feats, face_normals_z,
                                       sigmainv, boxlen, knum, multiplier, eps)
    if is_list:
        interp = tuple(torch.split(interp, [f.shape[-1] for f in face_features], dim=-1))
    return interp, soft, fidx


# ---- kaolin.render.mesh.texture_mapping --------------------------------------------------------------------------------
class _TexMapFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, uv, tex, mode):
        dev = tex.device
        uv, tex = _f32(uv, dev), _f32(tex, dev)
        B, C, Ht, Wt = tex.shape
        n = uv.shape[1]
        out = torch.empty((B, n, C), device=dev, dtype=torch.float32)
        d = N.MMTexMapDesc(B, n, C, Ht, Wt, mode, N.ptr(uv), N.ptr(tex), N.ptr(out))
        N.check(N.lib().mm_texture_mapping_forward(ctypes.byref(d), N.current_stream(dev)), "mm_texture_mapping_forward")
        ctx.save_for_backward(uv, tex)
        ctx.mode = mode
        return out

    @staticmethod
    def backward(ctx, g):
        uv, tex = ctx.saved_tensors
        dev = tex.device
        B, C, Ht, Wt = tex.shape
        g = g.to(torch.float32).contiguous()
        guv = torch.empty_like(uv) if ctx.needs_input_grad[0] else None
        gtex = torch.empty_like(tex) if ctx.needs_input_grad[1] else None
        if guv is None and gtex is None:
            return None, None, None
        d = N.MMTexMapDesc(B, uv.shape[1], C, Ht, Wt, ctx.mode, N.ptr(uv), N.ptr(tex), None)
        ws = None
        if gtex is not None:                                     # 64-bit fixed-point scatter: a bitwise reproducible texture gradient
            ws = torch.empty(N.lib().mm_texture_mapping_backward_query_workspace(ctypes.byref(d)), device=dev, dtype=torch.uint8)
        gr = N.MMTexMapGrads(N.ptr(g), N.ptr(guv), N.ptr(gtex), N.ptr(ws), 0 if ws is None else ws.numel())
        N.check(N.lib().mm_texture_mapping_backward(ctypes.byref(d), ctypes.byref(gr), N.current_stream(dev)), "mm_texture_mapping_backward")
        return guv, gtex, None


def texture_mapping(texture_coordinates, texture_maps, mode='nearest'):
    """kaolin.render.mesh.texture_mapping: coordinates (B,H,W,2) or (B,N,2) in [0,1] (v up), maps (B,C,Ht,Wt) or (C,Ht,Wt)
    -> (B,H,W,C) / (B,N,C).  'nearest' or 'bilinear' = grid_sample(align_corners=False, padding_mode='border').
    NON-FINITE upstream gradients: the backward accumulates the texture gradient in per-image fixed point (bitwise reproducible), scaled by the
    image's largest |gradient| -- one NaN / inf element makes that image's WHOLE texture gradient NaN, where ATen's grid_sampler / kaolin poison only
    the touched texels (include/mm_render.h: MMTexMapGrads.workspace).  Loud by design; mask non-finite values BEFORE the backward, not per texel after."""
    if mode not in ('nearest', 'bilinear'):
        raise ValueError("texture_mapping: mode must be 'nearest' or 'bilinear'")
    N.require_device(texture_coordinates, texture_maps)
    B = texture_coordinates.shape[0]
    if texture_maps.dim() == 3:
        texture_maps = texture_maps.unsqueeze(0).expand(B, -1, -1, -1)
    if texture_coordinates.shape[-1] != 2 or texture_maps.dim() != 4 or texture_maps.shape[0] != B:
        raise RuntimeError("texture_mapping: coordinates (B,...,2) and maps (B,C,Ht,Wt) expected")
    lead = texture_coordinates.shape[:-1]
    out = _TexMapFn.apply(texture_coordinates.reshape(B, -1, 2), texture_maps, 1 if mode == 'bilinear' else 0)
    return out.reshape(lead + (texture_maps.shape[1],))


# ---- kaolin.render.mesh.spherical_harmonic_lighting --------------------------------------------------------------------
class _ShFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, nrm, lights):
        dev = nrm.device
        nrm, lights = _f32(nrm, dev), _f32(lights, dev)
        B, n = nrm.shape[:2]
        out = torch.empty((B, n), device=dev, dtype=torch.float32)
        d = N.MMShDesc(B, n, N.ptr(nrm), N.ptr(lights), N.ptr(out))
        N.check(N.lib().mm_sh_lighting_forward(ctypes.byref(d), N.current_stream(dev)), "mm_sh_lighting_forward")
        ctx.save_for_backward(nrm, lights)
        return out

    @staticmethod
    def backward(ctx, g):
        nrm, lights = ctx.saved_tensors
        dev = nrm.device
        g = g.to(torch.float32).contiguous()
        gn = torch.empty_like(nrm) if ctx.needs_input_grad[0] else None
        gl = torch.empty_like(lights) if ctx.needs_input_grad[1] else None
        if gn is None and gl is None:
            return None, None
        d = N.MMShDesc(nrm.shape[0], nrm.shape[1], N.ptr(nrm), N.ptr(lights), None)
        gr = N.MMShGrads(N.ptr(g), N.ptr(gn), N.ptr(gl))
        N.check(N.lib().mm_sh_lighting_backward(ctypes.byref(d), ctypes.byref(gr), N.current_stream(dev)), "mm_sh_lighting_backward")
        return gn, gl


def spherical_harmonic_lighting(imnormal, lights):
    """kaolin.render.mesh.spherical_harmonic_lighting: normals (B,...,3), lights (B,9) -> (B,...)."""
    N.require_device(imnormal, lights)
    B = imnormal.shape[0]
    if imnormal.shape[-1] != 3 or lights.shape != (B, 9):
        raise RuntimeError("spherical_harmonic_lighting: imnormal (B,...,3) and lights (B,9) expected")
    lead = imnormal.shape[:-1]
    return _ShFn.apply(imnormal.reshape(B, -1, 3), lights).reshape(lead)


# ---- kaolin.metrics.render.mask_iou ------------------------------------------------------------------------------------
class _MaskIouFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lhs, rhs):
        dev = lhs.device
        lhs, rhs = _f32(lhs, dev), _f32(rhs, dev)
        B = lhs.shape[0]
        sums = torch.empty((B, 2), device=dev, dtype=torch.float32)
        loss = torch.empty((), device=dev, dtype=torch.float32)
        d = N.MMMaskIouDesc(B, lhs.numel() // B, N.ptr(lhs), N.ptr(rhs), N.ptr(sums), N.ptr(loss))
        N.check(N.lib().mm_mask_iou_forward(ctypes.byref(d), N.current_stream(dev)), "mm_mask_iou_forward")
        ctx.save_for_backward(lhs, rhs, sums)
        return loss

    @staticmethod
    def backward(ctx, g):
        lhs, rhs, sums = ctx.saved_tensors
        dev = lhs.device
        B = lhs.shape[0]
        g = g.to(device=dev, dtype=torch.float32).contiguous()
        gl = torch.empty_like(lhs) if ctx.needs_input_grad[0] else None
        gr = torch.empty_like(rhs) if ctx.needs_input_grad[1] else None
        if gl is None and gr is None:
            return None, None
        d = N.MMMaskIouDesc(B, lhs.numel() // B, N.ptr(lhs), N.ptr(rhs), N.ptr(sums), None)
        N.check(N.lib().mm_mask_iou_backward(ctypes.byref(d), N.ptr(g), N.ptr(gl), N.ptr(gr), N.current_stream(dev)), "mm_mask_iou_backward")
        return gl, gr


def mask_iou(lhs_mask, rhs_mask):
    """kaolin.metrics.render.mask_iou: 1 - mean_b[ sum(l*r) / (sum(l + r - l*r) + 1e-10) ] for (B,H,W) masks.

    The reference's evaluation loop calls it on (1,H,W) HOST tensors read from PNG files (trainer.py:793,933): those are moved to
    the current HIP device, reduced there by the same kernel, and the scalar is returned on the host like upstream would."""
    if lhs_mask.shape != rhs_mask.shape or lhs_mask.dim() != 3:
        raise RuntimeError("mask_iou expects two (B,H,W) masks, got %s / %s" % (tuple(lhs_mask.shape), tuple(rhs_mask.shape)))
    if not lhs_mask.is_cuda and not rhs_mask.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError("mask_iou runs on the MI355X; no HIP device is available and there is no CPU fallback")
        return _MaskIouFn.apply(lhs_mask.cuda(), rhs_mask.cuda()).to(lhs_mask.device)
    dev = lhs_mask.device if lhs_mask.is_cuda else rhs_mask.device
    return _MaskIouFn.apply(lhs_mask.to(dev), rhs_mask.to(dev))


# ---- pytorch3d.loss.chamfer_distance -----------------------------------------------------------------------------------
def chamfer_distance(x, y, x_lengths=None, y_lengths=None, x_normals=None, y_normals=None, weights=None,
                     batch_reduction="mean", point_reduction="mean", norm=2, **kwargs):
    """pytorch3d.loss.chamfer_distance with the defaults the reference uses (networks.py:342,356; trainer.py:445,469,483):
    returns (loss, None).  Anything but the default reductions / L2 / full-length clouds is refused rather than approximated."""
    if x_lengths is not None or y_lengths is not None or x_normals is not None or y_normals is not None or weights is not None or \
            batch_reduction != "mean" or point_reduction != "mean" or norm != 2 or kwargs:
        raise NotImplementedError("only pytorch3d.loss.chamfer_distance(x, y) with default arguments is implemented for MI355X")
    return _chamfer(x, y)
