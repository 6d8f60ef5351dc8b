"""Host-side mirror of the reference's ``DiffRender`` (/root/reference/networks.py:164-491) over the gfx950 C ABI.

Same constructor, attributes, method names, argument meaning and return values as the reference class, so a
``trainer.py``-style loop can switch with ``from mm_amd import DiffRender``.  ``render`` and ``recon_data`` run the
hand-written HIP kernels of ``lib/libmm_render.so`` through ``torch.autograd.Function`` wrappers; there is no CPU or
eager-torch fallback for them.  The mesh regularisers (``recon_flip``, ``calc_reg_*``) run as one HIP
launch per direction (``mesh_reg.py``), and so do the attribute losses of ``recon_att`` (``att_loss.py``); its chamfer term is
a HIP nearest-neighbour kernel.
"""
import ctypes
import math

import numpy as np
import torch

from . import _native as N
from . import att_loss, mesh_reg, obj_io, template


class _PooledWorkspace(object):
    """A render workspace borrowed from its DiffRender's pool for as long as the autograd node that owns it lives (forward ->
    last backward, retain_graph included); it goes back when the node is freed.  Five or six renders of a trainer iteration thus
    cycle through a handful of buffers instead of allocating ~200 MB each (the library never allocates; the host side owns scratch)."""

    def __init__(self, pool, key, nbytes, device):
        # The pool is keyed by the stream the work is enqueued on as well: a buffer released by a node whose kernels are still queued
        # on stream A is only ever handed to another render on stream A, i.e. behind them in stream order (the torch caching allocator
        # the C++ nodes use gives the same guarantee).
        self.pool, self.key = pool, key + (torch.cuda.current_stream(device).cuda_stream,)
        free = pool.setdefault(self.key, [])
        self.buf = free.pop() if free else torch.empty(nbytes, device=device, dtype=torch.uint8)

    def __del__(self):
        try:
            free = self.pool.setdefault(self.key, [])
            if len(free) < 8:
                free.append(self.buf)
        except Exception:                                        # interpreter shutdown
            pass


def _render_inputs(dr, no_mask, vertices, textures, lights, bg, azimuths, elevations, distances, biases):
    """fp32 dense device tensors + the shape checks of the render boundary (shared by render and render_recon)."""
    N.require_device(vertices, textures, lights, bg, azimuths, elevations, distances, biases)
    dev = azimuths.device
    f32 = lambda t: N.as_f32(t, dev)
    vertices, textures, lights, bg = f32(vertices), f32(textures), f32(lights), f32(bg)
    azimuths, elevations, distances, biases = f32(azimuths).reshape(-1), f32(elevations).reshape(-1), f32(distances).reshape(-1), f32(biases)
    B = azimuths.shape[0]
    H, W = dr.render_height, dr.image_size
    if vertices.shape != (B, dr.num_vertices, 3):
        raise RuntimeError("vertices must be (B,%d,3), got %s" % (dr.num_vertices, tuple(vertices.shape)))
    if textures.dim() != 4 or textures.shape[0] != B or textures.shape[1] != 3:
        raise RuntimeError("textures must be (B,3,Ht,Wt), got %s" % (tuple(textures.shape),))
    if lights.shape != (B, 9) or biases.shape != (B, 2) or elevations.shape[0] != B or distances.shape[0] != B:
        raise RuntimeError("lights (B,9), biases (B,2), elevations/distances (B) expected")
    if no_mask:
        if bg is None:
            raise TypeError("render(no_mask=True) needs attributes['bg'] (B,3,H,W)")   # reference: None.permute fails
        if bg.shape != (B, 3, H, W):
            raise RuntimeError("bg must be (B,3,%d,%d), got %s" % (H, W, tuple(bg.shape)))
    return dev, B, H, W, vertices, textures, lights, bg, azimuths, elevations, distances, biases


class _RenderFn(torch.autograd.Function):
    """rgba (B,H,W,4), face_normals (B,F,3), imnormal (B,H,W,3), face_idx (B,H,W) = render(attributes).
    With ``gt`` (B,4,H,W): the recon_data loss of the batch is folded into the same kernels (MMRenderDesc.fused_*) and returned as a
    fifth output; the image is then an output without gradient (its only consumer, the loss, is already inside)."""

    @staticmethod
    def forward(ctx, dr, no_mask, want_imnormal, gt, vertices, textures, lights, bg, azimuths, elevations, distances, biases, contour):
        ctx.cam_shapes = (tuple(azimuths.shape), tuple(elevations.shape), tuple(distances.shape))   # (B), (B,1), ...: the gradients go back in these
        dev, B, H, W, vertices, textures, lights, bg, azimuths, elevations, distances, biases = _render_inputs(
            dr, no_mask, vertices, textures, lights, bg, azimuths, elevations, distances, biases)
        st = dr._static(dev)
        if want_imnormal == "geometry":                          # render_geometry: the vertex stage alone (MMRenderDesc.geometry_only)
            fn = torch.empty((B, dr.num_faces, 3), device=dev, dtype=torch.float32)
            d = dr._desc(st, B, False, vertices, textures, lights, None, azimuths, elevations, distances, biases, None, None, fn, None)
            d.geometry_only = 1
            nbytes = dr.workspace_bytes(d)
            holder = _PooledWorkspace(dr._ws_pool, (str(dev), nbytes), nbytes, dev)
            d.workspace, d.workspace_bytes = N.ptr(holder.buf), holder.buf.numel()
            with torch.cuda.device(dev):
                N.check(N.lib().mm_render_forward(ctypes.byref(d), N.current_stream(dev)), "mm_render_forward")
            ctx.dr, ctx.geometry, ctx.ws_holder = dr, True, holder
            ctx.save_for_backward(vertices, textures, azimuths, elevations, distances, biases)
            ctx.set_materialize_grads(False)
            return fn
        ctx.geometry = False
        rgba = torch.empty((B, H, W, 4), device=dev, dtype=torch.float32)
        face_idx = torch.empty((B, H, W), device=dev, dtype=torch.int32)
        fn = torch.empty((B, dr.num_faces, 3), device=dev, dtype=torch.float32)
        imn = torch.empty((B, H, W, 3), device=dev, dtype=torch.float32) if want_imnormal else None
        d = dr._desc(st, B, no_mask, vertices, textures, lights, bg, azimuths, elevations, distances, biases, rgba, face_idx, fn, imn)
        loss = None
        if gt is not None:
            N.require_device(gt)
            gt = N.as_f32(gt, dev)
            if gt.shape != (B, 4, H, W):
                raise RuntimeError("gt_data must be (B,4,%d,%d), got %s" % (H, W, tuple(gt.shape)))
            loss = torch.empty((), device=dev, dtype=torch.float32)
            d.fused_gt, d.fused_image_weight, d.fused_loss = N.ptr(gt), float(dr.image_weight), N.ptr(loss)
            d.fused_contour = float(contour)
        nbytes = dr.workspace_bytes(d)
        holder = _PooledWorkspace(dr._ws_pool, (str(dev), nbytes), nbytes, dev)
        ws = holder.buf
        d.workspace, d.workspace_bytes = N.ptr(ws), ws.numel()
        with torch.cuda.device(dev):                              # launches go to the tensors' device whatever the caller's current one is
            N.check(N.lib().mm_render_forward(ctypes.byref(d), N.current_stream(dev)), "mm_render_forward")
            if gt is not None:
                N.check(N.lib().mm_render_fused_loss(ctypes.byref(d), N.current_stream(dev)), "mm_render_fused_loss")
        ctx.dr, ctx.no_mask, ctx.fused, ctx.contour = dr, bool(no_mask), gt is not None, float(contour)
        ctx.options = int(d.options)                              # the backward uses the FORWARD's option bits (which walk form set the face flags), whatever dr.options says by then
        ctx.ws_holder = holder                                   # returned to the pool when this node dies
        ctx.save_for_backward(vertices, textures, lights, bg, azimuths, elevations, distances, biases, face_idx, fn, gt)
        # (the image is not saved: the backward re-forms the prediction per pixel, bit for bit, and the caller may overwrite rgba)
        ctx.mark_non_differentiable(face_idx)
        ctx.set_materialize_grads(False)                          # unused outputs arrive as None in backward, not as zero-filled tensors
        if imn is None:
            imn = torch.empty(0, device=dev)
        ctx.mark_non_differentiable(imn)
        if gt is None:
            return rgba, fn, imn, face_idx
        ctx.mark_non_differentiable(rgba)
        return rgba, fn, imn, face_idx, loss

    @staticmethod
    def backward(ctx, g_rgba, g_fn=None, _g_imn=None, _g_idx=None, g_loss=None):
        if ctx.geometry:                                         # (g_rgba is dL/dface_normals here: the node's only output)
            vertices, textures, azimuths, elevations, distances, biases = ctx.saved_tensors
            if g_rgba is None:
                return (None,) * 13
            dr, dev, B = ctx.dr, azimuths.device, azimuths.shape[0]
            g_fn = g_rgba.to(torch.float32).contiguous()
            d = dr._desc(dr._static(dev), B, False, vertices, textures, None, None, azimuths, elevations, distances, biases, None, None, None, None)
            d.geometry_only = 1
            d.face_normals = g_fn.data_ptr()                     # (a valid pointer for the argument check; the backward does not read the normals)
            ws = ctx.ws_holder.buf
            d.workspace, d.workspace_bytes = N.ptr(ws), ws.numel()
            gv = torch.empty_like(vertices)
            ga, ge, gd, gb = torch.empty_like(azimuths), torch.empty_like(elevations), torch.empty_like(distances), torch.empty_like(biases)
            g = N.MMRenderGrads(None, N.ptr(g_fn), N.ptr(gv), None, None, None, N.ptr(ga), N.ptr(ge), N.ptr(gd), N.ptr(gb))
            with torch.cuda.device(dev):
                N.check(N.lib().mm_render_backward(ctypes.byref(d), ctypes.byref(g), N.current_stream(dev)), "mm_render_backward")
            sa, se, sd = ctx.cam_shapes
            return None, None, None, None, gv, None, None, None, ga.reshape(sa), ge.reshape(se), gd.reshape(sd), gb, None
        vertices, textures, lights, bg, azimuths, elevations, distances, biases, face_idx, fn, gt = ctx.saved_tensors
        ws = ctx.ws_holder.buf
        dr, dev = ctx.dr, azimuths.device
        B = azimuths.shape[0]
        H, W = dr.render_height, dr.image_size
        st = dr._static(dev)
        g_fn = None if g_fn is None else g_fn.to(torch.float32).contiguous()
        d = dr._desc(st, B, ctx.no_mask, vertices, textures, lights, bg, azimuths, elevations, distances, biases, None, face_idx, fn, None)   # (rgba: not read by the backward)
        d.options = ctx.options
        if ctx.fused:
            # None = the loss output took no part in what is being differentiated (materialize_grads is off): its gradient is ZERO,
            # never one -- e.g. reg.backward() through attributes['face_normals'] after loss.backward(retain_graph=True)
            g_loss = (torch.zeros((), device=dev, dtype=torch.float32) if g_loss is None
                      else g_loss.to(device=dev, dtype=torch.float32).reshape(()).contiguous())
            d.fused_gt, d.fused_image_weight, d.fused_grad_loss = N.ptr(gt), float(dr.image_weight), N.ptr(g_loss)
            d.fused_contour = ctx.contour
        else:
            if g_rgba is None:
                g_rgba = torch.zeros((B, H, W, 4), device=dev, dtype=torch.float32)
            g_rgba = g_rgba.to(torch.float32).contiguous()
        d.workspace, d.workspace_bytes = N.ptr(ws), ws.numel()
        gv, gt_, gl = torch.empty_like(vertices), torch.empty_like(textures), torch.empty_like(lights)
        gbg = torch.empty_like(bg) if ctx.no_mask else None
        ga, ge, gd, gb = torch.empty_like(azimuths), torch.empty_like(elevations), torch.empty_like(distances), torch.empty_like(biases)
        g = N.MMRenderGrads(None if ctx.fused else N.ptr(g_rgba), N.ptr(g_fn), N.ptr(gv), N.ptr(gt_), N.ptr(gl), N.ptr(gbg), N.ptr(ga), N.ptr(ge), N.ptr(gd), N.ptr(gb))
        with torch.cuda.device(dev):
            N.check(N.lib().mm_render_backward(ctypes.byref(d), ctypes.byref(g), N.current_stream(dev)), "mm_render_backward")
            if ctx.dr.check_texture_records:                     # (synchronises: a diagnostic switch)
                ctx.dr.check_records(d, N.current_stream(dev))
        sa, se, sd = ctx.cam_shapes
        return None, None, None, None, gv, gt_, gl, gbg, ga.reshape(sa), ge.reshape(se), gd.reshape(sd), gb, None


class _ReconFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, image_weight, contour):
        N.require_device(pred, gt)
        dev = pred.device
        pred = pred.detach().to(torch.float32)
        if not _dense_non_overlapping(pred):
            pred = pred.contiguous()
        gt = gt.detach().to(device=dev, dtype=torch.float32).contiguous()
        B, C, H, W = pred.shape
        if C != 4 or gt.shape != pred.shape:
            raise RuntimeError("recon_data expects (B,4,H,W) prediction and target, got %s / %s" % (tuple(pred.shape), tuple(gt.shape)))
        loss = torch.empty((), device=dev, dtype=torch.float32)
        d = N.MMReconDesc()
        d.B, d.H, d.W = B, H, W
        d.pred, d.gt = N.ptr(pred), N.ptr(gt)
        for i, s in enumerate(pred.stride()):
            d.pred_strides[i] = s
        d.image_weight, d.contour = float(image_weight), float(contour)
        d.loss = N.ptr(loss)
        ws = torch.empty(N.lib().mm_recon_query_workspace(ctypes.byref(d)), device=dev, dtype=torch.uint8)
        d.workspace, d.workspace_bytes = N.ptr(ws), ws.numel()
        N.check(N.lib().mm_recon_data_forward(ctypes.byref(d), N.current_stream(dev)), "mm_recon_data_forward")
        ctx.save_for_backward(pred, gt, ws)
        ctx.cfg = (float(image_weight), float(contour))
        return loss

    @staticmethod
    def backward(ctx, g_loss):
        pred, gt, ws = ctx.saved_tensors
        dev = pred.device
        B, _, H, W = pred.shape
        g_loss = g_loss.to(device=dev, dtype=torch.float32).contiguous()
        grad = torch.empty_strided(pred.shape, pred.stride(), device=dev, dtype=torch.float32)
        d = N.MMReconDesc()
        d.B, d.H, d.W = B, H, W
        d.pred, d.gt = N.ptr(pred), N.ptr(gt)
        for i, s in enumerate(pred.stride()):
            d.pred_strides[i] = s
        d.image_weight, d.contour = ctx.cfg
        d.grad_loss, d.grad_pred = N.ptr(g_loss), N.ptr(grad)
        d.workspace, d.workspace_bytes = N.ptr(ws), ws.numel()
        N.check(N.lib().mm_recon_data_backward(ctypes.byref(d), N.current_stream(dev)), "mm_recon_data_backward")
        return grad, None, None, None


class _SplitBatchFn(torch.autograd.Function):
    """x (sum(sizes), ...) -> its per-set pieces along the batch, with a backward that CONCATENATES the pieces' gradients (one copy per piece
    into one buffer).  Plain slicing would have autograd build, per piece, a full-size zero tensor with the piece filled in and then add
    them up: three fills and two adds of the whole batched image where one write per piece suffices (render_many)."""

    @staticmethod
    def forward(ctx, x, *sizes):
        ctx.sizes, ctx.shape = sizes, tuple(x.shape)
        ctx.set_materialize_grads(False)
        out, o = [], 0
        for n in sizes:
            out.append(x.narrow(0, o, n))
            o += n
        return tuple(out)

    @staticmethod
    def backward(ctx, *grads):
        if all(g is None for g in grads):
            return (None,) * (1 + len(ctx.sizes))
        ref = next(g for g in grads if g is not None)
        full = torch.empty(ctx.shape, device=ref.device, dtype=ref.dtype)
        o = 0
        for n, g in zip(ctx.sizes, grads):
            if g is None:
                full.narrow(0, o, n).zero_()
            else:
                full.narrow(0, o, n).copy_(g)
            o += n
        return (full,) + (None,) * len(ctx.sizes)


def _dense_non_overlapping(t):
    expect = 1
    for size, stride in sorted(zip(t.shape, t.stride()), key=lambda p: p[1]):
        if size == 1:
            continue
        if stride != expect:
            return False
        expect *= size
    return True


class DiffRender(object):
    """Drop-in for ``networks.DiffRender`` (networks.py:164)."""

    def __init__(self, mesh_name, image_size, ratio=1, init_ellipsoid=1, image_weight=0.1, lambda_lpl=0.1, lambda_flat=0.001,
                 emit_imnormal=True, verbose=False):
        self.image_size = image_size
        self.image_weight = image_weight
        self.lambda_lpl = lambda_lpl
        self.lambda_flat = lambda_flat
        self.ratio = ratio
        self.emit_imnormal = emit_imnormal
        self.options = 0                                # MMRenderDesc.options: MM_OPT_* bits (SURVEY Appendix C switches); 0 = defaults
        # The render workspace is the library's minimum (mm_query_workspace) plus room for this many MORE texture-gradient records per pixel
        # than the 9/8 the minimum holds (include/mm_render.h); an image that runs out gets NaN texture gradients, and with
        # check_texture_records every backward of the class API asks the library (a stream synchronisation) and raises instead.
        self.extra_texture_records_per_pixel = 0.0
        self.check_texture_records = False
        # DEFERRED FUSION (C++ nodes only; csrc/mm_torch_ext.cpp, MMRenderDesc.fused_totals): `recon_data(pred, gt)` on the untouched image of an earlier
        # `render` of this process -- the un-modified trainer's order of calls (trainer.py:276,441) -- forms its value as always and routes its BACKWARD
        # through the render node: no dL/d image tensor, no loss-backward launch; every gradient of the render's inputs has the bits of the two separate
        # backward passes.  The one observable difference: recon_data's contribution to the gradient OF THE IMAGE ITSELF (torch.autograd.grad(loss, rgbs),
        # a hook on rgbs, rgbs.retain_grad()) does not exist as a tensor -- set this to False if the image's own gradient is inspected.
        self.defer_recon_fusion = True
        camera_fovy = np.arctan(1.0 / 2.5) * 2
        self.cam_proj = template.generate_perspective_projection(camera_fovy, ratio=1 / ratio)     # networks.py:172-174
        mesh = obj_io.load_template(mesh_name)                                                   # :176
        self.vertices_init = template.normalize_template(mesh.vertices, init_ellipsoid)          # :181-194
        self.faces = mesh.faces
        self.uvs = mesh.uvs
        self.face_uvs = template.index_vertices_by_faces(mesh.uvs.unsqueeze(0), mesh.face_uvs_idx).detach()  # :196-202
        self.num_faces = self.faces.shape[0]
        self.num_vertices = self.vertices_init.shape[0]
        self.flip_index = template.flip_pairing(self.vertices_init)                              # :215-217
        self.edges, self.edge2faces = template.edge_tables(self.faces)                           # :220-246
        self.vertices_laplacian_matrix = template.uniform_laplacian(self.num_vertices, self.faces)  # :249
        self.sign_init = torch.sign(self.vertices_init[:, 2])                                    # :252 (device copy made lazily)
        if torch.cuda.is_available():
            self.sign_init = self.sign_init.cuda()
        self.render_height = round(self.ratio * self.image_size)                                  # :298
        self._vc_table = template.vertex_corner_table(self.num_vertices, self.faces)     # (V, stride, 4): the backward's vertex -> corner gather
        self._static_cache = {}
        self._desc_cache = {}
        self._ws_pool = {}                                       # (device, bytes) -> free render workspaces (see _PooledWorkspace)
        self._status = None                                      # one pinned int32 the backward kernels add dropped-record counts to (MMRenderDesc.status_flag)
        # dibr_rasterization defaults (kaolin v0.12.0): sigmainv=7000, boxlen=0.02, knum=30, multiplier=1000, eps=1e-8
        self.sigmainv, self.boxlen, self.knum, self.multiplier, self.eps = 7000.0, 0.02, 30, 1000.0, 1e-8
        if verbose:
            print("Vertices Number:", self.num_vertices)
            print("Faces Number:", self.faces.shape)
            print("Unique Edge Number: %d" % self.edges.shape[0])

    # ---- device-resident static template data (the reference re-uploads faces/face_uvs on every call, :272-273) ----
    def _static(self, device):
        key = str(device)
        st = self._static_cache.get(key)
        if st is None:
            st = {"faces": self.faces.to(device=device, dtype=torch.int32).contiguous(),
                  "face_uvs": self.face_uvs.to(device=device, dtype=torch.float32).reshape(-1, 3, 2).contiguous(),
                  "vc_table": self._vc_table.to(device=device, dtype=torch.int32).contiguous()}
            self._static_cache[key] = st
        return st

    def _render_node(self, no_mask, gt, vertices, textures, lights, bg, azimuths, elevations, distances, biases, contour=0.0):
        """One autograd node for the render (+ the fused loss if gt is given).  With lib/mm_torch_ext.so built, the node is C++
        (csrc/mm_torch_ext.cpp: no Python in the backward); otherwise the torch.autograd.Function above issues the same ABI calls."""
        self._raise_if_records_were_dropped()                    # (an overflow of an EARLIER step's backward: a host read of pinned memory, no sync)
        ext = None if self.check_texture_records else N.torch_ext()   # (the diagnostic switch lives in the Python nodes)
        if ext is None:
            return _RenderFn.apply(self, no_mask, self.emit_imnormal, gt, vertices, textures, lights, bg, azimuths, elevations, distances, biases,
                                   float(contour))
        N.require_device(azimuths)
        if no_mask and bg is None:
            raise TypeError("render(no_mask=True) needs attributes['bg'] (B,3,H,W)")   # reference: None.permute fails
        if textures.dim() != 4:
            raise RuntimeError("textures must be (B,3,Ht,Wt), got %s" % (tuple(textures.shape),))
        dev = azimuths.device
        proto, nbytes = self._proto(self._static(dev), azimuths.numel(), no_mask, textures.shape[2], textures.shape[3], float(contour) if gt is not None else 0.0)
        return ext.render(N.fn_addr("mm_render_forward"), N.fn_addr("mm_render_fused_loss"), N.fn_addr("mm_render_backward"), proto, nbytes,
                          vertices, textures, lights, bg, azimuths, elevations, distances, biases, gt, bool(self.emit_imnormal),
                          float(self.image_weight), torch._C._cuda_getCurrentRawStream(dev.index), bool(self.defer_recon_fusion))

    def _status_ptr(self):
        """Address of this object's pinned status word (device-writable host memory): every descriptor built here carries it, so that a
        backward that drops texture-gradient records -- eager, C++ node or captured graph -- is noticed WITHOUT a synchronisation."""
        if self._status is None:
            self._status = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._status_word = ctypes.c_int32.from_address(self._status.data_ptr())     # (a plain host read per poll: ~0.1 us)
        return self._status.data_ptr()

    def poll_dropped_records(self, reset=True):
        """Texture-gradient records dropped by backward passes that have COMPLETED since the last poll (no synchronisation: a host read)."""
        if self._status is None:
            return 0
        n = self._status_word.value
        if n and reset:
            self._status_word.value = 0
        return n

    def _raise_if_records_were_dropped(self):
        n = self.poll_dropped_records()
        if n:
            raise RuntimeError("an earlier mm_render_backward of this DiffRender dropped %d texture-gradient records (record pool overflow): the texture "
                               "gradients of the affected images were NaN. Raise DiffRender.extra_texture_records_per_pixel." % n)

    def workspace_bytes(self, d):
        """Bytes of the render workspace for the shape in MMRenderDesc `d`: the library's minimum + the extra record pool asked for."""
        extra = int(np.ceil(max(0.0, float(self.extra_texture_records_per_pixel)) * d.H * d.W)) * 24 * d.B
        return int(N.lib().mm_query_workspace(ctypes.byref(d))) + extra

    @staticmethod
    def check_records(d, stream):
        """Raise if the last mm_render_backward on MMRenderDesc `d` dropped texture-gradient records (synchronises `stream`)."""
        dropped = (ctypes.c_int32 * d.B)()
        st = N.lib().mm_render_status(ctypes.byref(d), stream, dropped)
        if st != 0 and not any(dropped):
            N.check(st, "mm_render_status")
        if st != 0:
            raise RuntimeError("mm_render_backward: the texture-record pool overflowed (records dropped per image: %s); the texture gradients of those "
                               "images are NaN. Raise DiffRender.extra_texture_records_per_pixel." % (list(dropped),))

    def _proto(self, st, B, no_mask, Ht, Wt, contour=0.0, geometry_only=False):
        """(bytes of the MMRenderDesc prototype of this shape, its workspace size) for the C++ host path; cached like _desc's prototypes.
        contour: MMRenderDesc.fused_contour -- the C++ node copies the prototype for its forward AND its backward, so the weight travels in it."""
        key = ("bytes", id(st), B, int(bool(no_mask)), Ht, Wt, self.knum, self.sigmainv, self.boxlen, self.multiplier, self.eps, self.options,
               float(self.extra_texture_records_per_pixel), float(contour), bool(geometry_only))
        hit = self._desc_cache.get(key)
        if hit is None:
            d = N.MMRenderDesc()
            d.geometry_only = 1 if geometry_only else 0
            d.B, d.H, d.W, d.V, d.F = B, self.render_height, self.image_size, self.num_vertices, self.num_faces
            d.Ht, d.Wt = Ht, Wt
            d.no_mask, d.knum = int(bool(no_mask)), self.knum
            for i in range(3):
                d.proj[i] = float(self.cam_proj[i, 0])
            d.sigmainv, d.boxlen, d.multiplier, d.eps = self.sigmainv, self.boxlen, self.multiplier, self.eps
            d.faces, d.face_uvs = N.ptr(st["faces"]), N.ptr(st["face_uvs"])
            d.vc_table, d.vc_stride = N.ptr(st["vc_table"]), int(st["vc_table"].shape[1])
            d.options = self.options
            d.status_flag = self._status_ptr()
            d.fused_contour = float(contour)
            hit = (bytes(d), self.workspace_bytes(d))
            if len(self._desc_cache) > 32:
                self._desc_cache.clear()
            self._desc_cache[key] = hit
        return hit

    def _desc(self, st, B, no_mask, vertices, textures, lights, bg, azimuths, elevations, distances, biases, rgba, face_idx, fn, imn):
        """MMRenderDesc for one call.  The constant part (sizes, dibr constants, template pointers) is filled once per shape and
        copied; only the per-call pointers are set here (host time matters: this path is enqueue-bound)."""
        key = (id(st), B, int(bool(no_mask)), textures.shape[2], textures.shape[3], self.knum, self.sigmainv, self.boxlen, self.multiplier,
               self.eps, self.options)
        proto = self._desc_cache.get(key)
        if proto is None:
            proto = N.MMRenderDesc()
            proto.B, proto.H, proto.W, proto.V, proto.F = B, self.render_height, self.image_size, self.num_vertices, self.num_faces
            proto.Ht, proto.Wt = textures.shape[2], textures.shape[3]
            proto.no_mask, proto.knum = int(bool(no_mask)), self.knum
            for i in range(3):
                proto.proj[i] = float(self.cam_proj[i, 0])
            proto.sigmainv, proto.boxlen, proto.multiplier, proto.eps = self.sigmainv, self.boxlen, self.multiplier, self.eps
            proto.faces, proto.face_uvs = N.ptr(st["faces"]), N.ptr(st["face_uvs"])
            proto.vc_table, proto.vc_stride = N.ptr(st["vc_table"]), int(st["vc_table"].shape[1])
            proto.options = self.options
            proto.status_flag = self._status_ptr()
            if len(self._desc_cache) > 32:
                self._desc_cache.clear()
            self._desc_cache[key] = proto
        d = N.MMRenderDesc.from_buffer_copy(proto)
        dp = lambda t: None if t is None else t.data_ptr()
        d.vertices, d.textures, d.lights, d.bg = dp(vertices), dp(textures), dp(lights), dp(bg)
        d.azimuths, d.elevations, d.distances, d.biases = dp(azimuths), dp(elevations), dp(distances), dp(biases)
        d.rgba, d.face_idx, d.face_normals, d.imnormal = dp(rgba), dp(face_idx), dp(fn), dp(imn)
        return d

    # ---- networks.py:258-324 -------------------------------------------------------------------------------------
    def render(self, no_mask=False, **attributes):
        azimuths = attributes['azimuths']
        elevations = attributes['elevations']
        distances = attributes['distances']
        biases = attributes['biases']
        bg = attributes['bg']
        vertices = attributes['vertices']
        textures = attributes['textures']
        lights = attributes['lights']
        rgba, fn, imn, face_idx = self._render_node(bool(no_mask), None, vertices, textures, lights, bg if no_mask else None,
                                                    azimuths, elevations, distances, biases)
        rgbs = rgba.permute(0, 3, 1, 2)                 # (B,4,H,W) view of NHWC memory, like networks.py:317
        attributes['face_normals'] = fn
        attributes['imnormal'] = imn if self.emit_imnormal else None
        self.last_face_idx = face_idx                   # kaolin returns it from dibr_rasterization; the reference drops it
        return rgbs, attributes

    def render_many(self, attribute_sets, no_mask=False):
        """Several INDEPENDENT ``render`` calls as one: the attribute sets (same shapes) are concatenated along the batch and rendered by ONE pass
        of the kernels over sum(B) images -- the renders of trainer.py:276, :345 and :347 all exist as attributes before any of them runs, and
        three launches-bound passes of 48 images cost more than one of 144.  Returns [(rgbs, attributes), ...] like the separate calls would:
        per-set views of the batched outputs; gradients flow back through the concatenation into every set's own tensors.
        Results per image are those of the separate calls bit for bit (an image's render does not depend on its batch)."""
        sets = list(attribute_sets)
        if not sets:
            return []
        keys = ('vertices', 'textures', 'lights', 'azimuths', 'elevations', 'distances', 'biases') + (('bg',) if no_mask else ())
        sizes = [int(a['azimuths'].reshape(-1).shape[0]) for a in sets]
        cat = {k: (sets[0][k] if len(sets) == 1 else torch.cat([a[k].reshape((n,) + tuple(a[k].shape[1:]) if a[k].dim() > 1 else (n,)) for a, n in zip(sets, sizes)], 0))
               for k in keys}
        rgba, fn, imn, face_idx = self._render_node(bool(no_mask), None, cat['vertices'], cat['textures'], cat['lights'], cat.get('bg'),
                                                    cat['azimuths'], cat['elevations'], cat['distances'], cat['biases'])
        self.last_face_idx = face_idx
        rgba_p = _SplitBatchFn.apply(rgba, *sizes) if len(sets) > 1 else (rgba,)
        fn_p = _SplitBatchFn.apply(fn, *sizes) if len(sets) > 1 else (fn,)
        out, o = [], 0
        for a, n, r, f in zip(sets, sizes, rgba_p, fn_p):
            a['face_normals'] = f
            a['imnormal'] = imn[o:o + n] if self.emit_imnormal else None
            out.append((r.permute(0, 3, 1, 2), a))
            o += n
        return out

    def render_geometry(self, **attributes):
        """``render`` for the call site that throws the image away (trainer.py:367: ``_, Aire = diffRender.render(**Aire)`` keeps only
        ``attributes['face_normals']``): the vertex stage alone -- camera, prepare_vertices, face normals -- with its backward to vertices and
        camera; nothing is rasterised, shaded or stored.  Returns the attributes with 'face_normals' set (bit-identical to render's) and
        'imnormal' None."""
        a = attributes
        self._raise_if_records_were_dropped()
        N.require_device(a['azimuths'])
        ext = None if self.check_texture_records else N.torch_ext()
        if ext is not None and hasattr(ext, "render_geometry"):   # the C++ node (csrc/mm_torch_ext.cpp: GeometryNode): no Python in forward or backward
            dev = a['azimuths'].device
            tex = a['textures']
            proto, nbytes = self._proto(self._static(dev), a['azimuths'].numel(), False, tex.shape[2], tex.shape[3], 0.0, geometry_only=True)
            attributes['face_normals'] = ext.render_geometry(N.fn_addr("mm_render_forward"), N.fn_addr("mm_render_backward"), proto, nbytes,
                                                             a['vertices'], a['azimuths'], a['elevations'], a['distances'], a['biases'])
        else:
            attributes['face_normals'] = _RenderFn.apply(self, False, "geometry", None, a['vertices'], a['textures'], a['lights'], None,
                                                         a['azimuths'], a['elevations'], a['distances'], a['biases'], 0.0)
        attributes['imnormal'] = None
        return attributes

    def render_recon(self, gt_data, no_mask=False, contour=0, **attributes):
        """render(**attributes) and recon_data(rendered, gt_data, no_mask, contour) in ONE pass over the pixels: the loss terms
        are reduced while the image is shaded and its gradient is formed inside the backward kernels (no loss launches, no dL/drgba
        round trip) -- the path bench.py's `value` times, reachable from the class API.  Returns (loss, rgbs, attributes); ``rgbs``
        carries no gradient here (use render + recon_data if the image feeds anything else that is differentiated).
        contour > 0 (recon_data's contour term, networks.py:379-388; trainer.py:441 passes opt.lambda_contour): folded in as well when the
        image's height and width are multiples of 4 (then the term's two nearest-neighbour resamplings stay inside a screen tile); other
        sizes raise -- call render(...) and recon_data(..., contour=...) for those.  (The reference also prints the term's value there.)"""
        contour = float(contour)
        if contour < 0:
            contour = 0.0                                        # networks.py:379 `if contour>0`
        if contour > 0 and (self.render_height % 4 or self.image_size % 4):
            raise ValueError("render_recon folds the contour term in only for image sizes that are multiples of 4 (got %dx%d); call render(...) and "
                             "recon_data(..., contour=%r)" % (self.render_height, self.image_size, contour))
        a = attributes
        rgba, fn, imn, face_idx, loss = self._render_node(bool(no_mask), gt_data, a['vertices'], a['textures'], a['lights'],
                                                          a['bg'] if no_mask else None, a['azimuths'], a['elevations'], a['distances'], a['biases'],
                                                          contour)
        attributes['face_normals'] = fn
        attributes['imnormal'] = imn if self.emit_imnormal else None
        self.last_face_idx = face_idx
        return loss, rgba.permute(0, 3, 1, 2), attributes

    # ---- networks.py:364-390 -------------------------------------------------------------------------------------
    def recon_data(self, pred_data, gt_data, no_mask=False, contour=0):
        ext = N.torch_ext()
        if ext is None:
            return _ReconFn.apply(pred_data, gt_data, self.image_weight, contour)
        N.require_device(pred_data, gt_data)
        # (pred_data the untouched image of one of this process's renders: its backward is routed through that render's node -- defer_recon_fusion)
        return ext.recon_data(N.fn_addr("mm_recon_query_workspace"), N.fn_addr("mm_recon_data_forward"), N.fn_addr("mm_recon_data_backward"),
                              pred_data, gt_data, float(self.image_weight), float(contour), torch._C._cuda_getCurrentRawStream(pred_data.device.index),
                              N.fn_addr("mm_recon_data_totals"), bool(self.defer_recon_fusion))

    # ---- networks.py:326-362: seven means in one HIP launch per direction (att_loss.py / csrc/mm_attloss.hip); the chamfer
    # variant of the shape term is a HIP nearest-neighbour search + a differentiable gather ----------
    def recon_att(self, pred_att, target_att, L1=False, chamfer=False, azim=1):
        l = att_loss.attribute_losses(pred_att, target_att, L1)
        loss_cam = azim * l[att_loss.AZIM] + l[att_loss.ELEV] + l[att_loss.DIST]
        if chamfer:
            from .chamfer import chamfer_distance
            loss_shape, _ = chamfer_distance(pred_att['vertices'], target_att['vertices'])
        else:
            loss_shape = l[att_loss.SHAPE]
        return loss_cam, loss_shape, l[att_loss.TEXTURE], 0.1 * l[att_loss.LIGHT], l[att_loss.BIAS]

    # ---- mesh regularisers (networks.py:392-491): HIP kernels, one launch per direction (mesh_reg.py / csrc/mm_reg.hip) ----
    def _reg_tables(self, device):
        key = ("reg", str(device))
        tab = self._static_cache.get(key)
        if tab is None:
            tab = mesh_reg.build_tables(self, device)
            self._static_cache[key] = tab
        return tab

    def _reg(self, terms, vertices=None, delta=None, fn=None, temp=2.0, eps=0.001):
        return mesh_reg.MeshRegFn.apply(self, terms, temp, eps, vertices, delta, fn)

    def recon_flip(self, att, L1):                               # networks.py:392-410
        if L1:
            # the reference multiplies (B,V,3) by (B,V) here and raises (networks.py:409); pinned in tests/golden/losses.npz
            raise RuntimeError("The size of tensor a (3) must match the size of tensor b (%d) at non-singleton dimension 2" % self.num_vertices)
        return self._reg(mesh_reg.mask(mesh_reg.FLIP), delta=att['delta_vertices'])[mesh_reg.FLIP]

    def calc_reg_loss(self, att):                                # networks.py:412-451
        l = self._reg(mesh_reg.mask(mesh_reg.LAPLACIAN, mesh_reg.FLAT), delta=att['delta_vertices'], fn=att['face_normals'])
        return self.lambda_lpl * l[mesh_reg.LAPLACIAN] + self.lambda_flat * l[mesh_reg.FLAT]

    def calc_reg_edge(self, pred):                               # networks.py:453-461
        return self._reg(mesh_reg.mask(mesh_reg.EDGE), vertices=pred)[mesh_reg.EDGE]

    def calc_reg_depth(self, pred):                              # networks.py:463-466
        return self._reg(mesh_reg.mask(mesh_reg.DEPTH), vertices=pred)[mesh_reg.DEPTH]

    def calc_reg_depthR(self, pred, temp=2, eps=0.001):          # networks.py:468-475
        return self._reg(mesh_reg.mask(mesh_reg.DEPTHR), vertices=pred, temp=temp, eps=eps)[mesh_reg.DEPTHR]

    def calc_reg_depthC(self, pred, eps=0.001):                  # networks.py:477-485
        return self._reg(mesh_reg.mask(mesh_reg.DEPTHC), vertices=pred, eps=eps)[mesh_reg.DEPTHC]

    def calc_reg_deform(self, pred):                             # networks.py:487-491
        return self._reg(mesh_reg.mask(mesh_reg.DEFORM), delta=pred)[mesh_reg.DEFORM]

    def regularization(self, Ae, Ai, Aire, opt):
        """trainer.py:54-74 ``regularization(diffRender, Ae, Ai, Aire, opt)`` with every enabled mesh term of an attribute set in
        ONE launch (plus one for its backward) instead of one call per term: returns (lossR_reg, lossR_flip, lossR_IC)."""
        M = mesh_reg
        terms = [M.LAPLACIAN, M.FLAT, M.FLIP]
        for lam, t in ((opt.lambda_edge, M.EDGE), (opt.lambda_depth, M.DEPTH), (opt.lambda_depthR, M.DEPTHR),
                       (opt.lambda_depthC, M.DEPTHC), (opt.lambda_deform, M.DEFORM)):
            if lam > 0:
                terms.append(t)
        if opt.flipL1:
            self.recon_flip(Ae, True)                            # raises, like the reference
        le, li = [self._reg(M.mask(*terms), vertices=A['vertices'], delta=A['delta_vertices'], fn=A['face_normals'],
                            temp=opt.temp) for A in (Ae, Ai)]
        lr = self._reg(M.mask(M.FLIP), delta=Aire['delta_vertices'])
        lossR_reg = opt.lambda_reg * (self.lambda_lpl * (le[M.LAPLACIAN] + li[M.LAPLACIAN]) + self.lambda_flat * (le[M.FLAT] + li[M.FLAT])) / 2.0
        lossR_flip = opt.lambda_flipz * (le[M.FLIP] + li[M.FLIP] + lr[M.FLIP]) / 3.0
        for lam, t in ((opt.lambda_edge, M.EDGE), (opt.lambda_depth, M.DEPTH), (opt.lambda_depthR, M.DEPTHR),
                       (opt.lambda_depthC, M.DEPTHC), (opt.lambda_deform, M.DEFORM)):
            if lam > 0:
                lossR_reg = lossR_reg + lam * (le[t] + li[t]) / 2.0
        parts = self.recon_att(Aire, deep_copy(Ai, detach=True), L1=opt.L1, chamfer=opt.chamfer, azim=opt.azim)
        lossR_IC = opt.lambda_ic * (parts[0] + parts[1] + parts[2] + parts[3] + parts[4])
        return lossR_reg, lossR_flip, lossR_IC


def deep_copy(att, index=None, detach=False):
    """networks.py:146-161 (device follows the attributes instead of the hard-coded 'cuda')."""
    if index is None:
        index = torch.arange(att['distances'].shape[0], device=att['distances'].device)
    copy_att = {}
    for key, value in att.items():
        if key in ('azimuths', 'bg', 'biases', 'elevations', 'distances', 'vertices', 'delta_vertices', 'textures', 'lights'):
            if value is None:
                copy_att[key] = None
            elif detach:
                copy_att[key] = value[index].clone().detach()
            else:
                copy_att[key] = value[index].clone()
    return copy_att
