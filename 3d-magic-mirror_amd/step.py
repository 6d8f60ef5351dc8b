"""Pre-planned render -> recon_data -> backward step over the C ABI, replayable as one HIP graph.

The autograd wrappers in diff_render.py allocate per call and cross the Python/torch boundary four times per step;
for a training loop (and for bench.py) the same four ABI calls are issued here against buffers allocated once, and
captured into a HIP graph so that one step costs one graph launch on the host.  Work per step is exactly
mm_render_forward + mm_recon_data_forward + mm_recon_data_backward + mm_render_backward (no skipped stage).
"""
import ctypes

import torch

from . import _native as N

LEAVES = ("vertices", "textures", "lights", "bg", "azimuths", "elevations", "distances", "biases")


class HipEvents:
    """Raw hipEvent_t pairs handed to the library's prof_events hook (timed on the stream the kernels run on)."""

    def __init__(self, nslots):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.n = nslots
        self.arr = (ctypes.c_void_p * (2 * nslots))()
        for i in range(2 * nslots):
            ev = ctypes.c_void_p()
            rc = self.hip.hipEventCreate(ctypes.byref(ev))
            if rc != 0:
                raise RuntimeError("hipEventCreate failed (%d)" % rc)
            self.arr[i] = ev

    def ptr(self):
        return ctypes.cast(self.arr, ctypes.c_void_p)

    def elapsed_ms(self, slot):
        ms = ctypes.c_float()
        rc = self.hip.hipEventElapsedTime(ctypes.byref(ms), self.arr[2 * slot], self.arr[2 * slot + 1])
        if rc != 0:
            self.hip.hipGetLastError()          # a slot the library did not run (never-recorded events): do not leave the error
            return float("nan")                 # behind for the caller's next runtime call to trip over
        return ms.value


class RenderLossStep:
    def __init__(self, dr, attributes, gt, no_mask=True, contour=0.0, emit_imnormal=False, loss_scale=None, fused=False):
        dev = attributes["azimuths"].device
        N.require_device(*[attributes[k] for k in LEAVES if attributes.get(k) is not None], gt)
        self.dr, self.dev, self.no_mask = dr, dev, bool(no_mask)
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        self.inp = {k: (f32(attributes[k]) if attributes.get(k) is not None else None) for k in LEAVES}
        self.gt = f32(gt)
        B = self.inp["azimuths"].shape[0]
        H, W = dr.render_height, dr.image_size
        self.B, self.H, self.W = B, H, W
        st = dr._static(dev)
        self.rgba = torch.empty((B, H, W, 4), device=dev)
        self.face_idx = torch.empty((B, H, W), device=dev, dtype=torch.int32)
        self.face_normals = torch.empty((B, dr.num_faces, 3), device=dev)
        self.imnormal = torch.empty((B, H, W, 3), device=dev) if emit_imnormal else None
        i = self.inp
        self.d = dr._desc(st, B, no_mask, i["vertices"], i["textures"], i["lights"], i["bg"] if no_mask else None, i["azimuths"],
                          i["elevations"], i["distances"], i["biases"], self.rgba, self.face_idx, self.face_normals, self.imnormal)
        self.ws = torch.empty(dr.workspace_bytes(self.d), device=dev, dtype=torch.uint8)
        self.d.workspace, self.d.workspace_bytes = N.ptr(self.ws), self.ws.numel()
        self.grad_rgba = torch.empty_like(self.rgba)
        self.grads = {k: (torch.empty_like(v) if v is not None and (k != "bg" or no_mask) else None) for k, v in self.inp.items()}
        g = self.grads
        self.g = N.MMRenderGrads(N.ptr(self.grad_rgba), None, N.ptr(g["vertices"]), N.ptr(g["textures"]), N.ptr(g["lights"]),
                                 N.ptr(g["bg"]), N.ptr(g["azimuths"]), N.ptr(g["elevations"]), N.ptr(g["distances"]), N.ptr(g["biases"]))
        self.loss = torch.zeros((), device=dev)
        r = N.MMReconDesc()
        r.B, r.H, r.W = B, H, W
        r.pred, r.gt = N.ptr(self.rgba), N.ptr(self.gt)
        for k, s in enumerate((H * W * 4, 1, W * 4, 4)):
            r.pred_strides[k] = s
        r.image_weight, r.contour = float(dr.image_weight), float(contour)
        # loss_scale: dL/dloss, e.g. B_rank/B_global when several steps share one batch mean
        self.loss_scale = None if loss_scale is None else torch.full((), float(loss_scale), device=dev)
        r.loss, r.grad_loss, r.grad_pred = N.ptr(self.loss), N.ptr(self.loss_scale), N.ptr(self.grad_rgba)
        self.rws = torch.empty(N.lib().mm_recon_query_workspace(ctypes.byref(r)), device=dev, dtype=torch.uint8)
        r.workspace, r.workspace_bytes = N.ptr(self.rws), self.rws.numel()
        self.r = r
        # fused mode: recon_data folded into the render kernels (MMRenderDesc.fused_*): two ABI calls per step.  Its contour term folds in for
        # image sizes that are multiples of 4 (MMRenderDesc.fused_contour); other sizes with contour > 0 keep the three recon_data launches.
        self.fused = bool(fused) and not (contour > 0 and (H % 4 or W % 4))
        if self.fused:
            self.d.fused_contour = max(0.0, float(contour))
            self.d.fused_gt = N.ptr(self.gt)
            self.d.fused_image_weight = float(dr.image_weight)
            self.d.fused_loss = N.ptr(self.loss)
            self.d.fused_grad_loss = N.ptr(self.loss_scale)
        self.graph = None
        self.ev_render = self.ev_recon = None

    def set_inputs(self, attributes, gt):
        """Point the step at another batch of the SAME shapes (device tensors, fp32, dense): only the descriptors' input
        pointers change, every output / scratch buffer is reused.  Lets a loop rotate through more input data than the
        256 MiB Infinity Cache holds without one workspace per batch.  Not valid between capture() and replay()."""
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        new = {k: (f32(attributes[k]) if attributes.get(k) is not None else None) for k in LEAVES}
        for k, v in new.items():
            old = self.inp[k]
            if (v is None) != (old is None) or (v is not None and (v.shape != old.shape or v.device != old.device)):
                raise RuntimeError("set_inputs: '%s' does not match the planned shape" % k)
        gt = f32(gt)
        if gt.shape != self.gt.shape or gt.device != self.gt.device:
            raise RuntimeError("set_inputs: gt does not match the planned shape")
        self.inp, self.gt = new, gt
        i, d = self.inp, self.d
        d.vertices, d.textures, d.lights = N.ptr(i["vertices"]), N.ptr(i["textures"]), N.ptr(i["lights"])
        d.bg = N.ptr(i["bg"] if self.no_mask else None)
        d.azimuths, d.elevations, d.distances, d.biases = N.ptr(i["azimuths"]), N.ptr(i["elevations"]), N.ptr(i["distances"]), N.ptr(i["biases"])
        self.r.gt = N.ptr(self.gt)
        if self.fused:
            d.fused_gt = N.ptr(self.gt)

    def run(self, stream=None):
        """Enqueue one full step on ``stream`` (a torch.cuda.Stream; default: the current stream)."""
        L = N.lib()
        s = ctypes.c_void_p((stream or torch.cuda.current_stream(self.dev)).cuda_stream)
        N.check(L.mm_render_forward(ctypes.byref(self.d), s), "mm_render_forward")
        if not self.fused:
            N.check(L.mm_recon_data_forward(ctypes.byref(self.r), s), "mm_recon_data_forward")
            N.check(L.mm_recon_data_backward(ctypes.byref(self.r), s), "mm_recon_data_backward")
        N.check(L.mm_render_backward(ctypes.byref(self.d), ctypes.byref(self.g), s), "mm_render_backward")

    def run_forward(self, stream=None):
        """The forward half of run() (fused mode): render + recon_data value, on ``stream``."""
        L = N.lib()
        s = ctypes.c_void_p((stream or torch.cuda.current_stream(self.dev)).cuda_stream)
        N.check(L.mm_render_forward(ctypes.byref(self.d), s), "mm_render_forward")
        if self.fused:
            N.check(L.mm_render_fused_loss(ctypes.byref(self.d), s), "mm_render_fused_loss")
        else:
            N.check(L.mm_recon_data_forward(ctypes.byref(self.r), s), "mm_recon_data_forward")

    def run_backward(self, stream=None):
        L = N.lib()
        s = ctypes.c_void_p((stream or torch.cuda.current_stream(self.dev)).cuda_stream)
        if not self.fused:
            N.check(L.mm_recon_data_backward(ctypes.byref(self.r), s), "mm_recon_data_backward")
        N.check(L.mm_render_backward(ctypes.byref(self.d), ctypes.byref(self.g), s), "mm_render_backward")

    def dropped_records(self, stream=None):
        """Per-image counts of texture-gradient records the last backward had no room for (mm_render_status; synchronises)."""
        out = (ctypes.c_int32 * self.d.B)()
        N.lib().mm_render_status(ctypes.byref(self.d), ctypes.c_void_p((stream or torch.cuda.current_stream(self.dev)).cuda_stream), out)
        return list(out)

    def capture(self):
        """Capture run() into a HIP graph (torch.cuda.CUDAGraph is only the capture/replay plumbing)."""
        side = torch.cuda.Stream(self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            self.run(side)                      # warm-up outside capture (module load, first-touch)
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            self.run(side)
        return self.graph

    def replay(self):
        self.graph.replay()

    # ---- per-kernel timing through the library's event hook (eager launches only) ------------------------------
    def enable_profiling(self):
        self.ev_render = HipEvents(len(N.PROF_RENDER))
        self.ev_recon = HipEvents(len(N.PROF_RECON))
        self.d.prof_events = self.ev_render.ptr()
        self.r.prof_events = self.ev_recon.ptr()

    def disable_profiling(self):
        self.d.prof_events = None
        self.r.prof_events = None

    def kernel_times_ms(self):
        """Call after run() + synchronize with profiling enabled."""
        out = {name: self.ev_render.elapsed_ms(i) for i, name in enumerate(N.PROF_RENDER)}
        if not self.fused:
            out.update({name: self.ev_recon.elapsed_ms(i) for i, name in enumerate(N.PROF_RECON) if name != "recon_contour" or self.r.contour > 0})
        return out


def _leaf_has_hooks(t):
    """A tensor hook or a post-accumulate-grad hook is registered on the leaf: its gradient must reach the engine as a tensor of its own."""
    return bool(getattr(t, "_backward_hooks", None)) or bool(getattr(t, "_post_accumulate_grad_hooks", None))


def _graphed_input_grads(owner, leaf_inputs, shapes, g):
    """What a graphed node hands back for its eight attribute inputs after the backward graph has run (g: the static gradient buffers).
    Every input gets its gradient THROUGH THE ENGINE in the input's own shape:
      * a non-leaf input: a view of the static buffer (its producer consumes it within this backward pass);
      * a LEAF without hooks (what ``loss.backward()`` of a training loop meets: the consumer is the leaf's AccumulateGrad): a view of the static
        buffer as well -- AccumulateGrad takes a gradient tensor nobody else holds as ``.grad`` without copying it, so ``.grad`` is the static
        memory until the next replay, exactly as the class docstrings say (and as torch.cuda.make_graphed_callables behaves); a ``.grad`` that
        is KEPT across steps is copied out before the next replay (``_GraphedFn.backward``), so accumulation keeps its meaning;
      * a leaf WITH a tensor hook / post-accumulate hook, or every leaf of an object built with ``copy_leaf_grads=True``: a private copy (the hook
        may keep or modify what it is handed; with the flag nothing the caller holds ever aliases static memory -- round 4's default).
    Round 4 cloned for every leaf: eight copy launches per step (30 MB at B=48, 128x128) that made the captured path the slowest way to call
    the class.  ``torch.autograd.grad(loss, leaf)`` through this node returns the view: static memory, overwritten by the next call -- the
    contract of every output of a graphed object.
    ``owner.fast_leaf_grads`` (opt-in) bypasses the engine altogether: a leaf's ``.grad`` is assigned the static buffer (or accumulated into) and
    the engine gets None -- hooks do not fire and ``torch.autograd.grad`` sees no gradient on that path."""
    out = []
    for k, leaf, shp in zip(LEAVES, leaf_inputs, shapes):
        if g[k] is None or shp is None:
            out.append(None)
        elif leaf is None:
            out.append(g[k].reshape(shp))                        # a non-leaf input: consumed by the producer's backward inside this pass
        elif not owner.fast_leaf_grads:
            gk = g[k].reshape(shp)
            out.append(gk.clone() if (owner.copy_leaf_grads or _leaf_has_hooks(leaf)) else gk)
        else:
            gk = g[k].reshape(shp)
            if leaf.grad is None:
                leaf.grad = gk
            else:
                leaf.grad.add_(gk)
            out.append(None)
    return out


class _GraphedFn(torch.autograd.Function):
    """loss = graphed(leaves...): the forward replays the captured render + recon_data graph, the backward the captured backward graph."""

    @staticmethod
    def forward(ctx, gs, gt, *leaves):
        gs._load_inputs(leaves, gt)
        gs.fwd_graph.replay()
        ctx.gs = gs
        # (fast_leaf_grads only: inputs that are autograd LEAVES get their gradient assigned directly, see _graphed_input_grads)
        ctx.leaf_inputs = [t if (t is not None and t.is_leaf and t.requires_grad) else None for t in leaves]
        ctx.shapes = [None if t is None else tuple(t.shape) for t in leaves]
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(gs.step.face_idx)
        # loss, face_normals: differentiable outputs; the image carries no gradient (its only consumer, the loss, is inside)
        return gs.step.loss, gs.step.face_normals

    @staticmethod
    def backward(ctx, g_loss, g_fn):
        gs = ctx.gs
        g = gs.step.grads
        # A leaf whose .grad still aliases the static gradient buffer (kept from the previous step instead of being reset to None) is about
        # to be overwritten in place by the replay: give it a private copy first, so that accumulation keeps its meaning.
        for k, leaf in zip(LEAVES, ctx.leaf_inputs):
            if leaf is not None and leaf.grad is not None and g[k] is not None and leaf.grad.data_ptr() == g[k].data_ptr():
                leaf.grad = leaf.grad.clone()
        gs._load_upstream(g_loss, g_fn)
        gs.bwd_graph.replay()
        return (None, None) + tuple(_graphed_input_grads(gs, ctx.leaf_inputs, ctx.shapes, g))


class GraphedRenderRecon:
    """``DiffRender.render`` + ``DiffRender.recon_data`` (contour = 0) + their backward as TWO captured HIP graphs behind one autograd
    node: what trainer.py does with ``Xer, Ae = diffRender.render(**Ae)`` (:276), ``diffRender.recon_data(Xer, Xa)`` (:441) and
    ``lossR.backward()`` (:509-518) costs the host two graph launches per step instead of ~12 kernel launches, ~25 allocations and two
    autograd nodes of descriptor plumbing -- the eager class API is bound by exactly that host time (0.18 ms per step against 0.105 ms of
    GPU time at B=48, 128x128).

    The library never allocates and never synchronises, so the capture is plain: fixed device buffers for the eight attribute tensors and
    the target ("static input slots", ``self.inputs`` / ``self.gt``), for every output and every gradient.  A call copies its arguments
    into the slots (skipped for an argument that IS the slot: a caller that lets its networks write into ``inputs[...]`` pays no copy),
    replays the forward graph and returns ``(loss, rgbs, attributes)`` like ``render_recon``; ``loss.backward()`` replays the backward
    graph.  As with torch.cuda.make_graphed_callables, outputs and gradients live in static memory: they are overwritten by the next call.
    The ``.grad`` of an attribute that is an autograd leaf IS that static memory until then (reset it to None between steps, as
    ``optimizer.zero_grad()`` does by default; a ``.grad`` that is kept is copied out first, so accumulation over steps stays correct).  Results are bit-identical to the eager path (same kernels, same launch order)."""

    def __init__(self, dr, example_attributes, gt, no_mask=True, fast_leaf_grads=False, copy_leaf_grads=False):
        dev = example_attributes["azimuths"].device
        f32 = lambda t: t.detach().to(torch.float32).contiguous().clone()
        self.dr, self.dev, self.no_mask = dr, dev, bool(no_mask)
        self.fast_leaf_grads = bool(fast_leaf_grads)             # see _graphed_input_grads
        self.copy_leaf_grads = bool(copy_leaf_grads)
        self.inputs = {k: (f32(example_attributes[k]) if example_attributes.get(k) is not None and (k != "bg" or no_mask) else None) for k in LEAVES}
        self.gt = f32(gt)
        self.step = RenderLossStep(dr, self.inputs, self.gt, no_mask=no_mask, emit_imnormal=dr.emit_imnormal, fused=True)
        # upstream gradients: static slots the backward graph reads (dL/dloss, dL/dface_normals)
        self.g_loss = torch.ones((), device=dev, dtype=torch.float32)
        self.g_fn = torch.zeros_like(self.step.face_normals)
        self._g_fn_zero = True
        self.step.loss_scale = self.g_loss
        self.step.d.fused_grad_loss = N.ptr(self.g_loss)
        self.step.g.grad_face_normals = N.ptr(self.g_fn)
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self.step.run_forward(side); self.step.run_backward(side)          # warm-up outside capture (module load, first touch)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.fwd_graph, self.bwd_graph = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.fwd_graph, stream=side):
            self.step.run_forward(side)
        with torch.cuda.graph(self.bwd_graph, stream=side):
            self.step.run_backward(side)

    def _load_inputs(self, leaves, gt):
        for k, t in zip(LEAVES, leaves):
            slot = self.inputs[k]
            if slot is None or t is None:
                continue
            if t.data_ptr() != slot.data_ptr():
                slot.copy_(t.detach().reshape(slot.shape), non_blocking=True)
        if gt is not None and gt.data_ptr() != self.gt.data_ptr():
            self.gt.copy_(gt.detach(), non_blocking=True)

    def _load_upstream(self, g_loss, g_fn):
        if g_loss is None:
            self.g_loss.zero_()                                  # the loss took no part in what is differentiated: zero, never one
        elif g_loss.data_ptr() != self.g_loss.data_ptr():
            self.g_loss.copy_(g_loss.detach().reshape(()), non_blocking=True)
        if g_fn is not None:
            self.g_fn.copy_(g_fn.detach(), non_blocking=True); self._g_fn_zero = False
        elif not self._g_fn_zero:
            self.g_fn.zero_(); self._g_fn_zero = True

    def __call__(self, gt_data=None, **attributes):
        """(loss, rgbs, attributes) = render_recon(gt_data, no_mask, **attributes) through the captured graphs."""
        leaves = tuple(attributes.get(k) if (k != "bg" or self.no_mask) else None for k in LEAVES)
        loss, fn = _GraphedFn.apply(self, gt_data, *leaves)
        attributes["face_normals"] = fn
        attributes["imnormal"] = self.step.imnormal
        self.dr.last_face_idx = self.step.face_idx
        return loss, self.step.rgba.permute(0, 3, 1, 2), attributes

    def run(self):
        """No autograd at all: replay forward + backward on what the slots hold (dL/dloss = 1, dL/dface_normals = 0); gradients in ``self.grads``."""
        self.g_loss.fill_(1.0)                                   # (whatever an earlier autograd backward left in the upstream slots)
        if not self._g_fn_zero:
            self.g_fn.zero_(); self._g_fn_zero = True
        self.fwd_graph.replay(); self.bwd_graph.replay()
        return self.step.loss

    @property
    def grads(self):
        return self.step.grads


class _GraphedRenderFn(torch.autograd.Function):
    """rgba (B,H,W,4), face_normals = graphed render(leaves...): forward and backward are one captured graph each."""

    @staticmethod
    def forward(ctx, gr, *leaves):
        gr._load_inputs(leaves)
        gr.fwd_graph.replay()
        ctx.gr = gr
        ctx.leaf_inputs = [t if (t is not None and t.is_leaf and t.requires_grad) else None for t in leaves]
        ctx.shapes = [None if t is None else tuple(t.shape) for t in leaves]
        ctx.set_materialize_grads(False)
        return gr.step.rgba, gr.step.face_normals

    @staticmethod
    def backward(ctx, g_rgba, g_fn):
        gr = ctx.gr
        g = gr.step.grads
        for k, leaf in zip(LEAVES, ctx.leaf_inputs):             # (see _GraphedFn.backward: a kept .grad that aliases the static buffer is copied out first)
            if leaf is not None and leaf.grad is not None and g[k] is not None and leaf.grad.data_ptr() == g[k].data_ptr():
                leaf.grad = leaf.grad.clone()
        if g_rgba is None:
            if not gr._g_rgba_zero:
                gr.step.grad_rgba.zero_(); gr._g_rgba_zero = True
        else:
            gr.step.grad_rgba.copy_(g_rgba.detach(), non_blocking=True); gr._g_rgba_zero = False   # (any strides: the image reaches the caller as a permuted view)
        if g_fn is not None:
            gr.g_fn.copy_(g_fn.detach(), non_blocking=True); gr._g_fn_zero = False
        elif not gr._g_fn_zero:
            gr.g_fn.zero_(); gr._g_fn_zero = True
        gr.bwd_graph.replay()
        return (None,) + tuple(_graphed_input_grads(gr, ctx.leaf_inputs, ctx.shapes, g))


class GraphedRender:
    """``DiffRender.render`` alone (no loss folded in) and its backward as two captured HIP graphs behind one autograd node: the render calls of
    a trainer iteration whose images feed something other than ``recon_data`` -- trainer.py:345-367 renders three more views per iteration for the
    discriminator and the cycle losses.  Same contract as GraphedRenderRecon: static input slots (``inputs``), static outputs (the image, the
    face normals, face_idx and imnormal are overwritten by the next call of THIS object: use one object per render of an iteration), leaves get
    the static gradient buffers as ``.grad``.  The upstream gradient of the image is copied into a static slot (50 MB at B=48, 256x256: the one
    copy this path cannot avoid -- the loss lives outside).  Bit-identical to the eager render."""

    def __init__(self, dr, example_attributes, no_mask=True, fast_leaf_grads=False, copy_leaf_grads=False):
        dev = example_attributes["azimuths"].device
        f32 = lambda t: t.detach().to(torch.float32).contiguous().clone()
        self.dr, self.dev, self.no_mask = dr, dev, bool(no_mask)
        self.fast_leaf_grads = bool(fast_leaf_grads)             # see _graphed_input_grads
        self.copy_leaf_grads = bool(copy_leaf_grads)
        self.inputs = {k: (f32(example_attributes[k]) if example_attributes.get(k) is not None and (k != "bg" or no_mask) else None) for k in LEAVES}
        B, H, W = self.inputs["azimuths"].shape[0], dr.render_height, dr.image_size
        self.step = RenderLossStep(dr, self.inputs, torch.zeros((B, 4, H, W), device=dev), no_mask=no_mask, emit_imnormal=dr.emit_imnormal, fused=False)
        self.g_fn = torch.zeros_like(self.step.face_normals)
        self.step.g.grad_face_normals = N.ptr(self.g_fn)
        self.step.grad_rgba.zero_()
        self._g_fn_zero = self._g_rgba_zero = True
        L = N.lib()
        fwd = lambda s: N.check(L.mm_render_forward(ctypes.byref(self.step.d), ctypes.c_void_p(s.cuda_stream)), "mm_render_forward")
        bwd = lambda s: N.check(L.mm_render_backward(ctypes.byref(self.step.d), ctypes.byref(self.step.g), ctypes.c_void_p(s.cuda_stream)), "mm_render_backward")
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            fwd(side); bwd(side)                                 # warm-up outside capture (module load, first touch)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.fwd_graph, self.bwd_graph = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.fwd_graph, stream=side):
            fwd(side)
        with torch.cuda.graph(self.bwd_graph, stream=side):
            bwd(side)

    def _load_inputs(self, leaves):
        for k, t in zip(LEAVES, leaves):
            slot = self.inputs[k]
            if slot is None or t is None:
                continue
            if t.data_ptr() != slot.data_ptr():
                slot.copy_(t.detach().reshape(slot.shape), non_blocking=True)

    def __call__(self, **attributes):
        """(rgbs, attributes) = render(no_mask, **attributes) through the captured graphs."""
        leaves = tuple(attributes.get(k) if (k != "bg" or self.no_mask) else None for k in LEAVES)
        rgba, fn = _GraphedRenderFn.apply(self, *leaves)
        attributes["face_normals"] = fn
        attributes["imnormal"] = self.step.imnormal
        self.dr.last_face_idx = self.step.face_idx
        return rgba.permute(0, 3, 1, 2), attributes
