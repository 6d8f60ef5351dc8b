"""Pre-planned render -> recon_data -> backward step over the C ABI, replayable as one HIP graph (RenderLossStep.capture / replay).

(Rounds 3-4 also carried captured steps behind autograd nodes -- DiffRender.graphed_step / graphed_render.  Removed in round 5: on this stack a
replayed 3-kernel graph costs the GPU more than the three launches it replaces and the host no less than the C++ eager node; measured
385 k images/s with zero-copy input slots and 297 k with inputs copied in, against 424 k for DiffRender.render_recon -- profiles/r05_api_paths.md.)

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

    def run_deferred(self, stream=None):
        """The un-fused step with DEFERRED fusion (MMRenderDesc.fused_totals, ABI 6): render, recon_data's forward on the image, then ONE backward
        call that forms dL/d image itself from recon_data's per-image totals -- no mm_recon_data_backward, no grad_rgba round trip, the same bits."""
        if self.fused or self.r.contour > 0:
            raise RuntimeError("run_deferred is the un-fused step without the contour term")
        L = N.lib()
        s = ctypes.c_void_p((stream or torch.cuda.current_stream(self.dev)).cuda_stream)
        N.check(L.mm_render_forward(ctypes.byref(self.d), s), "mm_render_forward")
        N.check(L.mm_recon_data_forward(ctypes.byref(self.r), s), "mm_recon_data_forward")
        d = self.render_desc()
        d.fused_gt, d.fused_image_weight, d.fused_grad_loss = N.ptr(self.gt), float(self.dr.image_weight), N.ptr(self.loss_scale)
        d.fused_totals = self.recon_totals_ptr()
        g = self.grads_struct()
        g.grad_rgba = None
        N.check(L.mm_render_backward(ctypes.byref(d), ctypes.byref(g), s), "mm_render_backward (deferred)")

    def render_desc(self):
        """a copy of this step's MMRenderDesc (tests poke at the copy)"""
        return N.MMRenderDesc.from_buffer_copy(self.d)

    def grads_struct(self):
        return N.MMRenderGrads.from_buffer_copy(self.g)

    def recon_totals_ptr(self):
        return N.lib().mm_recon_data_totals(ctypes.byref(self.r))

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
