"""ctypes binding of the C ABI in include/mm_render.h (lib/libmm_render.so).

The product path has NO fallback: if the shared library cannot be loaded, or a call returns a negative MMStatus, a
RuntimeError is raised.  torch is used here only for device memory and the current HIP stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmm_render.so")
_LIB = None

c_f = ctypes.c_float
c_i = ctypes.c_int32
c_p = ctypes.c_void_p


class MMRenderDesc(ctypes.Structure):
    _fields_ = [("B", c_i), ("H", c_i), ("W", c_i), ("V", c_i), ("F", c_i), ("Ht", c_i), ("Wt", c_i), ("no_mask", c_i),
                ("knum", c_i), ("proj", c_f * 3), ("sigmainv", c_f), ("boxlen", c_f), ("multiplier", c_f), ("eps", c_f),
                ("faces", c_p), ("face_uvs", c_p), ("vc_table", c_p), ("vc_stride", c_i),
                ("vertices", c_p), ("textures", c_p), ("lights", c_p), ("bg", c_p), ("azimuths", c_p), ("elevations", c_p),
                ("distances", c_p), ("biases", c_p),
                ("rgba", c_p), ("face_idx", c_p), ("face_normals", c_p), ("imnormal", c_p),
                ("workspace", c_p), ("workspace_bytes", ctypes.c_size_t), ("prof_events", c_p),
                ("fused_gt", c_p), ("fused_image_weight", c_f), ("fused_loss", c_p), ("fused_grad_loss", c_p),
                ("options", c_i), ("geometry_only", c_i), ("status_flag", c_p), ("fused_contour", c_f), ("fused_totals", c_p)]


class MMRenderGrads(ctypes.Structure):
    _fields_ = [("grad_rgba", c_p), ("grad_face_normals", c_p), ("grad_vertices", c_p), ("grad_textures", c_p),
                ("grad_lights", c_p), ("grad_bg", c_p), ("grad_azimuths", c_p), ("grad_elevations", c_p),
                ("grad_distances", c_p), ("grad_biases", c_p)]


class MMReconDesc(ctypes.Structure):
    _fields_ = [("B", c_i), ("H", c_i), ("W", c_i), ("pred", c_p), ("pred_strides", ctypes.c_int64 * 4), ("gt", c_p),
                ("image_weight", c_f), ("contour", c_f), ("loss", c_p), ("grad_loss", c_p), ("grad_pred", c_p),
                ("workspace", c_p), ("workspace_bytes", ctypes.c_size_t), ("prof_events", c_p)]

class MMMeshRegDesc(ctypes.Structure):
    _fields_ = [("B", c_i), ("V", c_i), ("F", c_i), ("E", c_i), ("terms", ctypes.c_uint32),
                ("lap_offsets", c_p), ("lap_cols", c_p), ("lap_vals", c_p), ("lapT_offsets", c_p), ("lapT_cols", c_p), ("lapT_vals", c_p),
                ("edges", c_p), ("edge2faces", c_p), ("ve_offsets", c_p), ("ve_items", c_p), ("fe_offsets", c_p), ("fe_items", c_p),
                ("flip_index", c_p), ("flipT_offsets", c_p), ("flipT_items", c_p), ("sign_init", c_p),
                ("vertices", c_p), ("delta_vertices", c_p), ("face_normals", c_p), ("ratio", c_f), ("temp", c_f), ("eps", c_f),
                ("losses", c_p), ("workspace", c_p), ("workspace_bytes", ctypes.c_size_t)]


class MMAttributes(ctypes.Structure):
    _fields_ = [(k, c_p) for k in ("azimuths", "elevations", "distances", "biases", "vertices", "textures", "lights")]


class MMAttLossDesc(ctypes.Structure):
    _fields_ = [("B", c_i), ("V", c_i), ("Ht", c_i), ("Wt", c_i), ("l1", c_i), ("pred", MMAttributes), ("target", MMAttributes),
                ("losses", c_p), ("workspace", c_p), ("workspace_bytes", ctypes.c_size_t)]


class MMAttLossGrads(ctypes.Structure):
    _fields_ = [("weights", c_p), ("pred", MMAttributes), ("target", MMAttributes)]


class MMTexFlowDesc(ctypes.Structure):
    _fields_ = [("B", c_i), ("C", c_i), ("H", c_i), ("W", c_i), ("Ho", c_i), ("Wo", c_i), ("image", c_p), ("flow", c_p), ("textures", c_p)]


class MMTexFlowGrads(ctypes.Structure):
    _fields_ = [("grad_textures", c_p), ("grad_flow", c_p), ("grad_image", c_p)]


class MMMeshRegGrads(ctypes.Structure):
    _fields_ = [("weights", c_p), ("grad_vertices", c_p), ("grad_delta_vertices", c_p), ("grad_face_normals", c_p)]


class MMPrepareDesc(ctypes.Structure):
    _fields_ = [("B", c_i), ("V", c_i), ("F", c_i), ("proj", c_f * 3), ("faces", c_p), ("vc_offsets", c_p), ("vc_items", c_p),
                ("vertices", c_p), ("transform", c_p), ("face_vertices_camera", c_p), ("face_vertices_image", c_p), ("face_normals", c_p),
                ("workspace", c_p), ("workspace_bytes", ctypes.c_size_t), ("proj_device", c_p)]


class MMPrepareGrads(ctypes.Structure):
    _fields_ = [("grad_face_vertices_camera", c_p), ("grad_face_vertices_image", c_p), ("grad_face_normals", c_p),
                ("grad_vertices", c_p), ("grad_transform", c_p)]


class MMDibrDesc(ctypes.Structure):
    _fields_ = [("B", c_i), ("H", c_i), ("W", c_i), ("F", c_i), ("D", c_i), ("knum", c_i), ("sigmainv", c_f), ("boxlen", c_f),
                ("multiplier", c_f), ("eps", c_f), ("face_vertices_z", c_p), ("face_vertices_image", c_p), ("face_features", c_p),
                ("face_normals_z", c_p), ("interpolated_features", c_p), ("soft_mask", c_p), ("face_idx", c_p),
                ("workspace", c_p), ("workspace_bytes", ctypes.c_size_t), ("options", c_i)]


class MMDibrGrads(ctypes.Structure):
    _fields_ = [("grad_interpolated_features", c_p), ("grad_soft_mask", c_p), ("grad_face_vertices_image", c_p), ("grad_face_features", c_p)]


class MMTexMapDesc(ctypes.Structure):
    _fields_ = [("B", c_i), ("N", c_i), ("C", c_i), ("Ht", c_i), ("Wt", c_i), ("mode", c_i), ("uv", c_p), ("textures", c_p), ("out", c_p)]


class MMTexMapGrads(ctypes.Structure):
    _fields_ = [("grad_out", c_p), ("grad_uv", c_p), ("grad_textures", c_p), ("workspace", c_p), ("workspace_bytes", ctypes.c_size_t)]


class MMShDesc(ctypes.Structure):
    _fields_ = [("B", c_i), ("N", c_i), ("normals", c_p), ("lights", c_p), ("out", c_p)]


class MMShGrads(ctypes.Structure):
    _fields_ = [("grad_out", c_p), ("grad_normals", c_p), ("grad_lights", c_p)]


class MMMaskIouDesc(ctypes.Structure):
    _fields_ = [("B", c_i), ("N", c_i), ("lhs", c_p), ("rhs", c_p), ("sums", c_p), ("loss", c_p)]


PROF_RENDER = ("vertex_fwd", "raster_fwd", "pixel_bwd", "gather_bwd", "vertex_bwd", "order")
ABI_VERSION = 6
OPT_WALK_BLOCK, OPT_WALK_WAVE = 1 << 1, 1 << 2
OPT_CULL_STRICT, OPT_SOFT_SKIP_CULLED, OPT_BBOX_HALF_OPEN, OPT_BARY_ONE_MINUS, OPT_SH_ORDER_XYZ = 1 << 4, 1 << 5, 1 << 6, 1 << 7, 1 << 8
OPT_BBOX_MIN_CLOSED_MAX_OPEN = 1 << 9
OPT_WALK_QUEUE, OPT_WALK_BATCH = 1 << 10, 1 << 11
OPT_MANY_IN_FLIGHT = 1 << 12          # hint: several independent calls in flight (identical results; include/mm_render.h)
PROF_RECON = ("recon_partial", "recon_final", "recon_bwd", "recon_contour")

EXPORTS = ("mm_query_workspace", "mm_render_forward", "mm_render_backward", "mm_render_status", "mm_render_fused_loss", "mm_debug_workspace_layout", "mm_recon_query_workspace",
           "mm_recon_data_forward", "mm_recon_data_backward", "mm_recon_data_totals", "mm_build_vertex_corner_csr", "mm_build_vertex_corner_csr_device", "mm_build_vertex_corner_table", "mm_nearest_neighbour", "mm_chamfer_nearest", "mm_status_string", "mm_last_error_detail",
           "mm_mesh_reg_query_workspace", "mm_mesh_reg_forward", "mm_mesh_reg_backward", "mm_texture_flow_forward",
           "mm_texture_flow_backward", "mm_attribute_loss_query_workspace", "mm_attribute_loss_forward",
           "mm_attribute_loss_backward",
           "mm_prepare_vertices_query_workspace", "mm_prepare_vertices_forward", "mm_prepare_vertices_backward",
           "mm_face_normals_forward", "mm_face_normals_backward", "mm_dibr_query_workspace", "mm_dibr_rasterization_forward",
           "mm_dibr_rasterization_backward", "mm_texture_mapping_forward", "mm_texture_mapping_backward", "mm_texture_mapping_backward_query_workspace", "mm_sh_lighting_forward",
           "mm_sh_lighting_backward", "mm_mask_iou_forward", "mm_mask_iou_backward", "mm_struct_size",
           "mm_abi_version")


_STRUCT_IDS = None


def lib():
    """Load libmm_render.so (building it in-tree first, under a file lock, if the sources are newer and hipcc is available).
    A failed rebuild raises -- a stale library is never used silently -- and the library's ABI version and struct sizes are
    checked against this binding's mirror of include/mm_render.h."""
    global _LIB
    if _LIB is not None:
        return _LIB
    from . import build_native
    if build_native.needs_build():
        if os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
            try:
                build_native.build()
            except Exception as e:
                raise RuntimeError("libmm_render.so is out of date and rebuilding it failed: %s" % e)
        elif os.path.exists(LIB_PATH):
            raise RuntimeError("libmm_render.so is older than its sources and hipcc is not available to rebuild it")
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libmm_render.so not found at %s (run __graft_entry__.build())" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(L, name):
            raise RuntimeError("libmm_render.so does not export %s" % name)
    L.mm_query_workspace.restype = ctypes.c_size_t
    L.mm_query_workspace.argtypes = [ctypes.POINTER(MMRenderDesc)]
    L.mm_render_forward.argtypes = [ctypes.POINTER(MMRenderDesc), c_p]
    L.mm_render_backward.argtypes = [ctypes.POINTER(MMRenderDesc), ctypes.POINTER(MMRenderGrads), c_p]
    L.mm_render_fused_loss.argtypes = [ctypes.POINTER(MMRenderDesc), c_p]
    L.mm_render_status.argtypes = [ctypes.POINTER(MMRenderDesc), c_p, ctypes.POINTER(ctypes.c_int32)]
    L.mm_recon_query_workspace.restype = ctypes.c_size_t
    L.mm_recon_query_workspace.argtypes = [ctypes.POINTER(MMReconDesc)]
    L.mm_recon_data_forward.argtypes = [ctypes.POINTER(MMReconDesc), c_p]
    L.mm_recon_data_backward.argtypes = [ctypes.POINTER(MMReconDesc), c_p]
    L.mm_recon_data_totals.restype = c_p
    L.mm_recon_data_totals.argtypes = [ctypes.POINTER(MMReconDesc)]
    L.mm_nearest_neighbour.argtypes = [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p]
    L.mm_chamfer_nearest.argtypes = [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p]
    L.mm_mesh_reg_query_workspace.restype = ctypes.c_size_t
    L.mm_mesh_reg_query_workspace.argtypes = [ctypes.POINTER(MMMeshRegDesc)]
    L.mm_mesh_reg_forward.argtypes = [ctypes.POINTER(MMMeshRegDesc), c_p]
    L.mm_mesh_reg_backward.argtypes = [ctypes.POINTER(MMMeshRegDesc), ctypes.POINTER(MMMeshRegGrads), c_p]
    L.mm_attribute_loss_query_workspace.restype = ctypes.c_size_t
    L.mm_attribute_loss_query_workspace.argtypes = [ctypes.POINTER(MMAttLossDesc)]
    L.mm_attribute_loss_forward.argtypes = [ctypes.POINTER(MMAttLossDesc), c_p]
    L.mm_attribute_loss_backward.argtypes = [ctypes.POINTER(MMAttLossDesc), ctypes.POINTER(MMAttLossGrads), c_p]
    L.mm_texture_flow_forward.argtypes = [ctypes.POINTER(MMTexFlowDesc), c_p]
    L.mm_texture_flow_backward.argtypes = [ctypes.POINTER(MMTexFlowDesc), ctypes.POINTER(MMTexFlowGrads), c_p]
    P = ctypes.POINTER
    L.mm_prepare_vertices_query_workspace.restype = ctypes.c_size_t
    L.mm_prepare_vertices_query_workspace.argtypes = [P(MMPrepareDesc)]
    L.mm_prepare_vertices_forward.argtypes = [P(MMPrepareDesc), c_p]
    L.mm_prepare_vertices_backward.argtypes = [P(MMPrepareDesc), P(MMPrepareGrads), c_p]
    L.mm_face_normals_forward.argtypes = [ctypes.c_int64, c_i, c_p, c_p, c_p]
    L.mm_face_normals_backward.argtypes = [ctypes.c_int64, c_i, c_p, c_p, c_p, c_p]
    L.mm_dibr_query_workspace.restype = ctypes.c_size_t
    L.mm_dibr_query_workspace.argtypes = [P(MMDibrDesc)]
    L.mm_dibr_rasterization_forward.argtypes = [P(MMDibrDesc), c_p]
    L.mm_dibr_rasterization_backward.argtypes = [P(MMDibrDesc), P(MMDibrGrads), c_p]
    L.mm_texture_mapping_forward.argtypes = [P(MMTexMapDesc), c_p]
    L.mm_texture_mapping_backward.argtypes = [P(MMTexMapDesc), P(MMTexMapGrads), c_p]
    L.mm_texture_mapping_backward_query_workspace.argtypes = [P(MMTexMapDesc)]
    L.mm_texture_mapping_backward_query_workspace.restype = ctypes.c_size_t
    L.mm_sh_lighting_forward.argtypes = [P(MMShDesc), c_p]
    L.mm_sh_lighting_backward.argtypes = [P(MMShDesc), P(MMShGrads), c_p]
    L.mm_mask_iou_forward.argtypes = [P(MMMaskIouDesc), c_p]
    L.mm_mask_iou_backward.argtypes = [P(MMMaskIouDesc), c_p, c_p, c_p, c_p]
    L.mm_struct_size.restype = ctypes.c_size_t
    L.mm_struct_size.argtypes = [ctypes.c_int]
    L.mm_build_vertex_corner_csr.argtypes = [c_i, c_i, c_p, c_p, c_p]
    L.mm_build_vertex_corner_csr_device.argtypes = [c_i, c_i, c_p, c_p, c_p, c_p, c_p]
    L.mm_status_string.restype = ctypes.c_char_p
    L.mm_status_string.argtypes = [ctypes.c_int]
    L.mm_last_error_detail.restype = ctypes.c_char_p
    L.mm_last_error_detail.argtypes = []
    L.mm_abi_version.restype = ctypes.c_int
    if L.mm_abi_version() != ABI_VERSION:
        raise RuntimeError("libmm_render.so has ABI version %d, this binding mirrors version %d" % (L.mm_abi_version(), ABI_VERSION))
    mirrors = (MMRenderDesc, MMRenderGrads, MMReconDesc, MMMeshRegDesc, MMMeshRegGrads, MMAttLossDesc, MMAttLossGrads, MMTexFlowDesc,
               MMTexFlowGrads, MMPrepareDesc, MMPrepareGrads, MMDibrDesc, MMDibrGrads, MMTexMapDesc, MMTexMapGrads, MMShDesc, MMShGrads,
               MMMaskIouDesc)
    for i, cls in enumerate(mirrors):
        if L.mm_struct_size(i) != ctypes.sizeof(cls):
            raise RuntimeError("struct layout mismatch for %s: library %d bytes, binding %d" % (cls.__name__, L.mm_struct_size(i), ctypes.sizeof(cls)))
    _LIB = L
    return L


def check(status, what):
    if status != 0:
        detail = lib().mm_last_error_detail().decode() if status == -4 else ""
        raise RuntimeError("%s failed: %s (MMStatus %d)%s" % (what, lib().mm_status_string(status).decode(), status,
                                                            " [" + detail + "]" if detail else ""))


_EXT = False


def torch_ext():
    """The optional C++ host path (lib/mm_torch_ext.so, built by build_native.build_torch_ext): the module, or None if it is not there or
    MM_NO_TORCH_EXT is set (then diff_render.py issues the same ABI calls from Python).  Never built lazily: 30 s of g++ do not belong in a
    first render call; __graft_entry__.build() builds it."""
    global _EXT
    if _EXT is False:
        _EXT = None
        from . import build_native as bn
        if not os.environ.get("MM_NO_TORCH_EXT") and os.path.exists(bn.EXT) and bn.ext_needs_build():
            # (r06: a header edit without a rebuild left the class API on its Python path, three times the host time per step, without a word)
            import warnings
            warnings.warn("lib/mm_torch_ext.so is older than its sources (csrc/mm_torch_ext.cpp, include/mm_render.h): the autograd API uses its "
                          "slower Python host path. Rebuild with `python __graft_entry__.py`.", RuntimeWarning, stacklevel=2)
        if not os.environ.get("MM_NO_TORCH_EXT") and os.path.exists(bn.EXT) and not bn.ext_needs_build():
            import importlib.machinery
            import importlib.util
            try:
                loader = importlib.machinery.ExtensionFileLoader("mm_torch_ext", bn.EXT)
                mod = importlib.util.module_from_spec(importlib.util.spec_from_loader("mm_torch_ext", loader))
                loader.exec_module(mod)
                if mod.desc_bytes() == ctypes.sizeof(MMRenderDesc):
                    _EXT = mod
            except Exception:                                      # a stale binary of another torch build: use the Python path
                _EXT = None
    return _EXT


_ADDR = {}


def fn_addr(name):
    """address of an exported function of the library, for the C++ host path"""
    a = _ADDR.get(name)
    if a is None:
        a = _ADDR[name] = ctypes.cast(getattr(lib(), name), ctypes.c_void_p).value
    return a


def as_f32(t, dev):
    """``t`` detached as a dense fp32 tensor on ``dev``; the common case (already so) costs one attribute test each."""
    if t is None:
        return None
    if t.dtype is torch.float32 and t.device == dev and t.is_contiguous():
        return t.detach()
    return t.detach().to(device=dev, dtype=torch.float32).contiguous()


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def current_stream(device):
    """The current HIP stream of ``device`` as a void* (the raw-handle query when torch offers it: the Stream object costs microseconds)."""
    if _raw_stream is not None and device.index is not None:
        return ctypes.c_void_p(_raw_stream(device.index))
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("the MI355X render path needs tensors in device memory (got a %s tensor); "
                               "there is no CPU fallback" % t.device)
