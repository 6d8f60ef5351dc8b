"""MI355X-native differentiable render + reconstruction-loss path for 3D-Magic-Mirror (gfx950 HIP kernels behind a
C ABI; the host side mirrors the reference's DiffRender API).  Import as ``importlib.import_module('3d-magic-mirror_amd')``
or through the root-level ``mm_amd`` alias."""
from . import mesh_reg, obj_io, template, texture_flow, synthetic  # noqa: F401
from .diff_render import DiffRender, deep_copy  # noqa: F401
from .obj_io import import_mesh, save_mesh  # noqa: F401
from .texture_flow import sample_texture  # noqa: F401
