"""kaolin.render.mesh (call sites /root/reference/networks.py:284-306)."""
from .._mm import ops

prepare_vertices = ops.prepare_vertices
dibr_rasterization = ops.dibr_rasterization
texture_mapping = ops.texture_mapping
spherical_harmonic_lighting = ops.spherical_harmonic_lighting
