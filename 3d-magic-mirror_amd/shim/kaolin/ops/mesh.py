"""kaolin.ops.mesh (call sites /root/reference/networks.py:201,249,289)."""
from .._mm import ops

index_vertices_by_faces = ops.index_vertices_by_faces
face_normals = ops.face_normals
uniform_laplacian = ops.uniform_laplacian
