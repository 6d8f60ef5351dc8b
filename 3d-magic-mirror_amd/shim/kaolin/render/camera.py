"""kaolin.render.camera (call site /root/reference/networks.py:174)."""
from .._mm import ops

generate_perspective_projection = ops.generate_perspective_projection
