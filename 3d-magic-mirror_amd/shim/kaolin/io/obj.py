"""kaolin.io.obj (call sites /root/reference/networks.py:176, test.py:220-223)."""
from .._mm import ops

import_mesh = ops.import_mesh
