"""``pytorch3d`` as far as /root/reference's training path uses it (``pytorch3d.loss.chamfer_distance``), on the MI355X
nearest-neighbour kernel: see ../README.md.  (template-change-animation.py's pytorch3d renderer is an offline visualisation
script outside the path and is not provided.)"""
from . import loss  # noqa: F401

__version__ = "0.7.0+mi355x"
