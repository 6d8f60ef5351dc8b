"""``kaolin`` as far as /root/reference uses it (v0.12.0 names), on the MI355X kernels: see ../README.md."""
from . import io, metrics, ops, render  # noqa: F401

__version__ = "0.12.0+mi355x"
