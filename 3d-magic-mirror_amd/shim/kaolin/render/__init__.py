from . import camera, mesh  # noqa: F401
