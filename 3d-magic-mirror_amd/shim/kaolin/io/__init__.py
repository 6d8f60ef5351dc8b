from . import obj  # noqa: F401
