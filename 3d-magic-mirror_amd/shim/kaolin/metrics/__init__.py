from . import render  # noqa: F401
