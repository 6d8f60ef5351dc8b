"""Locates the product package (``3d-magic-mirror_amd`` is not a valid identifier, so it is imported by name)."""
import importlib
import os
import sys

_ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
if _ROOT not in sys.path:
    sys.path.append(_ROOT)                                       # behind the caller's own entries: the repo's top-level names (tests, oracle, bench, ...)
                                                                 # must not shadow the reference's modules
ops = importlib.import_module("3d-magic-mirror_amd.ops")
