"""Locates the product package (``3d-magic-mirror_amd`` is not a valid identifier, so it is imported by name)."""
import importlib
import os
import sys

_ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
ops = importlib.import_module("3d-magic-mirror_amd.ops")
