"""Importable alias for the package directory ``3d-magic-mirror_amd`` (not a valid Python identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("3d-magic-mirror_amd")
sys.modules[__name__] = _pkg
