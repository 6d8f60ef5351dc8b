"""Runs GPU parity tests against a compile-time variant of the library (profiles/tools/variant_sweep.py build name=-DFLAG=...):
   python profiles/tools/run_variant_tests.py <name> [pytest -k expression]
e.g. `smallplan=-DMM_PLAN_LDS_FACES=1024` sends every template through the plan's non-staged path (meshes of more than 14 336 faces)."""
import sys, importlib, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3d-magic-mirror_amd")
pkg._native.LIB_PATH = os.path.join(ROOT, "3d-magic-mirror_amd", "lib", "var_%s.so" % sys.argv[1])
importlib.import_module("3d-magic-mirror_amd.build_native").needs_build = lambda: False
import pytest
sys.exit(pytest.main([os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q", "-x"] + (["-k", sys.argv[2]] if len(sys.argv) > 2 else [])))
