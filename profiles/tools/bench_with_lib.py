"""bench.py against a compile-time variant of the library (profiles/tools/variant_sweep.py build name=...):  python profiles/tools/bench_with_lib.py <name|base> [bench.py arguments]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
lib = sys.argv[1]; sys.argv = ["bench.py"] + sys.argv[2:]
pkg = importlib.import_module("3d-magic-mirror_amd")
if lib != "base":
    pkg._native.LIB_PATH = os.path.join(ROOT, "3d-magic-mirror_amd", "lib", "var_%s.so" % lib)
    importlib.import_module("3d-magic-mirror_amd.build_native").needs_build = lambda: False
import bench
bench.main()
