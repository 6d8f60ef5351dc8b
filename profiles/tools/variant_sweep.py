"""A/B of compile-time variants of the library, one stream, per-kernel HIP-event times.
  build (no GPU needed):   python profiles/tools/variant_sweep.py build  name=-DFLAG=1,-DOTHER=2  name2=...
  run   (GPU box):         python profiles/tools/variant_sweep.py run config2 [config3 ...]
The variants live in 3d-magic-mirror_amd/lib/var_<name>.so (git-ignored, they travel with the gpurun snapshot); `base` is the product build."""
import sys, os, glob, importlib, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIBDIR = os.path.join(ROOT, "3d-magic-mirror_amd", "lib")

if sys.argv[1] == "build":
    bn = importlib.import_module("3d-magic-mirror_amd.build_native")
    for old in glob.glob(os.path.join(LIBDIR, "var_*.so")):
        os.remove(old)
    for spec in sys.argv[2:]:
        name, flags = spec.split("=", 1)
        bn.build(out=os.path.join(LIBDIR, "var_%s.so" % name), extra_flags=[f for f in flags.split(",") if f])
        print("built", name, flags)
else:
    cfgs = sys.argv[2:] or ["config2"]
    libs = [("base", "")] + [(os.path.basename(p)[4:-3], p) for p in sorted(glob.glob(os.path.join(LIBDIR, "var_*.so")))]
    for name, path in libs:
        env = dict(os.environ)
        if path:
            env["MM_DBG_LIB"] = path
        out = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "tools", "kernel_times.py")] + cfgs, env=env, capture_output=True, text=True)
        for line in out.stdout.splitlines():
            if line.startswith(("config", "market")) or ":" in line.split(" ")[0]:
                print("%-14s %s" % (name, line), flush=True)
        if out.returncode:
            print(name, "FAILED", out.stderr[-400:])
