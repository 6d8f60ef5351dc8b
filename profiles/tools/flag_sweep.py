"""Compiler-flag A/B of the render path's four hot translation units (r06, third session): backend scheduling strategies, clause lengths,
optimisation levels, xnack-specific code objects.  None of them changes an fp32 expression (no reassociation, no contraction), so results stay
bit-identical; adopted ones still go through the GPU suite.
  build (no GPU):  python profiles/tools/flag_sweep.py build              -> 3d-magic-mirror_amd/lib/var_<name>.so for every entry of VARIANTS
  run (GPU box):   python profiles/tools/variant_sweep.py run config2 config3 config5 config2x8
Objects of the files a variant does not touch are compiled once (lib/flag_sweep_base/) and re-linked."""
import sys, os, glob, importlib, subprocess, shutil
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
bn = importlib.import_module("3d-magic-mirror_amd.build_native")
LIBDIR = os.path.join(ROOT, "3d-magic-mirror_amd", "lib")
BASE = os.path.join(LIBDIR, "flag_sweep_base")
HOT = ["mm_raster.hip", "mm_pixel_bwd.hip", "mm_backward.hip", "mm_vertex.hip"]
ML = lambda *o: [x for opt in o for x in ("-mllvm", opt)]
VARIANTS = {
    "rebuilt": [],                                               # the product's flags through this tool (must equal `base`)
    "maxilp": ML("-amdgpu-sched-strategy=max-ilp"),
    "maxclause": ML("-amdgpu-sched-strategy=max-memory-clause"),
    "iterminreg": ML("-amdgpu-sched-strategy=iterative-minreg"),
    "itermaxocc": ML("-amdgpu-sched-strategy=iterative-maxocc"),
    "iterilp": ML("-amdgpu-sched-strategy=iterative-ilp"),
    "bias100": ML("-amdgpu-schedule-metric-bias=100"),
    "bias0": ML("-amdgpu-schedule-metric-bias=0"),
    "wprio": ML("-amdgpu-set-wave-priority"),
    "trackers": ML("-amdgpu-use-amdgpu-trackers"),
    "nohighrp": ML("-amdgpu-disable-unclustered-high-rp-reschedule"),
    "relaxocc": ML("-amdgpu-schedule-relaxed-occupancy"),
    "clause4": ML("-amdgpu-max-memory-clause=4"),
    "clause32": ML("-amdgpu-max-memory-clause=32"),
    "nopostsched": ML("-enable-post-misched=0"),
    "nomisched": ML("-enable-misched=0"),
    "O2": ["-O2"],
    "Os": ["-Os"],
    "nounroll": ["-fno-unroll-loops"],
    "xnackoff": ["--offload-arch=gfx950:xnack-"],                # replaces gfx950 (see build_variant)
}


def compile_one(src, flags, obj):
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + ["-c", os.path.join(bn.CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    return r.returncode, r.stderr[-600:]


def build_variant(name, extra, files=HOT):
    d = os.path.join(LIBDIR, "flag_sweep_" + name)
    os.makedirs(d, exist_ok=True)
    objs, jobs = [], []
    for src, mode in bn.SOURCES.items():
        if src in files and name != "base":
            flags = list(bn.FLAGS) + mode + extra
            if any(f.startswith("--offload-arch=") for f in extra):
                flags = [f for f in flags if f != "--offload-arch=gfx950"]
            obj = os.path.join(d, src.replace(".hip", ".o"))
            jobs.append((src, flags, obj))
        else:
            obj = os.path.join(BASE, src.replace(".hip", ".o"))
        objs.append(obj)
    with ThreadPoolExecutor(max_workers=4) as pool:
        res = list(pool.map(lambda j: compile_one(*j), jobs))
    bad = [(j[0], r[1]) for j, r in zip(jobs, res) if r[0]]
    if bad:
        print("variant", name, "FAILED to compile:", bad[0][0], bad[0][1][-300:])
        shutil.rmtree(d, ignore_errors=True)
        return False
    arch = [f for f in extra if f.startswith("--offload-arch=")] or ["--offload-arch=gfx950"]
    # (a library whose objects carry different target ids cannot be linked as one: xnack variants rebuild every file)
    out = os.path.join(LIBDIR, "var_%s.so" % name)
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + arch + ["-fno-gpu-rdc", "-shared", "-fPIC"] + objs + ["-o", out], capture_output=True, text=True)
    shutil.rmtree(d, ignore_errors=True)
    if r.returncode:
        print("variant", name, "FAILED to link:", r.stderr[-300:])
        return False
    print("built", name, " ".join(extra))
    return True


if __name__ == "__main__":
    if sys.argv[1] == "build":
        names = sys.argv[2:] or list(VARIANTS)
        os.makedirs(BASE, exist_ok=True)
        with ThreadPoolExecutor(max_workers=8) as pool:
            res = list(pool.map(lambda it: compile_one(it[0], list(bn.FLAGS) + it[1], os.path.join(BASE, it[0].replace(".hip", ".o"))), bn.SOURCES.items()))
        assert all(r[0] == 0 for r in res), res
        for old in glob.glob(os.path.join(LIBDIR, "var_*.so")):
            os.remove(old)
        for n in names:
            extra = VARIANTS[n]
            build_variant(n, extra, files=list(bn.SOURCES) if n.startswith("xnack") else HOT)
        shutil.rmtree(BASE, ignore_errors=True)
