"""Timing experiment (libmm_pp.so variant): gather_bwd with one kind of its workgroups left out (MM_DBG_GATHER bit 0: texture tiles,
bit 1: face sweeps, bit 2: chunk waves).  Results are wrong by construction; only the kernel's duration is read."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cfgs = sys.argv[1:] or ["config2"]
import importlib
sys.path.insert(0, ROOT)
bn = importlib.import_module("3d-magic-mirror_amd.build_native")
var = os.path.join(os.path.dirname(bn.LIB), "libmm_pp.so")
bn.build(out=var, extra_flags=["-DMM_PHASE_PROF"])
for skip, label in ((0, "all"), (1, "no texture tiles"), (2, "no face sweeps"), (3, "chunk waves only"), (6, "texture tiles only"), (5, "face sweeps only")):
    env = dict(os.environ, MM_DBG_GATHER=str(skip), MM_DBG_LIB=var)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "tools", "kernel_times.py")] + cfgs, env=env, capture_output=True, text=True).stdout
    for line in out.splitlines():
        if line.startswith("config"):
            print("%-20s %s" % (label, line[:150]))
