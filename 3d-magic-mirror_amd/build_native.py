"""Builds libmm_render.so (the gfx950 HIP kernels + C ABI) in-tree with hipcc.  No GPU needed (offline cross-compile).

Flags that matter:
  --offload-arch=gfx950   CDNA4 only; no other targets, no compatibility layers
  -ffp-contract=off       every fp32 expression rounds as written (face_idx bit-parity with the CPU oracle)
  -munsafe-fp-atomics     atomicAdd(float) lowers to the hardware global_atomic_add_f32 (no CAS loop)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libmm_render.so")
SOURCES = ["mm_abi.hip", "mm_vertex.hip", "mm_raster.hip", "mm_backward.hip", "mm_loss.hip", "mm_nn.hip"]
HEADERS = ["mm_device.h", os.path.join("..", "..", "include", "mm_render.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-munsafe-fp-atomics",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + os.environ.get("MM_EXTRA_FLAGS", "").split() + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
