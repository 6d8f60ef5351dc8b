"""Builds libmm_render.so (the gfx950 HIP kernels + C ABI) in-tree with hipcc.  No GPU needed (offline cross-compile).

Flags that matter:
  --offload-arch=gfx950   CDNA4 only; no other targets, no compatibility layers
  -ffp-contract=off       every fp32 expression rounds as written (face_idx bit-parity with the CPU oracle); HIP's default
                          correctly rounded fp32 '/' and sqrtf are kept for the same reason
  -munsafe-fp-atomics     atomicAdd(float) lowers to the hardware global_atomic_add_f32 / ds_add_f32 (no CAS loop) -- ONLY for the three translation units
                          that still hold a float atomic (FP_ATOMICS below: the un-fused operators' fallback forms, README "float atomics"); the render
                          path, recon_data and every other 8(f) kernel are compiled without it and contain none
Per file: the gathers of the backward (mm_backward.hip) are held to 1e-4, not to the bit, and bound by instruction issue, so they are
compiled with fma contraction and the 2.5-ulp division/sqrt sequences (about a third fewer vector instructions).  The pixel pass of the
backward (mm_pixel_bwd.hip) recomputes the forward's per-pixel quantities and is compiled like the forward (see csrc/mm_backward.h).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libmm_render.so")
# -fno-slp-vectorize on the relaxed files: the SLP vectoriser's packed-fp32 code costs more v_mov shuffling than it saves here
# (measured: same instruction count, +2 % images/s without it; the exact files lose 7 % instructions without it and keep it).
EXACT = ["-ffp-contract=off"]
RELAXED = ["-ffp-contract=fast", "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize"]
FP_ATOMICS = ["-munsafe-fp-atomics"]     # mm_dibr.hip (features of more than 8 channels: LDS adds within one wave), mm_ops.hip (texture_mapping backward WITHOUT a
                                        # workspace), mm_texflow.hip (gradient to the sampled image): tests/test_gpu_float_atomics.py pins their run-to-run spread
SOURCES = {"mm_abi.hip": EXACT, "mm_reg.hip": EXACT, "mm_texflow.hip": EXACT + FP_ATOMICS, "mm_attloss.hip": EXACT, "mm_vertex.hip": EXACT, "mm_raster.hip": EXACT,
           "mm_backward.hip": RELAXED, "mm_pixel_bwd.hip": EXACT, "mm_loss.hip": EXACT, "mm_nn.hip": EXACT, "mm_dibr.hip": EXACT + FP_ATOMICS, "mm_ops.hip": EXACT + FP_ATOMICS}
HEADERS = ["mm_device.h", "mm_raster_common.h", "mm_raster_walk.h", "mm_backward.h", "mm_order.h", os.path.join("..", "..", "include", "mm_render.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


# A TEST build of the same sources (tests/test_gpu_self_offsets.py): the backward's plan workgroups publish the texture-record list offsets
# ~0.3 ms late and the pixel lanes give up waiting after four polls, so every lane takes the path that forms its offset from the forward's
# counts -- the path that makes the backward's progress independent of the order workgroups are dispatched in.  Built by __graft_entry__.build(); never loaded by the product.
TEST_LIB_SELF_OFFSETS = os.path.join(HERE, "lib", "libmm_render_test_self_offsets.so")
TEST_FLAGS_SELF_OFFSETS = ["-DMM_DBG_LATE_TOFF", "-DMM_TOFF_SPIN_MAX=4"]


def build_test_variants(force=False):
    if force or not os.path.exists(TEST_LIB_SELF_OFFSETS) or os.path.getmtime(TEST_LIB_SELF_OFFSETS) < os.path.getmtime(build()):
        build(out=TEST_LIB_SELF_OFFSETS, extra_flags=TEST_FLAGS_SELF_OFFSETS)
    return TEST_LIB_SELF_OFFSETS


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in list(SOURCES) + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, out=None, extra_flags=None):
    """Compile + link under an exclusive file lock, into a private directory, and publish the library with one atomic rename:
    N ranks starting together on a fresh checkout (torchrun bench.py) build once, and nobody can dlopen a half-written file."""
    import fcntl
    import shutil
    import tempfile
    target = out or LIB                                          # out: a variant build (profiles/tools), e.g. lib/libmm_timeline.so
    if out is None and not force and not needs_build():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    with open(os.path.join(os.path.dirname(LIB), ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if out is None and not force and not needs_build():  # another process built it while we waited for the lock
                return LIB
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            extra = os.environ.get("MM_EXTRA_FLAGS", "").split() + list(extra_flags or [])
            tmp = tempfile.mkdtemp(prefix="obj.", dir=os.path.dirname(LIB))
            try:
                def compile_one(item):
                    src, mode = item
                    obj = os.path.join(tmp, src.replace(".hip", ".o"))
                    cmd = [hipcc] + FLAGS + mode + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
                    if verbose:
                        print(" ".join(cmd))
                    subprocess.check_call(cmd)
                    return obj

                with ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:
                    objs = list(pool.map(compile_one, SOURCES.items()))
                built = os.path.join(tmp, "libmm_render.so")
                cmd = [hipcc, "--offload-arch=gfx950", "-fno-gpu-rdc", "-shared", "-fPIC"] + objs + ["-o", built]
                if verbose:
                    print(" ".join(cmd))
                subprocess.check_call(cmd)
                os.replace(built, target)                        # atomic on one filesystem
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return target


EXT = os.path.join(HERE, "lib", "mm_torch_ext.so")
EXT_SRC = os.path.join(CSRC, "mm_torch_ext.cpp")


def ext_needs_build():
    if not os.path.exists(EXT):
        return True
    t = os.path.getmtime(EXT)
    return any(os.path.getmtime(d) > t for d in (EXT_SRC, os.path.join(HERE, "..", "include", "mm_render.h")))


def build_torch_ext(force=False, verbose=False):
    """The optional host-side fast path of the autograd API (csrc/mm_torch_ext.cpp): host compiler only, ~30 s, same lock / atomic publish
    as the library.  Plumbing above the C ABI -- diff_render.py works without it (Python path, same calls)."""
    import fcntl
    import sysconfig
    import tempfile
    import torch
    from torch.utils.cpp_extension import include_paths
    if not force and not ext_needs_build():
        return EXT
    os.makedirs(os.path.dirname(EXT), exist_ok=True)
    with open(os.path.join(os.path.dirname(EXT), ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not ext_needs_build():
                return EXT
            tl = os.path.join(os.path.dirname(torch.__file__), "lib")
            fd, tmp = tempfile.mkstemp(prefix="ext.", suffix=".so", dir=os.path.dirname(EXT))
            os.close(fd)
            cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-shared", "-fPIC", "-DTORCH_EXTENSION_NAME=mm_torch_ext",
                   "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
                   "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",    # torch's own HIP stream / guard headers (host code only, no device code)
                   EXT_SRC, "-o", tmp] + \
                  ["-I" + i for i in include_paths() + [sysconfig.get_paths()["include"], os.environ.get("ROCM_PATH", "/opt/rocm") + "/include"]] + \
                  ["-L" + tl, "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip", "-ltorch_python", "-Wl,-rpath," + tl]
            if verbose:
                print(" ".join(cmd))
            try:
                subprocess.check_call(cmd)
                os.replace(tmp, EXT)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return EXT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
