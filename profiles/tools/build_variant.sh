#!/bin/bash
# build_variant.sh NAME SRC.hip [extra flags]: links lib/libmm_var<NAME>.so from the current objects with SRC recompiled under the
# extra flags (e.g. -DMM_COOP_TILES=0).  Load it with MM_DBG_LIB=... in profiles/tools/kernel_times.py / valu_variant.sh.
set -e
cd /root/repo/3d-magic-mirror_amd
name=$1; src=$2; shift 2
python build_native.py > /dev/null
mode="-ffp-contract=off"
[ "$src" = mm_backward.hip ] && mode="-ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize"
flags=$(python - <<PY
import importlib.util
s = importlib.util.spec_from_file_location("b", "build_native.py"); m = importlib.util.module_from_spec(s); s.loader.exec_module(m)
print(" ".join(m.FLAGS))
PY
)
/opt/rocm/bin/hipcc $flags $mode "$@" -c csrc/$src -o /tmp/var_$name.o
objs=$(ls lib/obj/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fno-gpu-rdc -shared -fPIC $objs /tmp/var_$name.o -o lib/libmm_var$name.so
echo lib/libmm_var$name.so
