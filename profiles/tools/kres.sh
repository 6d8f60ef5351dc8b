#!/bin/bash
# kernel resource usage of one csrc translation unit as hipcc sees it, with the PRODUCT's flags for that file (build_native.SOURCES / FLAGS):
#   profiles/tools/kres.sh mm_raster.hip [extra flags]
# name | SGPRs | VGPRs | spills (SGPR/VGPR) | scratch B | occupancy (waves/SIMD) | LDS B ; then code bytes per kernel
cd "$(dirname "$0")/../../3d-magic-mirror_amd/csrc" || exit 1
src=$1; shift
flags=$(cd ../.. && python - "$src" <<'PY'
import importlib, sys
bn = importlib.import_module("3d-magic-mirror_amd.build_native")
print(" ".join(bn.FLAGS + bn.SOURCES[sys.argv[1]]))
PY
)
obj=$(mktemp /tmp/kres.XXXXXX.o)
/opt/rocm/bin/hipcc $flags "$@" -Rpass-analysis=kernel-resource-usage -c "$src" -o "$obj" 2>&1 | \
  awk '/Function Name:/ {name=$NF} /remark:.*Name:/ {name=$(NF-1)} /TotalSGPRs:/ {s=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /SGPRs Spill/ {ss=$(NF-1)} /VGPRs Spill/ {vs=$(NF-1)} /Occupancy/ {oc=$(NF-1)} /LDS Size/ {print name, "sgpr", s, "vgpr", v, "spill(sgpr/vgpr)", ss "/" vs, "scratch", sc, "occ", oc, "lds", $(NF-1)}' | c++filt | sed 's/mm:://g'
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$obj" --output="$obj.co" --unbundle 2>/dev/null && \
  /opt/rocm/lib/llvm/bin/llvm-readelf -s --wide "$obj.co" 2>/dev/null | awk '$4=="FUNC" {print $3, $8}' | c++filt | sed 's/mm:://g' | sort -k2 | awk '{print "   code bytes", $1, $2, $3, $4}'
rm -f "$obj" "$obj.co"
