#!/bin/bash
# kernel resource usage of one csrc translation unit as hipcc sees it:  profiles/tools/kres.sh mm_raster.hip [extra flags]   (EXACT flags unless MM_RELAXED=1)
# name | SGPRs | VGPRs | spills (S/V) | scratch B | occupancy (waves/SIMD) | LDS B | code bytes
cd "$(dirname "$0")/../../3d-magic-mirror_amd/csrc" || exit 1
src=$1; shift
mode="-ffp-contract=off"; [ -n "$MM_RELAXED" ] && mode="-ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize"
obj=$(mktemp /tmp/kres.XXXXXX.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-gpu-rdc $mode "$@" -Rpass-analysis=kernel-resource-usage -c "$src" -o "$obj" 2>&1 | \
  awk '/Function Name:/ {name=$NF} /remark:.*Name:/ {name=$(NF-1)} /TotalSGPRs:/ {s=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /SGPRs Spill/ {ss=$(NF-1)} /VGPRs Spill/ {vs=$(NF-1)} /Occupancy/ {oc=$(NF-1)} /LDS Size/ {print name, "sgpr", s, "vgpr", v, "spill", ss "/" vs, "scratch", sc, "occ", oc, "lds", $(NF-1)}' | c++filt | sed 's/mm:://g'
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$obj" --output="$obj.co" --unbundle 2>/dev/null && \
  /opt/rocm/lib/llvm/bin/llvm-readelf -s --wide "$obj.co" 2>/dev/null | awk '$4=="FUNC" {print $3, $8}' | c++filt | sed 's/mm:://g' | sort -k2 | awk '{print "   code bytes", $1, $2, $3, $4}'
rm -f "$obj" "$obj.co"
