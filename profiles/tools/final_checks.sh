#!/bin/bash
# Round-end checks on the final kernels (GPU box, repo root): randomised parity + soaks -> gpurun_out/final/
#   python profiles/tools/variant_sweep.py build largeshapes=-DMM_FL4_MIN_B=1,-DMM_WAVE_SHAPE_MIN_TILES=1,-DMM_VIMG_BWD_MIN_B=1     (before, no GPU needed)
O=gpurun_out/final; mkdir -p $O
{ echo "## python profiles/tools/fuzz_parity.py 700 ${1:-11011}   (product)"; timeout 1000 python profiles/tools/fuzz_parity.py 700 ${1:-11011} 2>&1 | grep -v "^ok\|amdgpu.ids"; } > $O/fuzz_product.txt
{ echo "## MM_OPTIONS=4096 python profiles/tools/fuzz_parity.py 300 ${2:-12012}   (MM_OPT_MANY_IN_FLIGHT on every case)"; MM_OPTIONS=4096 timeout 600 python profiles/tools/fuzz_parity.py 300 ${2:-12012} 2>&1 | grep -v "^ok\|amdgpu.ids"; } > $O/fuzz_hint.txt
{ echo "## MM_DBG_LIB=var_largeshapes.so python profiles/tools/fuzz_parity.py 300 ${3:-13013}   (the large-batch shapes forced on at every size)"; MM_DBG_LIB=$PWD/3d-magic-mirror_amd/lib/var_largeshapes.so timeout 600 python profiles/tools/fuzz_parity.py 300 ${3:-13013} 2>&1 | grep -v "^ok\|amdgpu.ids"; } > $O/fuzz_largeshapes.txt
{ echo "## python profiles/tools/fuzz_8f.py"; timeout 600 python profiles/tools/fuzz_8f.py 2>&1 | tail -3; } > $O/fuzz_8f.txt
{ echo "## python profiles/tools/determinism_soak.py"; timeout 600 python profiles/tools/determinism_soak.py 2>&1 | tail -3; echo "## python profiles/tools/walk_soak.py 300"; timeout 900 python profiles/tools/walk_soak.py 300 2>&1 | grep -v amdgpu.ids | tail -8; } > $O/soaks.txt
tail -2 $O/fuzz_product.txt $O/fuzz_hint.txt $O/fuzz_largeshapes.txt $O/fuzz_8f.txt; cat $O/soaks.txt
