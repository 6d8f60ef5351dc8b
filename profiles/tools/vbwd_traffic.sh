#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (KiB per launch) and duration of vertex_bwd for the product library and the break-down variants lib/var_skip*.so
# (profiles/tools/variant_sweep.py build skip1=-DMM_DBG_VBWD_SKIP=1 ...):   bash profiles/tools/vbwd_traffic.sh [config]
CFG=${1:-config2}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in base skip1 skip2 skip4 skip7; do
  LIBV=""; [ "$v" != base ] && LIBV=$REPO/3d-magic-mirror_amd/lib/var_$v.so
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/vb_$v_$c
    MM_DBG_LIB=$LIBV rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/vb_${v}_$c -o p -- python $REPO/profiles/tools/kernel_times.py $CFG > /tmp/vb_${v}_$c.log 2>&1
    python $REPO/profiles/tools/pmc_kernel.py /tmp/vb_${v}_$c vertex_bwd $v
  done
  MM_DBG_LIB=$LIBV rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vb_${v}_t -o s -- python $REPO/profiles/tools/kernel_times.py $CFG > /tmp/vb_${v}_t.log 2>&1
  python $REPO/profiles/tools/kstat.py /tmp/vb_${v}_t vertex_bwd $v
done
