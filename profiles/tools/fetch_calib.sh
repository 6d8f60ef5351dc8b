#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on this path's access patterns (GPU box, repo root): bash profiles/tools/fetch_calib.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/fetch_calib
rm -rf $OUT; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib $REPO/profiles/tools/fetch_calib.hip || exit 1
cd /tmp && export TMPDIR=/tmp
timeout 300 /tmp/fetch_calib > $OUT/known.json
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -o f -- /tmp/fetch_calib > $OUT/f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -o w -- /tmp/fetch_calib > $OUT/w.log 2>&1
cd $REPO && python profiles/tools/fetch_calib.py $OUT
