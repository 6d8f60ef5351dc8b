#!/bin/bash
# rocprofv3 per-kernel averages of the SURVEY 8(f) kernels -> gpurun_out/f8/ + a table (profiles/tools/f8_summary.py)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $REPO/gpurun_out/f8; mkdir -p $REPO/gpurun_out/f8
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/f8 -o f8 -- python $REPO/profiles/tools/f8_workload.py > $REPO/gpurun_out/f8/run.log 2>&1
cd $REPO && python profiles/tools/f8_summary.py
