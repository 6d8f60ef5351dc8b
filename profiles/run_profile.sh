#!/bin/bash
# rocprofv3 evidence for one bench config, on the GPU box from the repo root:   bash profiles/run_profile.sh config2 [quick]
#   1. --kernel-trace --stats of the single-stream eager bench               -> gpurun_out/prof_<config>/stats/
#   2. PMC counters in SEPARATE passes (kernel-trace only, no other domains)   -> gpurun_out/prof_<config>/pmc/pass*/
#   3. profiles/summarize_profile.py <tag> <config> turns both into the tracked summaries (run it on the box too: it stamps
#      the kernel-source digest the counters were measured on).
# "quick": the two SQ passes only (instruction mix / stalls), no TCC passes.
set -u
CONFIG=${1:-config2}
QUICK=${2:-}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_$CONFIG
rm -rf $OUT; mkdir -p $OUT/stats $OUT/pmc
BENCH="python $REPO/bench.py --config $CONFIG --mode eager --streams 1 --cpu-seconds 0 --profile-steps 0 --api-steps 0 --shim-steps 0 --trainer-steps 0 --options-steps 0 --settle-seconds 0.2 --reps 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $BENCH --steps 50 --warmup 5 > $OUT/stats/bench.log 2>&1
i=0
SETS=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"
      "SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE")
if [ -z "$QUICK" ]; then SETS+=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum"); fi
for set in "${SETS[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc/pass$i -o p -- $BENCH --steps 6 --warmup 2 > $OUT/pmc/pass$i.log 2>&1
done
ls $OUT/stats | head -5; find $OUT/pmc -name "*counter_collection.csv" | wc -l
