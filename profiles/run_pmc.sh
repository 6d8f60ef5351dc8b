#!/bin/bash
# Collects per-kernel PMC counters for the bench workload in separate rocprofv3 passes (kernel-trace only, no
# sys/hip/hsa trace domains) and writes CSVs under gpurun_out/pmc/.  Run on the GPU box from the repo root.
set -u
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc
mkdir -p $OUT
CMD="python /root/repo/bench.py --mode eager --streams 1 --cpu-seconds 0 --profile-steps 0 --steps 10 --warmup 3 ${BENCH_ARGS:-}"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pass$i -o p -- $CMD > $OUT/pass$i.log 2>&1
done
find $OUT -name "*.csv" | head -30
