#!/bin/bash
# Two SQ counter passes only (instruction mix / stalls) for the bench workload; CSVs under gpurun_out/pmcq/.
set -u
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmcq
rm -rf $OUT; mkdir -p $OUT
CMD="python /root/repo/bench.py --mode eager --streams 1 --cpu-seconds 0 --profile-steps 0 --steps 10 --warmup 3 ${BENCH_ARGS:-}"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pass$i -o p -- $CMD > $OUT/pass$i.log 2>&1
done
python /root/repo/profiles/summarize_pmc.py $OUT | grep -A17 "raster_fwd"
