#!/bin/bash
# r06: what the large-batch kernel shapes buy FOUR B=48 steps in flight on four streams (profiles/r06_many_in_flight_ab.md).
#   build (no GPU):  python profiles/tools/variant_sweep.py build fl4=-DMM_FL4_MIN_B=1 vimg=-DMM_VIMG_BWD_MIN_B=1 largeshapes=-DMM_FL4_MIN_B=1,-DMM_WAVE_SHAPE_MIN_TILES=1,-DMM_VIMG_BWD_MIN_B=1
#   run (GPU box):   bash profiles/tools/many_in_flight_ab.sh [libs...]      (default: base only -- the product's own hinted leg)
mkdir -p gpurun_out/many_in_flight
A="--api-steps 0 --shim-steps 0 --trainer-steps 0 --cpu-seconds 0 --cpu-single-seconds 0 --options-steps 200 --profile-steps 0"
LIBS=${@:-base}
for rep in 1 2; do
for L in $LIBS; do
  python profiles/tools/bench_with_lib.py $L $A > gpurun_out/many_in_flight/bench_${L}_$rep.json 2>/dev/null
  python -c "
import json,sys; d=json.load(open('gpurun_out/many_in_flight/bench_${L}_$rep.json')); print('$L', $rep, {k:d.get(k) for k in ('value','value_four_streams','value_four_streams_hinted','value_four_streams_walk_wave')})"
done; done
