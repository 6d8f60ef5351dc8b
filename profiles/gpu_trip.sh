#!/bin/bash
# One GPU-box session: parity tests, VALU calibration, bench lines, rocprofv3 evidence per config.  From the repo root:
#   gpurun --timeout 3000 -- 'bash profiles/gpu_trip.sh r02 "config2 config3 config5"'
# Everything lands under gpurun_out/trip_<tag>/ (merged back); copy what is to be judged into profiles/.
TAG=${1:-r02}
CONFIGS=${2:-config2}
SKIP_TESTS=${3:-}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
OUT=gpurun_out/trip_$TAG
mkdir -p $OUT
export MM_PROFILE_OUT=$REPO/$OUT/profiles
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
if [ -z "$SKIP_TESTS" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
  tail -15 $OUT/pytest_gpu.log
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_calib profiles/tools/valu_calib.hip && timeout 120 /tmp/valu_calib > $OUT/valu_calibration.json; cat $OUT/valu_calibration.json
for C in $CONFIGS; do
  timeout 1200 bash profiles/run_profile.sh $C > $OUT/profile_$C.log 2>&1
  python profiles/summarize_profile.py $TAG $C > $OUT/summary_$C.md 2>&1; head -40 $OUT/summary_$C.md
  # raw CSVs are big (gpurun merges at most 64 MiB back): the summaries above are what is kept
  rm -rf gpurun_out/prof_$C
done
# the bench lines AFTER the counter passes, with this trip's counters installed: their roofline.traffic is then the fresh one
cp $OUT/profiles/traffic_latest.json $OUT/profiles/valu_latest.json profiles/ 2>/dev/null
for C in $CONFIGS; do
  EXTRA=""; [ "$C" != "config2" ] && EXTRA="--trainer-steps 0 --cpu-seconds 0 --options-steps 0 --steps 200"
  timeout 900 python bench.py --config $C $EXTRA > $OUT/bench_$C.json 2> $OUT/bench_$C.err; echo "bench $C rc=$?"; cut -c1-1500 $OUT/bench_$C.json
done
timeout 600 python bench.py --config config2x8 --trainer-steps 0 --cpu-seconds 0 --options-steps 0 --steps 200 > $OUT/bench_config2x8.json 2> $OUT/bench_config2x8.err; cut -c1-300 $OUT/bench_config2x8.json
du -sh gpurun_out
