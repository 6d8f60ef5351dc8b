#!/bin/bash
# ab_bench.sh "VARIANT ..." "CONFIG ..." [bench args]: bench.py value / value_one_stream of alternative builds (lib/libmm_var<V>.so,
# profiles/tools/build_variant.sh; "render" = the current build) on several configs.  Run on the GPU box (scratch copy of the repo).
cd /root/repo/3d-magic-mirror_amd/lib
cp libmm_render.so libmm_varrender.so
vars=$1; cfgs=$2; shift 2
for c in $cfgs; do for v in $vars; do
  cp libmm_var$v.so libmm_render.so
  echo -n "$c $v: "; timeout 300 python /root/repo/bench.py --config $c --cpu-seconds 0 --profile-steps 0 "$@" 2>&1 | tail -1 | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['value'], d.get('value_one_stream'))
except Exception as e: print('failed', e)"
done; done
cp libmm_varrender.so libmm_render.so
