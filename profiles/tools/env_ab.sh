#!/bin/bash
# One-stream step under HIP runtime environment knobs (GPU box, repo root):  bash profiles/tools/env_ab.sh "VAR=val" "VAR2=val" ...
for pass in 1 2; do
  echo "== default (pass $pass)"; python profiles/tools/kernel_times.py config2 config3 2>&1 | grep -E "^config|^market"
  for kv in "$@"; do
    echo "== $kv (pass $pass)"; env $kv python profiles/tools/kernel_times.py config2 config3 2>&1 | grep -E "^config|^market"
  done
done
