#!/bin/bash
# Shader clock under this path's load, and what pinning the performance level does (GPU box, repo root; changes the box's DPM setting for the rest of the call)
sample() { for i in $(seq 1 ${1:-12}); do rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.5; done; }
echo "== idle"; sample 2
echo "== default DPM, kernel_times config2 running"
python profiles/tools/kernel_times.py config2 config2 config2 config3 > /tmp/kt_default.txt 2>&1 &
P=$!; sleep 6; sample 10; wait $P; grep -E "^config" /tmp/kt_default.txt
echo "== rocm-smi --setperflevel high"; rocm-smi --setperflevel high 2>&1 | grep -v "^$" | head -5
python profiles/tools/kernel_times.py config2 config2 config2 config3 > /tmp/kt_high.txt 2>&1 &
P=$!; sleep 6; sample 10; wait $P; grep -E "^config" /tmp/kt_high.txt
echo "== rocm-smi --setperfdeterminism 2400 (if supported)"; rocm-smi --setperfdeterminism 2400 2>&1 | grep -v "^$" | head -5
python profiles/tools/kernel_times.py config2 config2 config3 2>&1 | grep -E "^config"
echo "== back to auto"; rocm-smi --resetperfdeterminism 2>&1 | head -3; rocm-smi --setperflevel auto 2>&1 | grep -v "^$" | head -3
python profiles/tools/kernel_times.py config2 2>&1 | grep -E "^config"
