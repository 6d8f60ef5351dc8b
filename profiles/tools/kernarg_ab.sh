#!/bin/bash
# Where HIP keeps kernel arguments (HIP_FORCE_DEV_KERNARG: host-coherent memory or device memory) against this path's kernels: every wave's chain
# starts with the scalar loads of its arguments.  GPU box, repo root:  bash profiles/tools/kernarg_ab.sh > gpurun_out/kernarg_ab.txt
for pass in 1 2; do
for v in unset 0 1; do
  if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  echo "== HIP_FORCE_DEV_KERNARG=$v (pass $pass)"
  ./profiles/tools/launch_floor.bin | grep -E "\(a\)|\(c\)|\(d\)|six"
  python profiles/tools/kernel_times.py config2 market config3 2>&1 | grep -E "^config|^market"
done
done
