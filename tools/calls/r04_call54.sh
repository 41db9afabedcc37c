#!/bin/bash
# round 4, call 54: A/B of the two-instruction clamp in the dense ring kernel (1 M hillslopes x 1 day of 1-minute rows):
# the library before (variants/libhlm_base.so) and after, alternately on one box
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c54
for rep in 1 2; do
  for LIB in variants/libhlm_base.so libhlm_hip.so; do
    echo "## $LIB" >> gpurun_out/c54/ab_dense.txt
    HLM_HIP_LIB=$PWD/tiger_hlm_gpu_amd/lib/$LIB timeout -k 10 300 python tools/dense_subwindow_probe.py --ns 1000000 --sub 1 2>&1 | grep -v amdgpu >> gpurun_out/c54/ab_dense.txt || exit 1
  done
done
cat gpurun_out/c54/ab_dense.txt
