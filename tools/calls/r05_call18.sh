#!/bin/bash
# round 5, call 18: Radau pass with Z recomputed from W (15 doubles less state): same box A/B
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
for v in rz0 rz1 rz1w1 rz0 rz1; do
  HLM_HIP_LIB=$PWD/tiger_hlm_gpu_amd/lib/variants/libhlm_$v.so timeout -k 10 200 python tools/radau_machine_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-420 || exit 1
done | tee $O/radau_z_probe.txt
