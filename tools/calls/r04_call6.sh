#!/bin/bash
# round 4, call 6: lane occupancy of the Radau state machine's sections (counter builds)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
for v in rc3 rc5; do
  export HLM_HIP_LIB=$PWD/tiger_hlm_gpu_amd/lib/variants/libhlm_$v.so
  echo "## $v"
  timeout -k 10 300 python tools/radau_divergence_probe.py 2>&1 | grep -v amdgpu.ids || exit 1
done
