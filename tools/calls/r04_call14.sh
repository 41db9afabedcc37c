#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
for v in w1 w1s5; do
  export HLM_HIP_LIB=$PWD/tiger_hlm_gpu_amd/lib/variants/libhlm_$v.so
  timeout -k 10 200 python tools/radau_machine_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-330 || exit 1
done
