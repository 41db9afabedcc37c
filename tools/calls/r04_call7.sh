#!/bin/bash
# round 4, call 7: Radau state machine thresholds after the trims, then the whole GPU suite (recording exemption sets
# to a scratch file for comparison with the committed ones)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c7
for v in "" rs5 rs7; do
  if [ -n "$v" ]; then export HLM_HIP_LIB=$PWD/tiger_hlm_gpu_amd/lib/variants/libhlm_$v.so; else unset HLM_HIP_LIB; fi
  timeout -k 10 200 python tools/radau_machine_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-330 || exit 1
done
unset HLM_HIP_LIB
export HLM_PARITY_RECORD=$PWD/gpurun_out/c7/exempt_sets_all.json
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/c7/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/c7/pytest_all.log | cut -c1-300
