#!/bin/bash
# round 4, call 12: Radau pass with lane refill (persistent grid) -- parity tests, timing by set-up threshold, counters
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c12
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "radau" > gpurun_out/c12/pytest_radau.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/c12/pytest_radau.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
for v in "" rs3 rs4 rs5; do
  if [ -n "$v" ]; then export HLM_HIP_LIB=$PWD/tiger_hlm_gpu_amd/lib/variants/libhlm_$v.so; else unset HLM_HIP_LIB; fi
  timeout -k 10 200 python tools/radau_machine_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-330 || exit 1
done
HLM_HIP_LIB=$PWD/tiger_hlm_gpu_amd/lib/variants/libhlm_rc6.so timeout -k 10 300 python tools/radau_divergence_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c12/radau_sections.txt
