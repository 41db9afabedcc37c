#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c13
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "radau" > gpurun_out/c13/pytest_radau.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/c13/pytest_radau.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
for v in "" rs5 rs7; do
  if [ -n "$v" ]; then export HLM_HIP_LIB=$PWD/tiger_hlm_gpu_amd/lib/variants/libhlm_$v.so; else unset HLM_HIP_LIB; fi
  timeout -k 10 200 python tools/radau_machine_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-330 || exit 1
done
