#!/bin/bash
# round 4, call 22: h.k products in the closed-form paths: timing (headline, wet, cold) and the headline parity tests
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c22
timeout -k 10 200 python tools/pieces_probe.py --pieces 0 32 --runs 0 2>&1 | grep -v amdgpu.ids | cut -c1-190 &&
timeout -k 10 200 python tools/pieces_probe.py --wet --pieces 32 --runs 0 2>&1 | grep -v amdgpu.ids | cut -c1-190 &&
timeout -k 10 200 python tools/pieces_probe.py --tmin -5 --pieces 32 --runs 1 --groups 0 8 2>&1 | grep -v amdgpu.ids | cut -c1-190 &&
timeout -k 10 900 python -m pytest tests/test_headline_kernel_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x > gpurun_out/c22/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/c22/pytest.log | cut -c1-300
