#!/bin/bash
# round 4, call 15: re-tiling between windows (compact_windows): bit identity, then timing on cold cells and headline
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c15
timeout -k 10 900 python -m pytest tests/test_headline_kernel_gpu.py -m gpu -q -x -k "compact_windows" > gpurun_out/c15/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/c15/pytest.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python tools/pieces_probe.py --tmin -5 --pieces 0 32 64 --runs 1 --windows 0 2 3 4 6 8 2>&1 | grep -v amdgpu.ids | cut -c1-170 &&
timeout -k 10 200 python tools/pieces_probe.py --pieces 32 --runs 0 --windows 0 3 2>&1 | grep -v amdgpu.ids | cut -c1-170
