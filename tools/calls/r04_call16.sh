#!/bin/bash
# round 4, call 16: groups of tiles packed per piece (queue_group_tiles): bit identity, timing on cold cells; the
# headline must not have moved
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c16
timeout -k 10 900 python -m pytest tests/test_headline_kernel_gpu.py tests/test_host_logic.py -m gpu -q -x -k "compact_windows" > gpurun_out/c16/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/c16/pytest.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/pieces_probe.py --pieces 0 32 --runs 0 1 2>&1 | grep -v amdgpu.ids | cut -c1-190 &&
timeout -k 10 500 python tools/pieces_probe.py --tmin -5 --pieces 32 64 128 256 --runs 1 --groups 0 4 8 16 2>&1 | grep -v amdgpu.ids | cut -c1-190
