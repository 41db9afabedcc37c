#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c18
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "compact_windows or time_pieces or launch_shapes or no_dense_launches" > gpurun_out/c18/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/c18/pytest.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/pieces_probe.py --pieces 0 32 --runs 0 1 2>&1 | grep -v amdgpu.ids | cut -c1-190 &&
timeout -k 10 300 python tools/pieces_probe.py --tmin -5 --pieces 32 64 --runs 1 --groups 0 4 8 16 2>&1 | grep -v amdgpu.ids | cut -c1-190 &&
timeout -k 10 300 python tools/pieces_probe.py --tmin -5 --pieces 64 --runs 1 --groups 8 --windows 2 4 2>&1 | grep -v amdgpu.ids | cut -c1-190
