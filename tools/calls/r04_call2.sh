#!/bin/bash
# round 4, call 2: runs of pieces in the queue kernel -- timing against round 3's protocol (run_pieces 1), bit identity,
# the launch-shape tests, HBM traffic of the new default
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c2
timeout -k 10 300 python tools/pieces_probe.py --pieces 0 32 64 --runs 1 0 2 4 8 2>&1 | grep -v amdgpu.ids | cut -c1-160 &&
timeout -k 10 200 python tools/pieces_probe.py --tmin -5 --pieces 0 32 64 128 --runs 1 0 2>&1 | grep -v amdgpu.ids | cut -c1-160 &&
timeout -k 10 200 python tools/pieces_probe.py --ns 100000 --pieces 0 32 64 --runs 1 0 2>&1 | grep -v amdgpu.ids | cut -c1-160 &&
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "time_pieces or launch_shapes or no_dense or c4_four or smoke" > gpurun_out/c2/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/c2/pytest.log | cut -c1-300
bash tools/profile_round.sh r04a > gpurun_out/c2/profile.log 2>&1; echo "profile rc=$?"; tail -3 gpurun_out/c2/profile.log
