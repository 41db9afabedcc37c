#!/bin/bash
# round 4, call 8: streamed forcings with growing chunks / queue-shaped window launches: bitwise tests + timing
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c8
timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "streamed_forcings or time_pieces or checkpoint" > gpurun_out/c8/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/c8/pytest.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/streamed_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-300
