#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c40
timeout -k 10 500 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "streamed_forcings or time_pieces" > gpurun_out/c40/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 gpurun_out/c40/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/streamed_probe.py > gpurun_out/c40/probe.txt 2>&1; echo "probe rc=$?"; grep -v amdgpu.ids gpurun_out/c40/probe.txt
