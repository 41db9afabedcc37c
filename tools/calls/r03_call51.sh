#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c51
timeout -k 10 200 python tools/pieces_probe.py --pieces 0 32 32 > gpurun_out/c51/pieces.txt 2>&1; grep -v amdgpu.ids gpurun_out/c51/pieces.txt
timeout -k 10 200 python tools/wet_probe.py >> gpurun_out/c51/pieces.txt 2>&1; grep -v amdgpu.ids gpurun_out/c51/pieces.txt | tail -2
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/c51/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -n "^FAILED\|passed\|failed" gpurun_out/c51/pytest.log | cut -c1-220 | tail -30
