#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c27
timeout -k 5 200 python tools/fixed_step_pieces_probe.py > gpurun_out/c27/fixed.txt 2>&1; echo "fixed probe rc=$?"; grep -vc "lanes differing 0," gpurun_out/c27/fixed.txt
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/c27/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 gpurun_out/c27/pytest.log
