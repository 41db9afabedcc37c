#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python tools/dense_subwindow_probe.py --sub 1 2 4 8 2>&1 | grep -v amdgpu.ids | cut -c1-200 &&
timeout -k 10 600 python -m pytest tests/test_output_selection_gpu.py -m gpu -q -x -k "sub_windows" 2>&1 | tail -2
