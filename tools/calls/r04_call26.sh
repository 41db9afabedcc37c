#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python tools/dense_subwindow_probe.py --ns 200000 --sub 1 2 5 2>&1 | grep -v amdgpu.ids | cut -c1-200 &&
timeout -k 10 600 python tools/dense_subwindow_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
