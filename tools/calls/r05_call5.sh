#!/bin/bash
# round 5, call 5: per-hillslope numbers of the intermittent-50k parity comparison (HLM_PARITY_DIAG)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O/diag
export HLM_PARITY_DIAG=$PWD/$O/diag
timeout -k 10 1100 python -m pytest tests/test_headline_kernel_gpu.py -m gpu -q -s -k "saturating and intermittent and queue" > $O/pytest_diag.log 2>&1; echo "pytest rc=$?"; ls -la $O/diag
