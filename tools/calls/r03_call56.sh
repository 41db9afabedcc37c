#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c56
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/c56/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -n "^FAILED\|passed\|failed" gpurun_out/c56/pytest.log | cut -c1-200 | tail -12
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c56/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/c56/smoke.log
