#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c34
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/c34/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 gpurun_out/c34/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c34/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/c34/smoke.log
