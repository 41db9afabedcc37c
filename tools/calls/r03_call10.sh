#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c10
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/c10/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/c10/pytest.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
