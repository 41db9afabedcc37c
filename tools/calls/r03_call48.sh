#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c48
rm -rf gpurun_out/prof_r03
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/c48/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/c48/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c48/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/c48/smoke.log
timeout -k 10 500 tools/profile_round.sh r03 > gpurun_out/c48/profile.log 2>&1; echo "profile rc=$?"
