#!/bin/bash
# round 4, call 58: HEAD with the two-instruction clamp and the wave-uniform step clip: the whole GPU suite, smoke(), the default bench line, the
# round's rocprofv3 record (kernel trace + FETCH / WRITE / GRBM passes) and the instruction-level PMC passes
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c58
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/c58/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/c58/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/c58/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc"; tail -2 gpurun_out/c58/smoke.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python bench.py > gpurun_out/c58/bench_line.json 2> gpurun_out/c58/bench.err; echo "bench rc=$?"; tail -c 300 gpurun_out/c58/bench_line.json; echo
bash tools/profile_round.sh r04 > gpurun_out/c58/profile.log 2>&1; echo "profile rc=$?"
timeout -k 10 1000 bash tools/pmc_issue.sh r04 > gpurun_out/c58/pmc_issue.log 2>&1; echo "pmc rc=$?"; grep "^done\|failed" gpurun_out/c58/pmc_issue.log | tr '\n' ' '
