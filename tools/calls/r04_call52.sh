#!/bin/bash
# round 4, call 52: HEAD after the Radau pass's batched reciprocals: the whole GPU suite, smoke(), the default bench
# line, then the instruction-level PMC passes (headline / wet / cold RK45 / Radau restart / Radau continue)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c52
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/c52/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/c52/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/c52/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc"; tail -2 gpurun_out/c52/smoke.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python bench.py > gpurun_out/c52/bench_line.json 2> gpurun_out/c52/bench.err; echo "bench rc=$?"; tail -c 300 gpurun_out/c52/bench_line.json; echo
timeout -k 10 1000 bash tools/pmc_issue.sh r04 > gpurun_out/c52/pmc_issue.log 2>&1; echo "pmc rc=$?"; grep "^done\|failed" gpurun_out/c52/pmc_issue.log | tr '\n' ' '
