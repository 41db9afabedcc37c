#!/bin/bash
# round 4, call 62: the whole GPU suite, smoke() and the default bench line on HEAD
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c62
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/c62/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/c62/pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/c62/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc"; tail -2 gpurun_out/c62/smoke.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python bench.py > gpurun_out/c62/bench_line.json 2> gpurun_out/c62/bench.err; echo "bench rc=$?"; tail -c 300 gpurun_out/c62/bench_line.json; echo
