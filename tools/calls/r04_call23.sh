#!/bin/bash
# round 4, call 23: whole GPU suite on the current head, smoke, then the round's bench line and profiles
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c23
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/c23/pytest_all.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/c23/pytest_all.log | cut -c1-300
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -4 &&
timeout -k 10 900 python bench.py > gpurun_out/c23/bench_line.json 2> gpurun_out/c23/bench.err; echo "bench rc=$?"; tail -c 300 gpurun_out/c23/bench_line.json; echo
