#!/bin/bash
# round 4, call 10: the full default bench line, then the round's rocprofv3 record of the same command (kernel trace +
# separate FETCH / WRITE / GRBM passes) and the instruction-level passes (headline / wet / cold)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c10
timeout -k 10 900 python bench.py > gpurun_out/c10/bench_line.json 2> gpurun_out/c10/bench.err; echo "bench rc=$?"; tail -c 600 gpurun_out/c10/bench_line.json; echo
bash tools/profile_round.sh r04 > gpurun_out/c10/profile.log 2>&1; echo "profile rc=$?"
timeout -k 10 900 bash tools/pmc_issue.sh r04b > gpurun_out/c10/pmc_issue.log 2>&1; echo "pmc rc=$?"; tail -3 gpurun_out/c10/pmc_issue.log | cut -c1-200
