#!/bin/bash
# round 4, call 24: the round's rocprofv3 record: kernel trace + FETCH / WRITE / GRBM passes of the bench command,
# instruction-level passes (headline / wet / cold RK45 / Radau restart / Radau continue), the dense kernel's record
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c24
bash tools/profile_round.sh r04 > gpurun_out/c24/profile.log 2>&1; echo "profile rc=$?"
timeout -k 10 900 bash tools/pmc_issue.sh r04 > gpurun_out/c24/pmc_issue.log 2>&1; echo "pmc rc=$?"; grep "^done\|failed" gpurun_out/c24/pmc_issue.log | tr '\n' ' '
bash tools/profile_dense.sh r04_dense > gpurun_out/c24/profile_dense.log 2>&1; echo "dense profile rc=$?"
