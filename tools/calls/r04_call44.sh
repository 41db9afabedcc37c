#!/bin/bash
# round 4, call 44: the instruction-level PMC passes of the round's final library (capped run schedule): headline / wet /
# cold RK45 / Radau restart / Radau continue
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c44
timeout -k 10 1100 bash tools/pmc_issue.sh r04 > gpurun_out/c44/pmc_issue.log 2>&1; echo "pmc rc=$?"; grep "^done\|failed" gpurun_out/c44/pmc_issue.log | tr '\n' ' '
