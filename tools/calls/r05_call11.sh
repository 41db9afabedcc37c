#!/bin/bash
# round 5, call 11: the whole GPU suite after the removal of the quarantined paths, then the instruction-level PMC
# passes of all workloads (headline, wet, saturating, active surface, cold, cold + Radau)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest_all2.log 2>&1; echo "pytest all rc=$?"; tail -4 $O/pytest_all2.log | cut -c1-300
bash tools/pmc_issue.sh r05 > $O/pmc_issue.log 2>&1; echo "pmc rc=$?"; tail -5 $O/pmc_issue.log | cut -c1-200
cp profiles/r05_pmc_issue.json $O/ 2>/dev/null
