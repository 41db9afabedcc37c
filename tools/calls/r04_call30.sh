#!/bin/bash
# round 4, call 30: launch shapes on the wet-surface workload (automatic runs against single pieces and uniform runs)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c30
timeout -k 10 600 python tools/pieces_probe.py --wet --pieces 0 32 --runs 0 1 2 4 8 > gpurun_out/c30/wet_runs.txt 2>&1; echo "rc=$?"; cat gpurun_out/c30/wet_runs.txt
