#!/bin/bash
# round 4, call 41: groups of tiles packed per RUN instead of per piece (queue_group_tiles with a run schedule) on the
# three workloads
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c41
for W in "" "--tmin -5" "--wet"; do
  echo "## workload: ${W:-headline}" >> gpurun_out/c41/group_runs.txt
  timeout -k 10 500 python tools/pieces_probe.py $W --reps 4 --pieces 32 --runs 0 -8 -4 --groups 0 8 >> gpurun_out/c41/group_runs.txt 2>&1 || exit 1
done
cat gpurun_out/c41/group_runs.txt
