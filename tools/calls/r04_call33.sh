#!/bin/bash
# round 4, call 33: the automatic run schedule with a cap on the run length (queue_run_pieces = -n) on the three
# workloads: headline, SURVEY's cold cells, wet surface
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c33
for W in "" "--tmin -5" "--wet"; do
  echo "## workload: ${W:-headline}" >> gpurun_out/c33/capped_runs.txt
  timeout -k 10 400 python tools/pieces_probe.py $W --reps 5 --pieces 32 --runs 0 -8 -4 1 0 >> gpurun_out/c33/capped_runs.txt 2>&1 || exit 1
done
cat gpurun_out/c33/capped_runs.txt
