#!/bin/bash
# round 5, call 2: why the GPU (STRICT) does not flag 6 hillslopes of intermittent-50k the oracle flags
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
for m in 0 1; do timeout -k 10 300 python tests/landing_replay_probe.py $m 4366 14558 17382 25770 26809 36323 2>&1 | grep -v amdgpu.ids | cut -c1-900; done | tee $O/landing_replay.txt
