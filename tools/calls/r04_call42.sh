#!/bin/bash
# round 4, call 42: groups packed per run (bit identity, 6 cases); then the 40-seed differential fuzz of the launches
# WITHOUT dense output against the oracle with the correctly rounded STRICT powers (both math modes)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c42
timeout -k 10 600 python -m pytest tests -m gpu -q -k "group_tiles" > gpurun_out/c42/group.log 2>&1; echo "group rc=$?"; tail -2 gpurun_out/c42/group.log
timeout -k 10 1000 python tests/fuzz_parity.py 40 1024 3 nodense > gpurun_out/c42/fuzz40_nodense.log 2>&1; echo "fuzz rc=$?"
grep -v amdgpu gpurun_out/c42/fuzz40_nodense.log | tail -2 | cut -c1-400
