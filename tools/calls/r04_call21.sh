#!/bin/bash
# round 4, call 21: 40-seed differential fuzz of the launches WITHOUT dense output against the oracle (both math modes,
# queue kernel = one wave per tile = groups of 8 tiles bit for bit)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c21
timeout -k 10 1100 python tests/fuzz_parity.py 40 1024 3 nodense > gpurun_out/c21/fuzz40_nodense.log 2>&1; echo "fuzz rc=$?"
grep -v amdgpu gpurun_out/c21/fuzz40_nodense.log | tail -4 | cut -c1-400
