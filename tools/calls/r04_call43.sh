#!/bin/bash
# round 4, call 43: the 40-seed differential fuzz WITH dense output (dense kernels; both math modes) against the oracle
# with the correctly rounded STRICT powers
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c43
timeout -k 10 1100 python tests/fuzz_parity.py 40 1024 3 > gpurun_out/c43/fuzz40.log 2>&1; echo "fuzz rc=$?"
grep -v amdgpu gpurun_out/c43/fuzz40.log | tail -2 | cut -c1-400
