#!/bin/bash
# round 4, call 1: the new headline-kernel parity tests (recording their exemption sets), smoke, and the
# instruction-level PMC passes of the headline kernel
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c1
export HLM_PARITY_RECORD=$PWD/gpurun_out/c1/exempt_sets_new.json
timeout -k 10 900 python -m pytest tests/test_headline_kernel_gpu.py -m gpu -q -x -s > gpurun_out/c1/pytest_headline.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids gpurun_out/c1/pytest_headline.log | tail -60 | cut -c1-300
unset HLM_PARITY_RECORD
timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -8 &&
timeout -k 10 900 bash tools/pmc_issue.sh r04 2>&1 | tail -80
