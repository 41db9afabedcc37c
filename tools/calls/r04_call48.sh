#!/bin/bash
# round 4, call 48: the WHOLE GPU suite against an EXPERIMENTAL=1 build of HEAD (the quarantined paths built in: nothing skips)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c48
HLM_HIP_LIB=$PWD/tiger_hlm_gpu_amd/lib/variants/libhlm_exp.so timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/c48/pytest_experimental_all.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/c48/pytest_experimental_all.log | cut -c1-200
