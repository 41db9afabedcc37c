#!/bin/bash
# round 4, call 11: section occupancy of the Radau state machine at its default threshold (counter build), then the
# instruction-level PMC passes incl. the Radau kernel itself (cold_radau / cold_radau_continue)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c11
HLM_HIP_LIB=$PWD/tiger_hlm_gpu_amd/lib/variants/libhlm_rc6.so timeout -k 10 300 python tools/radau_divergence_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c11/radau_sections.txt
timeout -k 10 1000 bash tools/pmc_issue.sh r04c > gpurun_out/c11/pmc_issue.log 2>&1; echo "pmc rc=$?"; grep "^done\|failed" gpurun_out/c11/pmc_issue.log
# the quarantined paths still pass their bit-identity tests in an EXPERIMENTAL=1 build
HLM_HIP_LIB=$PWD/tiger_hlm_gpu_amd/lib/variants/libhlm_exp.so timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "lane_refill or compaction or tile_layout or tile_row or toy_model_three or host_api_defaults" > gpurun_out/c11/pytest_experimental.log 2>&1; echo "experimental pytest rc=$?"; tail -3 gpurun_out/c11/pytest_experimental.log | cut -c1-200
