#!/bin/bash
# round 4, call 47: the quarantined launch paths in an EXPERIMENTAL=1 build of HEAD: the 22 tests that skip in the
# default build (lane refill, compaction between launches, tile-row layout, transposing stream path)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c47
HLM_HIP_LIB=$PWD/tiger_hlm_gpu_amd/lib/variants/libhlm_exp.so timeout -k 10 900 python -m pytest tests -m gpu -q -k "lane_refill or compaction or compacting or tile_layout or tile_row or toy_model_three or host_api_defaults or stream_path or layout" > gpurun_out/c47/pytest_experimental.log 2>&1; echo "experimental pytest rc=$?"; tail -4 gpurun_out/c47/pytest_experimental.log | cut -c1-200
