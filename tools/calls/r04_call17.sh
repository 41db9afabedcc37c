#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_headline_kernel_gpu.py -m gpu -q -x -k "compact_windows" 2>&1 | tail -2
timeout -k 10 300 python tools/pieces_probe.py --tmin -5 --pieces 32 64 --runs 1 --groups 0 8 16 2>&1 | grep -v amdgpu.ids | cut -c1-190
export HLM_HIP_LIB=$PWD/tiger_hlm_gpu_amd/lib/variants/libhlm_qprof.so
timeout -k 10 300 python tools/queue_prof_probe.py --tmin -5 --sched 1 --group 8 2>&1 | grep -v amdgpu.ids | cut -c1-330
