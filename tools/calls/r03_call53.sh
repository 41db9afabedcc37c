#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c53
for v in base wetu base wetu; do
  if [ $v = wetu ]; then export HLM_HIP_LIB=$PWD/tiger_hlm_gpu_amd/lib/variants/libhlm_wetu.so; else unset HLM_HIP_LIB; fi
  echo "## $v"
  timeout -k 10 120 python tools/wet_probe.py 2>&1 | grep -v amdgpu.ids
done
export HLM_HIP_LIB=$PWD/tiger_hlm_gpu_amd/lib/variants/libhlm_wetu.so
timeout -k 10 200 python tools/pieces_probe.py --pieces 0 32 2>&1 | grep -v amdgpu.ids | cut -c1-110
timeout -k 10 200 python tools/pieces_probe.py --wet --pieces 0 32 2>&1 | grep -v amdgpu.ids | cut -c1-110
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "time_pieces or launch_shapes or lane_refill or compaction or c4_four or closed_form or checkpoint or fuzz" > gpurun_out/c53/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/c53/pytest.log | cut -c1-200
