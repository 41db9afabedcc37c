#!/bin/bash
# round 5, call 10: Radau pass -- lane refill in per-wave chunks of the flag list against entry-by-entry (same box A/B)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
for v in rc0 rc64 rc256 rc1024 rc0; do
  HLM_HIP_LIB=$PWD/tiger_hlm_gpu_amd/lib/variants/libhlm_$v.so timeout -k 10 200 python tools/radau_machine_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-400 || exit 1
done | tee $O/radau_chunk_probe.txt
for v in rc0c rc256c; do
  echo "== $v"; HLM_HIP_LIB=$PWD/tiger_hlm_gpu_amd/lib/variants/libhlm_$v.so timeout -k 10 300 python tools/radau_divergence_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 || exit 1
done | tee $O/radau_chunk_sections.txt
