#!/bin/bash
# round 5, call 12: dense kernel experiments -- issue-slot sensitivity (extra SALU / VALU per attempt), output selection
# with K <= 2 at two and three waves per SIMD (same box A/B: tools/dense_variant_probe.py)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
V=tiger_hlm_gpu_amd/lib/variants
timeout -k 10 500 python tools/dense_variant_probe.py --reps 4 $V/libhlm_d0.so $V/libhlm_ds40.so $V/libhlm_ds80.so $V/libhlm_dv40.so $V/libhlm_dv80.so 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee $O/dense_slope.txt
timeout -k 10 400 python tools/dense_variant_probe.py --reps 4 --states 1 $V/libhlm_d0.so $V/libhlm_k2w2.so $V/libhlm_k2w3.so 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee $O/dense_sel_k.txt
