#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c7
V=tiger_hlm_gpu_amd/lib/variants
timeout -k 10 400 python tools/dense_variant_probe.py $V/libhlm_default.so $V/libhlm_unrolled.so 2>&1 | tee gpurun_out/c7/variants.txt
timeout -k 10 700 tools/profile_dense.sh r03_dense > gpurun_out/c7/profile_dense.log 2>&1; echo "profile rc=$?"; tail -40 gpurun_out/c7/profile_dense.log
