#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c6
V=tiger_hlm_gpu_amd/lib/variants
timeout -k 10 500 python tools/dense_variant_probe.py $V/libhlm_hairer.so $V/libhlm_hairer_l2.so $V/libhlm_hairer_l3.so $V/libhlm_hairer_pow2.so 2>&1 | tee gpurun_out/c6/variants.txt
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "radau_continue or sink" > gpurun_out/c6/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/c6/pytest.log
