#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c5
V=tiger_hlm_gpu_amd/lib/variants
timeout -k 10 500 python tools/dense_variant_probe.py $V/libhlm_fp_prio1.so $V/libhlm_hairer.so $V/libhlm_hairer_phase.so 2>&1 | tee gpurun_out/c5/variants.txt
export HLM_PARITY_RECORD=$GRAFT_REPO_ROOT/gpurun_out/c5/parity_exempt_sets.json
timeout -k 10 1000 python -m pytest tests -m gpu -q -s > gpurun_out/c5/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/c5/pytest.log
