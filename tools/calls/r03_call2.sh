#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c2
export HLM_PARITY_DIAG=$GRAFT_REPO_ROOT/gpurun_out/c2/parity_diag
export HLM_PARITY_RECORD=$GRAFT_REPO_ROOT/gpurun_out/c2/parity_exempt_sets.json
timeout -k 10 1000 python -m pytest tests -m gpu -q -s > gpurun_out/c2/pytest.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/c2/pytest.rc
tail -15 gpurun_out/c2/pytest.log
unset HLM_PARITY_DIAG HLM_PARITY_RECORD
V=tiger_hlm_gpu_amd/lib/variants
timeout -k 10 600 python tools/dense_variant_probe.py $V/libhlm_base.so $V/libhlm_phase.so $V/libhlm_prio1.so $V/libhlm_prio2.so $V/libhlm_r2.so 2>&1 | tee gpurun_out/c2/variants.txt
tools/pcsample.sh dense_ht host_trap 1 "--days 1 --dense-stride 1 --dense-tiled 2 --steps 2 --warmup 1 --no-variants --no-dense --no-cpu-baseline"
