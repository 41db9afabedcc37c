#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c39
HLM_HIP_LIB=$PWD/tiger_hlm_gpu_amd/lib/variants/libhlm_rctr.so timeout -k 10 300 python tools/radau_divergence_probe.py > gpurun_out/c39/div.txt 2>&1; echo "div rc=$?"; grep -v amdgpu.ids gpurun_out/c39/div.txt
timeout -k 10 900 python -m pytest tests -m gpu -q -k "radau" > gpurun_out/c39/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -25 gpurun_out/c39/pytest.log | cut -c1-220
timeout -k 10 300 python bench.py --no-dense --no-cpu-baseline --steps 1 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['variant_survey_c3_cold_cells']
print({k:(c[k]['ms'], c[k].get('implicit_attempts'), c[k].get('unsolved_after')) for k in ('rk45','rk45_plus_radau','rk45_plus_radau_continue')})"
