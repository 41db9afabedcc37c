#!/bin/bash
# round 4, call 37: STRICT powers with the opaque leading product: intermediates against the host, wrong roundings by
# range, the math self-test, then a recording run of the parity tests whose STRICT exempt sets move with the powers
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c37
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=fast -I tiger_hlm_gpu_amd/csrc tools/debug/strict_pow_debug.hip -o /tmp/spd > gpurun_out/c37/build.log 2>&1 || { cat gpurun_out/c37/build.log; exit 1; }
timeout -k 10 120 /tmp/spd > gpurun_out/c37/trace.txt 2>&1; echo "trace rc=$?"; tail -4 gpurun_out/c37/trace.txt
timeout -k 10 600 python tools/strict_pow_probe.py > gpurun_out/c37/strict_pow_probe.txt 2>&1; echo "probe rc=$?"; cat gpurun_out/c37/strict_pow_probe.txt
timeout -k 10 300 python -m pytest tests -m gpu -q -k "elementary_functions" > gpurun_out/c37/math.log 2>&1; echo "math rc=$?"; tail -3 gpurun_out/c37/math.log
HLM_PARITY_RECORD=$PWD/gpurun_out/c37/exempt_sets.json timeout -k 10 900 python -m pytest tests -m gpu -q -k "fuzz_parity or wide_parameter or toy_model_three or no_dense_launches_vs_oracle or cold" > gpurun_out/c37/pytest_record.log 2>&1; echo "record rc=$?"; tail -8 gpurun_out/c37/pytest_record.log
