#!/bin/bash
# round 4, call 36: the intermediates of the STRICT x^(2/3) on the device against the host, same seed
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c36
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=fast -I tiger_hlm_gpu_amd/csrc tools/debug/strict_pow_debug.hip -o /tmp/spd > gpurun_out/c36/build.log 2>&1 || { cat gpurun_out/c36/build.log; exit 1; }
timeout -k 10 120 /tmp/spd > gpurun_out/c36/trace.txt 2>&1; echo "rc=$?"; cat gpurun_out/c36/trace.txt
