#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c9
# 1. headline kernel: rocprofv3 trace + PMC passes (tools/profile_round.sh), condensed into profiles/ on the CPU side later
timeout -k 10 500 tools/profile_round.sh r03 > gpurun_out/c9/profile_round.log 2>&1; echo "profile_round rc=$?"
# 2. true cost of the wave-count tail of the dense kernel: 7.0, 7.63 (1 M) and 8.0 rounds of waves
for ns in 917504 1000000 1048576; do
  timeout -k 10 200 python tools/dense_variant_probe.py --ns $ns --reps 5 tiger_hlm_gpu_amd/lib/libhlm_hip.so 2>&1 | grep "round 1" | python -c "
import sys, json
l = sys.stdin.read(); d = json.loads(l[l.index('{'):]); print('ns', $ns, 'ms_mean', round(d['ms_mean'], 3), 'ms_min', round(d['ms_min'], 3), 'ns per hillslope', round(d['ms_mean'] * 1e6 / $ns, 3))" | tee -a gpurun_out/c9/tail.txt
done
# 3. lane utilisation of the Radau pass, restart and continue mode (cold C3)
timeout -k 10 400 tools/pmc_probe.sh r03_cold_restart "--tmin -5 --radau --steps 1 --warmup 1 --no-variants --no-dense --no-cpu-baseline" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" 2>&1 | tail -8 | tee gpurun_out/c9/cold_restart.txt
timeout -k 10 400 tools/pmc_probe.sh r03_cold_continue "--tmin -5 --radau --radau-continue --steps 1 --warmup 1 --no-variants --no-dense --no-cpu-baseline" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" 2>&1 | tail -8 | tee gpurun_out/c9/cold_continue.txt
