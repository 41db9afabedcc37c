#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c8
export HLM_PARITY_RECORD=$GRAFT_REPO_ROOT/gpurun_out/c8/parity_exempt_sets.json
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/c8/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/c8/pytest.log
unset HLM_PARITY_RECORD
timeout -k 10 700 tools/profile_dense.sh r03_dense_final > gpurun_out/c8/profile_dense.log 2>&1; echo "profile rc=$?"; tail -30 gpurun_out/c8/profile_dense.log
timeout -k 10 900 python bench.py > gpurun_out/c8/bench.json 2> gpurun_out/c8/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/c8/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c8/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
o=d['dense_output']
print('in_hbm', o['in_hbm_kernel']['kernel_ms'], o['in_hbm_kernel']['roofline']['frac'], o['in_hbm_kernel']['kernel_ms_each'])
print('streamed', o['streamed']['dense_output_MBps'], o['streamed']['one_state_selected_on_device']['wall_speedup_vs_all_states'])
print('c5', o['c5_share']['kernel_ms_sum'], o['c5_share']['frac_of_hbm_spec'])
print('e2e', d['driver_e2e'].get('phases_s'), d['driver_e2e'].get('process_wall_s'))
PY
