#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c49
timeout -k 10 900 python bench.py > gpurun_out/c49/bench.json 2> gpurun_out/c49/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c49/bench.json').read().strip().splitlines()[-1])
o=d['dense_output']
print(d['value'], d['ms_per_step'], d['roofline_valu_fp64']['frac'], 'in_hbm', o['in_hbm_kernel']['kernel_ms'], o['in_hbm_kernel']['roofline']['frac'], 'c5', o['c5_share']['frac_of_hbm_spec'], 'streamed', o['streamed']['dense_output_MBps'], 'e2e', d['driver_e2e'].get('process_wall_s'), d['driver_e2e'].get('first_process_wall_s'))
print(d.get('launch_shape'))
print(json.dumps(d.get('variant_forcings_streamed'))[:300])
print(d['gpu_over_cpu'] if 'gpu_over_cpu' in d else None, json.dumps(d.get('cpu_baseline'))[:200])
PY
