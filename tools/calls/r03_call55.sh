#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c55
timeout -k 10 900 python bench.py > gpurun_out/c55/bench.json 2> gpurun_out/c55/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c55/bench.json').read().strip().splitlines()[-1])
o=d['dense_output']
print(d['value'], d['ms_per_step'], d['roofline_valu_fp64']['frac'], 'in_hbm', o['in_hbm_kernel']['kernel_ms'], o['in_hbm_kernel']['roofline']['frac'], 'c5', o['c5_share']['frac_of_hbm_spec'], 'streamed', o['streamed']['dense_output_MBps'], 'e2e', d['driver_e2e'].get('process_wall_s'))
print(d.get('launch_shape')['kernel_ms_each'])
c=d['variant_survey_c3_cold_cells']
print({k:c[k]['ms'] for k in ('rk45','rk45_plus_radau','rk45_plus_radau_continue')})
print({k:(v.get('kernel_ms') or v.get('wall_ms')) for k,v in d.items() if k.startswith('variant_') and isinstance(v,dict)})
print(d['c2_dummy']['adaptive']['kernel_ms'], d['gpu_over_cpu'])
PY
