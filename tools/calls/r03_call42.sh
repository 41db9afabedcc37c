#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c42
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/c42/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/c42/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c42/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/c42/smoke.log
timeout -k 10 900 python bench.py > gpurun_out/c42/bench.json 2> gpurun_out/c42/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c42/bench.json').read().strip().splitlines()[-1])
o=d['dense_output']
print(d['value'], d['ms_per_step'], d['roofline_valu_fp64']['frac'], 'in_hbm', o['in_hbm_kernel']['kernel_ms'], o['in_hbm_kernel']['roofline']['frac'], 'c5', o['c5_share']['frac_of_hbm_spec'], 'streamed', o['streamed']['dense_output_MBps'], 'e2e', d['driver_e2e'].get('process_wall_s'), d['driver_e2e'].get('first_process_wall_s'))
print(d.get('launch_shape'))
print(json.dumps(d.get('variant_forcings_streamed')))
PY
