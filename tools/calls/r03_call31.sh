#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c31
timeout -k 10 300 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -x -k "launch_shapes" > gpurun_out/c31/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/c31/pytest.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 900 python bench.py > gpurun_out/c31/bench.json 2> gpurun_out/c31/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c31/bench.json').read().strip().splitlines()[-1])
o=d['dense_output']
print(d['value'], d['ms_per_step'], d['roofline_valu_fp64']['frac'], 'in_hbm', o['in_hbm_kernel']['kernel_ms'], o['in_hbm_kernel']['roofline']['frac'], 'c5', o['c5_share']['frac_of_hbm_spec'], 'streamed', o['streamed']['dense_output_MBps'], 'e2e', d['driver_e2e'].get('process_wall_s'))
print(d.get('launch_shape'))
print(json.dumps(d.get('variant_survey_c3_cold_cells'))[:1500])
print(json.dumps(d.get('cpu_baseline'))[:400])
PY
