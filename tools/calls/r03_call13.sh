#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c13
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/c13/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/c13/pytest.log
timeout -k 10 300 python tests/fuzz_parity.py 3 1024 3 > gpurun_out/c13/fuzz3.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/c13/fuzz3.log | cut -c1-300
timeout -k 10 900 python bench.py > gpurun_out/c13/bench.json 2> gpurun_out/c13/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c13/bench.json').read().strip().splitlines()[-1])
o=d['dense_output']
print(d['value'], d['ms_per_step'], 'in_hbm', o['in_hbm_kernel']['kernel_ms'], o['in_hbm_kernel']['roofline']['frac'], 'c5', o['c5_share']['frac_of_hbm_spec'], 'streamed', o['streamed']['dense_output_MBps'], 'e2e', d['driver_e2e'].get('process_wall_s'))
PY
