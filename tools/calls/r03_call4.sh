#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c4
V=tiger_hlm_gpu_amd/lib/variants
timeout -k 10 500 python tools/dense_variant_probe.py $V/libhlm_fp_prio1.so $V/libhlm_drain_prio1.so $V/libhlm_drain_prio1_l3.so $V/libhlm_drain_phase.so 2>&1 | tee gpurun_out/c4/variants.txt
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "radau or kernel_register or selection_with_flagged" > gpurun_out/c4/pytest_radau.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/c4/pytest_radau.log
timeout -k 10 600 python bench.py --no-dense --no-cpu-baseline > gpurun_out/c4/bench.json 2> gpurun_out/c4/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/c4/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c4/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
print(json.dumps(d['variant_survey_c3_cold_cells'], indent=1)[:1800])
PY
