#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c44
timeout -k 10 500 python -m pytest tests -m gpu -q -x -k "time_pieces or launch_shapes or radau_continue" > gpurun_out/c44/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/c44/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py --no-dense --no-cpu-baseline > gpurun_out/c44/bench.json 2> gpurun_out/c44/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c44/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['time_pieces'], d['launch_shape']['kernel_ms_each'])
c=d['variant_survey_c3_cold_cells']
print({k:c[k]['ms'] for k in ('rk45','rk45_plus_radau','rk45_plus_radau_continue')})
print({k:(v.get('kernel_ms') if isinstance(v,dict) else v) for k,v in d.items() if k.startswith('variant_') and isinstance(v,dict) and 'kernel_ms' in v})
PY
