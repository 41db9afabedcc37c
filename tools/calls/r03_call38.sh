#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c38
timeout -k 10 400 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "streamed_forcings" > gpurun_out/c38/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/c38/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py --no-dense --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/c38/bench.json 2> gpurun_out/c38/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c38/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
print(json.dumps(d.get('variant_forcings_streamed')))
print(json.dumps(d.get('driver_e2e'))[:600])
PY
