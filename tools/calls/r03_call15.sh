#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c15
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "time_pieces or lane_refill or radau_continue" > gpurun_out/c15/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/c15/pytest.log
timeout -k 10 400 python bench.py --no-dense --no-cpu-baseline > gpurun_out/c15/bench.json 2> gpurun_out/c15/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c15/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['time_pieces'], d.get('launch_shape'))
print({k:(v.get('kernel_ms') if isinstance(v,dict) else v) for k,v in d.items() if k.startswith('variant')})
print(json.dumps(d.get('cold_c3'))[:1500])
PY
timeout -k 10 500 tools/profile_round.sh r03 && python tools/summarize_profile.py gpurun_out/prof_r03 r03 > gpurun_out/c15/summary.json; echo "profile rc=$?"; tail -30 gpurun_out/c15/summary.json
