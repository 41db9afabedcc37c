#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c26
timeout -k 5 300 python tools/queue_hang_probe.py gpurun_out/c26/log.txt; rc=$?; echo "probe rc=$rc"
grep -v "^ok" gpurun_out/c26/log.txt | cut -c1-300; grep -c "^ok" gpurun_out/c26/log.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "time_pieces or lane_refill or radau_continue" > gpurun_out/c26/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 gpurun_out/c26/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py --no-dense --no-cpu-baseline > gpurun_out/c26/bench.json 2> gpurun_out/c26/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c26/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['time_pieces'], d.get('launch_shape'))
print({k:(v.get('kernel_ms') if isinstance(v,dict) else v) for k,v in d.items() if k.startswith('variant')})
print(json.dumps(d.get('cold_c3'))[:1500])
PY
