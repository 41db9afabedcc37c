#!/bin/bash
# round 5, call 20: streamed pipelines after the buffer reuse: sharded tests, streaming tests, the bench's dense section
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_sharded_gpu.py tests/test_output_selection_gpu.py tests/test_parity_gpu.py -m gpu -q -x -k "sharded or streamed or stream or sink or window" > $O/pytest_stream.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_stream.log | cut -c1-300
timeout -k 10 600 python bench.py --no-variants --no-cpu-baseline --steps 2 > $O/bench_e.json 2> $O/bench_e.err; echo "rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/bench_e.json").read().strip().splitlines()[-1])["dense_output"]
print(json.dumps(d.get("sharded_streamed")), d["streamed"]["dense_output_MBps"])
PY
