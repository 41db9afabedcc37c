#!/bin/bash
# round 5, call 19: the dense section of the bench with the sharded-streamed row
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 600 python bench.py --no-variants --no-cpu-baseline --steps 2 > $O/bench_d.json 2> $O/bench_d.err; echo "rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/bench_d.json").read().strip().splitlines()[-1])["dense_output"]
print(json.dumps(d.get("sharded_streamed")), d["streamed"]["dense_output_MBps"], d["in_hbm_kernel"]["roofline"])
PY
