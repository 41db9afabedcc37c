#!/bin/bash
# round 5, call 9: sharded entry (staged uploads, gather plan, dense through the sharded entry), then the full bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_sharded_gpu.py -m gpu -q -x -s > $O/pytest_sharded.log 2>&1; echo "pytest sharded rc=$?"; grep -v amdgpu.ids $O/pytest_sharded.log | grep "upload\|passed\|failed\|Error\|shard(s)" | cut -c1-400
timeout -k 10 900 python bench.py > $O/bench_b.json 2> $O/bench_b.err; echo "bench rc=$?"; tail -3 $O/bench_b.err | cut -c1-300
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/bench_b.json").read().strip().splitlines()[-1])
for k in ("c4_shard", "sharded_upload"):
    print(k, json.dumps(d.get(k))[:1500])
print("headline", d["ms_per_step"], d["roofline"]["frac"])
PY
