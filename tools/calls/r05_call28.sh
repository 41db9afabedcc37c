#!/bin/bash
# round 5, call 28: the whole GPU suite on HEAD (Radau queue + step memory), then the default bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $O/pytest_full.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest_full.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05/bench_line.json").read().strip().splitlines()[-1])
print(d["metric"], d["value"], d["ms_per_step"], json.dumps(d["roofline"])[:300])
c = d.get("variant_survey_c3_cold_cells", {})
print({k: (round(v["ms"], 2) if isinstance(v, dict) and "ms" in v else None) for k, v in c.items()})
PY
