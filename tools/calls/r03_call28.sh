#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c28
timeout -k 10 400 python bench.py --no-dense --no-cpu-baseline > gpurun_out/c28/bench.json 2> gpurun_out/c28/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c28/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['time_pieces'], d.get('launch_shape'))
print({k:(v.get('kernel_ms') if isinstance(v,dict) else v) for k,v in d.items() if k.startswith('variant')})
print(json.dumps(d.get('cold_c3'))[:1800])
PY
timeout -k 10 200 python tools/pieces_probe.py > gpurun_out/c28/pieces.txt 2>&1; echo "pieces rc=$?"; tail -25 gpurun_out/c28/pieces.txt
timeout -k 10 500 tools/profile_round.sh r03 && python tools/summarize_profile.py gpurun_out/prof_r03 r03 > gpurun_out/c28/summary.json; echo "profile rc=$?"; head -40 gpurun_out/c28/summary.json
