#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c47
timeout -k 10 500 python bench.py --no-dense --no-cpu-baseline > gpurun_out/c47/bench.json 2> gpurun_out/c47/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c47/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline_valu_fp64']['frac'], d['launch_shape']['kernel_ms_each'])
c=d['variant_survey_c3_cold_cells']
print({k:c[k]['ms'] for k in ('rk45','rk45_plus_radau','rk45_plus_radau_continue','rk45_compacting')})
print({k:(v.get('kernel_ms') or v.get('wall_ms')) for k,v in d.items() if k.startswith('variant_') and isinstance(v,dict)})
print(d['c2_dummy']['adaptive']['kernel_ms'], d['driver_e2e']['phases_s']['integrate_s'])
PY
for cfg in "50000 8 -5" "50000 8 0.5" "100000 2 -5"; do set -- $cfg; echo "## ns $1 days $2 tmin $3"; timeout -k 10 100 python tools/pieces_probe.py --ns $1 --days $2 --tmin $3 --reps 5 --pieces 0 16 2>&1 | grep -v amdgpu.ids | cut -c1-70; done
