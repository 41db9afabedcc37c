#!/bin/bash
# quick kernel timing on the GPU box: headline (fast, dry), wet variant, strict -- no dense / CPU legs
cd "$(dirname "$0")/.."
for m in fast strict; do
  python bench.py --math $m --no-dense --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
v=d.get('variant_wet_surface',{})
print('$m', 'dry ms', round(d['ms_per_step'],2), 'steps/s %.3e'%d['value'], '| wet', ' '.join('%s=%s'%(k,v[k]) for k in v if k in ('value','ms_per_step')))
"
done
