#!/bin/bash
# is the C5 share's gap to the one-launch kernel wave-count quantisation?  per-window kernel ms by hillslope count
set -o pipefail
cd "$GRAFT_REPO_ROOT"
for ns in 458752 500000 524288 589824 1000000; do
  timeout -k 10 300 python bench.py --no-variants --no-cpu-baseline --steps 1 --warmup 0 --dense-ns 65536 --dense-hbm-ns $ns --c5-ns $ns --c5-days 4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['dense_output']
c=d['c5_share']; h=d['in_hbm_kernel']
k=c['kernel_ms_each'][:-1]
print('ns', $ns, 'tiles/2048 = %.3f' % ($ns/64/2048), 'window ms', [round(x,2) for x in k], 'us per 1000 hillslopes: %.3f' % (1e3*sum(k[1:])/len(k[1:])/$ns*1000), '| one-launch plain kernel %.2f ms = %.3f us per 1000' % (h['kernel_ms'], 1e3*h['kernel_ms']/$ns*1000))
" || exit 1
done
