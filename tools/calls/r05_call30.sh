#!/bin/bash
# round 5, call 30: "continue" mode, how many of the 16 pieces are handed out one by one at the end (the rest: one item per system)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
B="--steps 3 --warmup 1 --no-variants --no-dense --no-cpu-baseline --tmin -5 --group-tiles 8 --radau --radau-continue"
for K in 3 1 2 4 6 9 15 3; do
  HLM_RADAU_TAIL_PIECES=$K timeout -k 10 300 python bench.py $B > $O/radau_x.json 2> $O/radau_x.err || exit 1
  python -c "
import json; d=json.loads(open('$O/radau_x.json').read().strip().splitlines()[-1]); print('tail pieces $K: ms_per_step', round(d['ms_per_step'],2))"
done
