#!/bin/bash
# round 5, call 21: Radau pass with time pieces (items = (piece, list entry)): the Radau tests, then A/B of the pieces on SURVEY's cold C3
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_output_selection_gpu.py -m gpu -q -x -k "radau" > $O/pytest_radau.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest_radau.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
B="--steps 3 --warmup 1 --no-variants --no-dense --no-cpu-baseline --tmin -5 --group-tiles 8"
for P in 1 16 8; do
  HLM_RADAU_PIECES=$P timeout -k 10 300 python bench.py $B --radau > $O/radau_p$P.json 2> $O/radau_p$P.err || exit 1
  python -c "
import json; d=json.loads(open('$O/radau_p$P.json').read().strip().splitlines()[-1]); print('restart pieces $P: ms_per_step', round(d['ms_per_step'],2))"
done
timeout -k 10 300 python bench.py $B --radau --radau-continue > $O/radau_cont.json 2> $O/radau_cont.err || exit 1
python -c "
import json; d=json.loads(open('$O/radau_cont.json').read().strip().splitlines()[-1]); print('continue: ms_per_step', round(d['ms_per_step'],2))"
timeout -k 10 300 python bench.py $B > $O/cold_rk45.json 2> $O/cold_rk45.err || exit 1
python -c "
import json; d=json.loads(open('$O/cold_rk45.json').read().strip().splitlines()[-1]); print('rk45 alone: ms_per_step', round(d['ms_per_step'],2))"
