#!/bin/bash
# round 5, call 24: Radau pass, defaults settled (16 pieces restarting, whole systems continuing, breaks of 16 lanes): Radau tests incl. the piece-count invariance, cold C3 timings
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_output_selection_gpu.py -m gpu -q -x -k "radau or toy" > $O/pytest_radau.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest_radau.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
B="--steps 3 --warmup 1 --no-variants --no-dense --no-cpu-baseline --tmin -5 --group-tiles 8"
for M in "" "--radau-continue"; do
  timeout -k 10 300 python bench.py $B --radau $M > $O/radau_x.json 2> $O/radau_x.err || exit 1
  python -c "
import json; d=json.loads(open('$O/radau_x.json').read().strip().splitlines()[-1]); print('default $M: ms_per_step', round(d['ms_per_step'],2))"
done
timeout -k 10 300 python bench.py $B > $O/radau_x.json 2> $O/radau_x.err || exit 1
python -c "
import json; d=json.loads(open('$O/radau_x.json').read().strip().splitlines()[-1]); print('rk45 alone: ms_per_step', round(d['ms_per_step'],2))"
