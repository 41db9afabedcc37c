#!/bin/bash
# round 5, call 27: Radau tests after the FAST NaN guard through Z; cold C3: launch shapes of the RK45 pass (group tiles x time pieces), Radau restart / continue
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_output_selection_gpu.py -m gpu -q -x -k "radau or toy" > $O/pytest_radau.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_radau.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
B="--steps 3 --warmup 1 --no-variants --no-dense --no-cpu-baseline --tmin -5"
for M in "--group-tiles 8 --radau" "--group-tiles 8 --radau --radau-continue" "--group-tiles 8" "--group-tiles 16" "--group-tiles 8 --time-pieces 64" "--group-tiles 16 --time-pieces 64" "--group-tiles 4" "--group-tiles 8 --time-pieces 16"; do
  timeout -k 10 300 python bench.py $B $M > $O/radau_x.json 2> $O/radau_x.err || { tail -3 $O/radau_x.err; continue; }
  python -c "
import json; d=json.loads(open('$O/radau_x.json').read().strip().splitlines()[-1]); print('[$M]: ms_per_step', round(d['ms_per_step'],2))"
done
