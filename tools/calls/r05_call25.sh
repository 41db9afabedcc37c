#!/bin/bash
# round 5, call 25: LLVM scheduling strategies (whole library built with them; the Radau pass = total - RK45 alone of the same build)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
B="--steps 3 --warmup 1 --no-variants --no-dense --no-cpu-baseline --tmin -5 --group-tiles 8"
V=$PWD/tiger_hlm_gpu_amd/lib/variants
for T in default ilp iilp; do
  L=$V/libhlm_$T.so; [ $T = default ] && L=$PWD/tiger_hlm_gpu_amd/lib/libhlm_hip.so
  for M in "--radau" "--radau --radau-continue" ""; do
    HLM_HIP_LIB=$L timeout -k 10 300 python bench.py $B $M > $O/radau_x.json 2> $O/radau_x.err || exit 1
    python -c "
import json; d=json.loads(open('$O/radau_x.json').read().strip().splitlines()[-1]); print('$T [$M]: ms_per_step', round(d['ms_per_step'],2))"
  done
done
