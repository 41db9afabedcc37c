#!/bin/bash
# round 5, call 23: Radau pass with time pieces -- instruction-level PMC of the new kernel (cold C3: RK45, + Radau restart, + Radau continue), break batch of 16
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
B="--steps 3 --warmup 1 --no-variants --no-dense --no-cpu-baseline --tmin -5 --group-tiles 8"
V=$PWD/tiger_hlm_gpu_amd/lib/variants
for T in b16 b8; do
for M in "" "--radau-continue"; do
  HLM_HIP_LIB=$V/libhlm_$T.so timeout -k 10 300 python bench.py $B --radau $M > $O/radau_x.json 2> $O/radau_x.err || exit 1
  python -c "
import json; d=json.loads(open('$O/radau_x.json').read().strip().splitlines()[-1]); print('$T $M: ms_per_step', round(d['ms_per_step'],2))"
done; done
timeout -k 10 900 bash tools/pmc_issue.sh r05b "cold cold_radau cold_radau_continue" > $O/pmc_issue_r05b.log 2>&1; echo "pmc rc=$?"; tail -60 $O/pmc_issue_r05b.log | grep -v "^ *\"flop" | cut -c1-200
cp profiles/r05b_pmc_issue.json gpurun_out/r05/ 2>/dev/null
