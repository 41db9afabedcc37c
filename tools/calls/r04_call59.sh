#!/bin/bash
# round 4, call 59: A/B of the FAST x^(2/3) NaN for negative arguments as two integer instructions in the queue kernel: the library
# before (variants/libhlm_base.so) and after, alternately on one box: headline, wet surface, cold cells
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c59
for rep in 1 2; do
for W in "" "--wet" "--tmin -5"; do
  for LIB in variants/libhlm_base.so libhlm_hip.so; do
    echo "## ${W:-headline} $LIB" >> gpurun_out/c59/ab.txt
    HLM_HIP_LIB=$PWD/tiger_hlm_gpu_amd/lib/$LIB timeout -k 10 300 python tools/pieces_probe.py $W --reps 5 --pieces 32 --runs 0 2>&1 | grep time_pieces >> gpurun_out/c59/ab.txt || exit 1
  done
done
done
cat gpurun_out/c59/ab.txt
timeout -k 10 600 python -m pytest tests -m gpu -q -k "time_pieces or headline or no_dense or elementary or single_evaluations" > gpurun_out/c59/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/c59/tests.log
