#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c30
for v in trim1 trim0 trim1 trim0; do
  if [ $v = trim0 ]; then export HLM_HIP_LIB=$PWD/tiger_hlm_gpu_amd/lib/variants/libhlm_qtrim0.so; else unset HLM_HIP_LIB; fi
  echo "## $v" >> gpurun_out/c30/ab.txt
  timeout -k 10 200 python tools/pieces_probe.py --pieces 0 15 15 --reps 4 >> gpurun_out/c30/ab.txt 2>&1 || exit 1
  echo "## $v cold" >> gpurun_out/c30/ab.txt
  timeout -k 10 200 python tools/pieces_probe.py --pieces 0 15 --tmin -5 --reps 3 >> gpurun_out/c30/ab.txt 2>&1 || exit 1
done
grep -v amdgpu.ids gpurun_out/c30/ab.txt
