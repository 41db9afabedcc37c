#!/bin/bash
# round 5, call 1: the new saturating / counting / group-list tests (recording exemption sets), the group-list stress
# under round 4's list rule (expected to FAIL), a default bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
export HLM_PARITY_RECORD=$PWD/$O/exempt_sets_new.json
timeout -k 10 900 python -m pytest tests/test_headline_kernel_gpu.py -m gpu -q -x -s -k "saturating or group_passes or group_tiles" > $O/pytest_new.log 2>&1; echo "pytest new rc=$?"; grep -v amdgpu.ids $O/pytest_new.log | tail -25 | cut -c1-400
unset HLM_PARITY_RECORD
HLM_HIP_LIB=$PWD/tiger_hlm_gpu_amd/lib/variants/libhlm_oldlist.so timeout -k 10 300 python -m pytest tests/test_headline_kernel_gpu.py -m gpu -q -x -k "group_passes" > $O/pytest_oldlist.log 2>&1; echo "pytest oldlist rc=$? (expected non-zero)"; grep -v amdgpu.ids $O/pytest_oldlist.log | tail -8 | cut -c1-600
timeout -k 10 600 python bench.py > $O/bench_a.json 2> $O/bench_a.err; echo "bench rc=$?"; cut -c1-1500 $O/bench_a.json
