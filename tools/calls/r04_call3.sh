#!/bin/bash
# round 4, call 3: where do the waves of the queue launch spend their time, by run schedule (cold cells and headline)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export HLM_HIP_LIB=$PWD/tiger_hlm_gpu_amd/lib/variants/libhlm_qprof.so
timeout -k 10 400 python tools/queue_prof_probe.py --tmin -5 --sched 1 auto 2 4 16,8,4,2,1,1 8,8,8,4,2,2 4,4,4,4,4,4,4,2,2 2>&1 | grep -v amdgpu.ids | cut -c1-330 &&
timeout -k 10 300 python tools/queue_prof_probe.py --sched 1 auto 8,8,8,4,2,2 2>&1 | grep -v amdgpu.ids | cut -c1-330 &&
timeout -k 10 300 python tools/queue_prof_probe.py --ns 100000 --tmin -5 --sched 1 auto 2 4 2>&1 | grep -v amdgpu.ids | cut -c1-330
