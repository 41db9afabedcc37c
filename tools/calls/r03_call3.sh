#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c3
free -g | head -2; nproc; cat /sys/fs/cgroup/memory.max 2>/dev/null
V=tiger_hlm_gpu_amd/lib/variants
timeout -k 10 500 python tools/dense_variant_probe.py $V/libhlm_oldflush.so $V/libhlm_fp.so $V/libhlm_fp_prio1.so $V/libhlm_fp_phase.so 2>&1 | tee gpurun_out/c3/variants.txt
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "dense or streamed or selection or tile or window" > gpurun_out/c3/pytest_dense.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/c3/pytest_dense.log
timeout -k 10 300 tools/pmc_probe.sh r03_dense_clock "--days 1 --dense-stride 1 --dense-tiled 2 --steps 3 --warmup 1 --no-variants --no-dense --no-cpu-baseline" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" 2>&1 | tail -8
timeout -k 10 900 python bench.py > gpurun_out/c3/bench.json 2> gpurun_out/c3/bench.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/c3/bench.json; tail -5 gpurun_out/c3/bench.err
