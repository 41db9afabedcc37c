#!/bin/bash
# round 4, call 63: the default bench line once more (call 62's config-5 share had one window of 31 at 29 ms)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c63 && timeout -k 10 800 python bench.py > gpurun_out/c63/bench_line.json 2> gpurun_out/c63/bench.err; echo rc=$?
