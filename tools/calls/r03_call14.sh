#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c14
timeout -k 10 600 python tools/pieces_probe.py 2>&1 | tee gpurun_out/c14/pieces.txt
timeout -k 10 300 python tools/pieces_probe.py --tmin -5 --pieces 0 8 2>&1 | tee gpurun_out/c14/pieces_cold.txt
timeout -k 10 300 python tools/pieces_probe.py --ns 100000 --pieces 0 4 8 2>&1 | tee gpurun_out/c14/pieces_100k.txt
timeout -k 10 200 python tools/pieces_probe.py --ns 1000 --days 3 --pieces 0 3 7 64 2>&1 | tee gpurun_out/c14/pieces_small.txt
