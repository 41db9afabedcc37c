#!/bin/bash
# round 4, call 35: where the device's STRICT powers round wrongly (seed residuals by range)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c35
timeout -k 10 600 python tools/strict_pow_probe.py > gpurun_out/c35/strict_pow_probe.txt 2>&1; echo "rc=$?"; cat gpurun_out/c35/strict_pow_probe.txt
