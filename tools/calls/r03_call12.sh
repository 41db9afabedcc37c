#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c12
V=tiger_hlm_gpu_amd/lib/variants
timeout -k 10 400 python tools/dense_variant_probe.py --reps 5 $V/libhlm_default.so $V/libhlm_unr2.so 2>&1 | tee gpurun_out/c12/variants.txt
timeout -k 10 300 python bench.py --scaling strong --total-ns 4000000 --no-variants --no-dense --no-cpu-baseline > gpurun_out/c12/c4_4M.json 2> gpurun_out/c12/c4.err; echo "c4 rc=$?"
python -c "
import json; d=json.loads(open('gpurun_out/c12/c4_4M.json').read().strip().splitlines()[-1]); print('C4 4M one GPU', d['value'], d['ms_per_step'], d['roofline_valu_fp64']['frac'])"
