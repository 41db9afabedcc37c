#!/bin/bash
# round 4, call 4: the native sharded run with the RCCL gather (one device: communicator + self send/recv), the driver's
# --gather route; log for profiles/
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c4
export NCCL_DEBUG=INFO
timeout -k 10 600 python -m pytest tests/test_sharded_gpu.py -m gpu -q -x -s > gpurun_out/c4/pytest_sharded.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu.ids gpurun_out/c4/pytest_sharded.log | grep -v "NCCL INFO" | tail -30 | cut -c1-300
grep "NCCL INFO" gpurun_out/c4/pytest_sharded.log | head -40 | cut -c1-200
unset NCCL_DEBUG
# the 1 M-hillslope headline through the native route on one device, 8 shards
timeout -k 10 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, time, json
sys.path.insert(0, ".")
import numpy as np
import tiger_hlm_gpu_amd as H
from tiger_hlm_gpu_amd import synth
ns, days = 1_000_000, 30
sp, y0 = synth.model204_params(0, ns), synth.model204_y0(ns)
cells = synth.cell_of_system(0, ns)
pr, t2m, dt, nT = synth.forcing_tables(days=days)
for shards, gather in ((1, "rccl"), (8, "rccl"), (8, "host")):
    t = time.perf_counter()
    got, st = H.api.run_rk45_sharded(H.Model204, y0, 0.0, days * 1440.0, sp=sp, tables=(pr, t2m), dt_hours=dt, cells=cells,
                                     num_shards=shards, num_devices=1, gather=gather)
    st["wall_s"] = time.perf_counter() - t
    st["attempts"] = int(got["n_attempt"].astype(np.int64).sum())
    print(json.dumps({"shards": shards, "gather": gather, **st}))
PY
