#!/bin/bash
# round 4, call 46: the multi-rank launch forms on HEAD, as far as one card allows: one rank under torchrun with RCCL
# (communicator + real gather), 2 and 4 gloo ranks sharing the card in the driver's launch form (weak and strong), and
# the native one-process run with 8 shards gathered over RCCL (rk45_solver --gpus 8 --oversubscribe --gather rccl)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c46
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --steps 3 --warmup 1 --no-variants --no-dense --no-cpu-baseline > gpurun_out/c46/nccl1.out 2> gpurun_out/c46/nccl1.err; echo "nccl1 rc=$? stdout lines: $(wc -l < gpurun_out/c46/nccl1.out)"
timeout -k 10 400 python bench.py --gpus 2 --backend gloo --ns-per-gpu 500000 --steps 2 --warmup 1 > gpurun_out/c46/gloo2.out 2> gpurun_out/c46/gloo2.err; echo "gloo2 rc=$? lines: $(wc -l < gpurun_out/c46/gloo2.out)"
timeout -k 10 400 python bench.py --gpus 4 --backend gloo --scaling strong --total-ns 1000001 --steps 2 --warmup 1 > gpurun_out/c46/gloo4.out 2> gpurun_out/c46/gloo4.err; echo "gloo4 rc=$? lines: $(wc -l < gpurun_out/c46/gloo4.out)"
python - <<'PY'
import json
for f in ("nccl1", "gloo2", "gloo4"):
    d = json.loads(open(f"gpurun_out/c46/{f}.out").read().strip().splitlines()[-1])
    print(f, d["n_gpus"], "%.4g" % d["value"], d["ms_per_step"], d["config"]["backend"], d.get("gather_collective"), [(r["hillslopes"], round(r["kernel_ms"], 1), round(r["gather_ms"], 2)) for r in d["per_rank"]])
PY
timeout -k 10 300 python -m pytest tests -m gpu -q -k "sharded" > gpurun_out/c46/sharded.log 2>&1; echo "sharded tests rc=$?"; tail -2 gpurun_out/c46/sharded.log
