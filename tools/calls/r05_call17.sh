#!/bin/bash
# round 5, call 17: the --gpus line under a launcher: one rank over RCCL (real communicator + gather), two and four gloo ranks
# sharing the card (weak and strong) -- the new `ranks` block included
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
B="--steps 2 --warmup 1 --no-variants --no-dense --no-cpu-baseline"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 $B > $O/bench_nccl_world1.json 2> $O/bench_nccl_world1.err; echo "nccl world 1 rc=$?"
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --ns-per-gpu 300000 $B > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo 2 rc=$?"
timeout -k 10 300 python bench.py --gpus 4 --backend gloo --scaling strong --total-ns 1000001 $B > $O/bench_gloo4_strong.json 2> $O/bench_gloo4.err; echo "gloo 4 strong rc=$?"
python - <<'PY'
import json
for f in ("bench_nccl_world1", "bench_gloo2", "bench_gloo4_strong"):
    d = json.loads(open(f"gpurun_out/r05/{f}.json").read().strip().splitlines()[-1])
    print(f, d["n_gpus"], "%.3g" % d["value"], d["ms_per_step"], json.dumps(d.get("ranks"))[:600], d.get("gather_collective"))
PY
