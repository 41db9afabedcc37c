#!/bin/bash
# GPU call 1 of round 3
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c1
export HLM_PARITY_DIAG=$GRAFT_REPO_ROOT/gpurun_out/c1/parity_diag
export HLM_PARITY_RECORD=$GRAFT_REPO_ROOT/gpurun_out/c1/parity_exempt_sets.json
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/c1/pytest.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/c1/pytest.rc
tail -5 gpurun_out/c1/pytest.log
unset HLM_PARITY_DIAG HLM_PARITY_RECORD
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 --no-variants --no-dense --no-cpu-baseline > gpurun_out/c1/nccl1.json 2> gpurun_out/c1/nccl1.err; echo "nccl1 rc=$?" | tee gpurun_out/c1/nccl1.rc
tail -c 600 gpurun_out/c1/nccl1.json; tail -5 gpurun_out/c1/nccl1.err
tools/pcsample.sh dense_st stochastic 65536 "--days 1 --dense-stride 1 --dense-tiled 2 --steps 2 --warmup 1 --no-variants --no-dense --no-cpu-baseline"
