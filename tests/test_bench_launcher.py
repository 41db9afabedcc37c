"""bench.py's N > 1 launch forms on CPU (no GPU: ``--rehearse`` replaces the integration by a
per-hillslope signature, everything around it -- self-launch of fresh ranks, torchrun env,
reference split rule, the one-collective gather with unequal shards, max-over-ranks timing, the
single JSON line and the exit code -- is the real code).  The reference's equivalent is
``mpirun -np N ./rk45_solver`` (README.md:120, main.cpp:258-309)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _one_json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("extra,shards", [
    (["--gpus", "2", "--ns-per-gpu", "2048", "--no-variants"], [2048, 2048]),
    (["--gpus", "3", "--scaling", "strong", "--total-ns", "10001"], [3334, 3334, 3333]),
])
def test_driver_form_self_launches_fresh_ranks(extra, shards):
    """``python bench.py --gpus N --steps K --warmup W`` (the driver's form, BENCH_r01.json.cmd)."""
    p = subprocess.run([sys.executable, BENCH, "--rehearse", "--backend", "gloo", "--steps", "2", "--warmup", "1"] + extra,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    line = _one_json_line(p.stdout)
    assert line["n_gpus"] == len(shards) and line["shards"] == shards and line["gather_ok"] is True
    assert line["hillslopes_total"] == sum(shards) and line["value"] is None


def test_torchrun_form():
    """The launch the task contract names: python -m torch.distributed.run ... bench.py --gpus N."""
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), BENCH, "--gpus", "2",
                        "--rehearse", "--ns-per-gpu", "1000", "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, cwd=ROOT,
                       env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert p.returncode == 0, p.stderr[-2000:]
    # (gloo's C++ layer prints its own "[Gloo] Rank ..." lines on stdout here; RCCL does not)
    js = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(js) == 1 and json.loads(js[0])["shards"] == [1000, 1000]


def test_failing_ranks_give_a_nonzero_exit_and_no_line():
    """Without --rehearse the ranks need a GPU each: here they fail, and the launcher must say so."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the ranks would run")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--ns-per-gpu", "64", "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, cwd=ROOT)
    assert p.returncode != 0
    assert p.stdout.strip() == ""
