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
    # exactly one line on stdout: what gloo's C++ layer ("[Gloo] Rank ...") and RCCL (its version banner) print
    # below Python goes to the ranks' stderr (bench._claim_stdout)
    assert _one_json_line(p.stdout)["shards"] == [1000, 1000]


def test_failing_ranks_give_a_nonzero_exit_and_no_line():
    """Without --rehearse the ranks need a GPU each: here they fail, and the launcher must say so."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the ranks would run")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--ns-per-gpu", "64", "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, cwd=ROOT)
    assert p.returncode != 0
    assert p.stdout.strip() == ""


def test_a_rank_that_dies_before_init_ends_the_launch_with_a_nonzero_exit():
    """Rank 1 exits before init_process_group while rank 0 is already waiting in it: the launcher must come
    back non-zero in bounded time (torchrun ends the surviving rank; failing that, the rank's own
    --init-timeout and the launcher's --launch-timeout do), and print no result line."""
    import time
    t0 = time.time()
    p = subprocess.run([sys.executable, BENCH, "--rehearse", "--backend", "gloo", "--gpus", "2", "--ns-per-gpu", "512",
                        "--steps", "1", "--warmup", "0", "--init-timeout", "20", "--launch-timeout", "120"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, cwd=ROOT,
                       env=dict(os.environ, HLM_BENCH_TEST_FAIL_INIT_RANK="1"))
    assert p.returncode != 0
    assert p.stdout.strip() == "", p.stdout
    assert "HLM_BENCH_TEST_FAIL_INIT_RANK" in p.stderr
    assert time.time() - t0 < 150


def test_launch_timeout_kills_the_ranks_and_returns_124():
    """Ranks that never finish (here: both wait in init for a third rendezvous member that does not exist is
    not constructible with torchrun, so rank 0 simply sleeps) are killed as a process group after --launch-timeout."""
    p = subprocess.run([sys.executable, BENCH, "--rehearse", "--backend", "gloo", "--gpus", "2", "--ns-per-gpu", "512",
                        "--steps", "1", "--warmup", "0", "--launch-timeout", "8"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, cwd=ROOT,
                       env=dict(os.environ, HLM_BENCH_TEST_HANG_RANK="0"))
    assert p.returncode == 124, (p.returncode, p.stderr[-1000:])
    assert p.stdout.strip() == ""
    assert "--launch-timeout" in p.stderr
