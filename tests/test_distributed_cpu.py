"""The N > 1 path on CPU: world_size-2 gloo processes shard the hillslopes with the
reference's split rule and gather finals / flags / counters to rank 0.  The integration
itself needs a GPU, so each rank fills its shard with a per-hillslope signature instead;
the GPU tests prove separately that shards are bit-identical to the whole
(test_shard_invariance_and_idempotence_bitwise)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tiger_hlm_gpu_amd import synth
        from tiger_hlm_gpu_amd.shard import gather_to_root, shard_bounds
        start, count = shard_bounds(total, world, rank)
        sp = synth.model204_params(start, count)
        # per-hillslope signature computed from the shard's OWN generated inputs
        y_final = torch.from_numpy(np.stack([sp["Hu"], sp["n_mann"], sp["slope"], sp["L"], sp["A_h"]], axis=1))
        stiff = torch.from_numpy((sp["stream"] % 3).astype(np.int32))
        natt = torch.from_numpy((sp["stream"] % 1000).astype(np.int32))
        gy = gather_to_root(y_final, total)
        gs = gather_to_root(stiff, total)
        gn = gather_to_root(natt, total)
        if rank == 0:
            q.put((gy.numpy(), gs.numpy(), gn.numpy()))
        else:
            assert gy is None and gs is None and gn is None
            q.put(None)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [1001, 64])
def test_gloo_world2_shard_and_gather(total):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res = [g for g in got if g is not None]
    assert len(res) == 1
    gy, gs, gn = res[0]
    from tiger_hlm_gpu_amd import synth
    sp = synth.model204_params(0, total)
    assert gy.shape == (total, 5) and gs.shape == (total,) and gn.shape == (total,)
    assert np.array_equal(gy, np.stack([sp["Hu"], sp["n_mann"], sp["slope"], sp["L"], sp["A_h"]], axis=1))
    assert np.array_equal(gs, (sp["stream"] % 3).astype(np.int32))
    assert np.array_equal(gn, (sp["stream"] % 1000).astype(np.int32))


def test_gather_is_identity_without_process_group():
    from tiger_hlm_gpu_amd.shard import gather_to_root
    x = torch.arange(10.0)
    assert gather_to_root(x, 10) is x


def _blob_worker(rank, world, port, total, route, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if route != "default":
        os.environ["HLM_GATHER_COLLECTIVE"] = route
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tiger_hlm_gpu_amd import shard
        assert shard.gather_collective() == ("gather" if route == "default" else route)
        used = []
        real_gather, real_all_gather = dist.gather, dist.all_gather
        dist.gather = lambda *a, **k: (used.append("gather"), real_gather(*a, **k))[1]
        dist.all_gather = lambda *a, **k: (used.append("all_gather"), real_all_gather(*a, **k))[1]
        start, count = shard.shard_bounds(total, world, rank)
        blob = torch.arange(shard.result_blob_bytes(count), dtype=torch.int64).mul_(rank + 3).remainder_(251).to(torch.uint8)
        for _ in range(2):
            got = shard.gather_blobs(blob, total)
        assert used == ["gather" if route == "default" else route] * 2, used
        if rank == 0:
            q.put([g.numpy().copy() for g in got])
        else:
            assert got is None
            q.put(None)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("route", ["default", "gather", "all_gather"])
def test_gather_blobs_unequal_shards_on_both_collectives(route):
    """The end-of-run gather as one collective over 52-byte-per-hillslope blobs: unequal shards (1001 over 3
    ranks: 334/334/333) are padded for the collective and cut back on the root.  The collective is chosen
    from the backend's name (or HLM_GATHER_COLLECTIVE) before any communication; both routes deliver the
    same bytes."""
    from tiger_hlm_gpu_amd import shard
    world, total = 3, 1001
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_blob_worker, args=(r, world, port, total, route, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res = [g for g in got if g is not None]
    assert len(res) == 1 and len(res[0]) == world
    for r in range(world):
        count = shard.shard_bounds(total, world, r)[1]
        want = (np.arange(shard.result_blob_bytes(count), dtype=np.int64) * (r + 3) % 251).astype(np.uint8)
        assert np.array_equal(res[0][r], want), r


def _failing_collective_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tiger_hlm_gpu_amd import shard
        calls = []

        def broken_gather(*a, **k):
            calls.append("gather")
            raise RuntimeError("simulated collective failure")
        dist.gather = broken_gather
        dist.all_gather = lambda *a, **k: calls.append("all_gather")
        blob = torch.zeros(shard.result_blob_bytes(shard.shard_bounds(10, world, rank)[1]), dtype=torch.uint8)
        try:
            shard.gather_blobs(blob, 10)
            q.put("swallowed")
        except RuntimeError as e:
            q.put(("raised", str(e), calls))
    finally:
        dist.destroy_process_group()


def test_a_failing_collective_raises_and_is_never_rerouted():
    """A RuntimeError out of the collective (timeout, RCCL async error, OOM on the root) must reach the
    caller: catching it and switching this rank to another collective would leave the other ranks waiting
    in the first one (round-2 advisor finding)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_collective_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for g in got:
        assert g != "swallowed" and g[0] == "raised" and "simulated" in g[1] and g[2] == ["gather"], g


def _single_rank_worker(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from tiger_hlm_gpu_amd import shard
        used = []
        real = dist.gather
        dist.gather = lambda *a, **k: (used.append("gather"), real(*a, **k))[1]
        blob = torch.arange(shard.result_blob_bytes(7), dtype=torch.int64).remainder_(251).to(torch.uint8)
        a = shard.gather_blobs(blob, 7)
        b = shard.gather_blobs(blob, 7, single_rank_collective=True)
        q.put((a[0] is blob, used, bool(torch.equal(b[0], blob)), b[0] is blob))
    finally:
        dist.destroy_process_group()


def test_single_rank_group_can_run_the_real_collective():
    """bench.py under a ONE-rank torchrun runs the gather for real (communicator creation + the collective
    execute once before the first 8-GPU contact); everywhere else a one-rank group is the identity."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_single_rank_worker, args=(_free_port(), q))
    p.start()
    identity, used, equal, same_object = q.get(timeout=180)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert identity and used == ["gather"] and equal and not same_object
