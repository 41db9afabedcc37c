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


def _blob_worker(rank, world, port, total, force_fallback, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tiger_hlm_gpu_amd import shard
        if force_fallback:  # a backend without gather: the same bytes must arrive through all_gather
            def no_gather(*a, **k):
                raise RuntimeError("gather is not supported by this backend")
            dist.gather = no_gather
        start, count = shard.shard_bounds(total, world, rank)
        blob = torch.arange(shard.result_blob_bytes(count), dtype=torch.int64).mul_(rank + 3).remainder_(251).to(torch.uint8)
        for _ in range(2):  # the second call takes the remembered route
            got = shard.gather_blobs(blob, total)
        if rank == 0:
            q.put([g.numpy().copy() for g in got])
        else:
            assert got is None
            q.put(None)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("force_fallback", [False, True])
def test_gather_blobs_unequal_shards_and_fallback(force_fallback):
    """The end-of-run gather as one collective over 52-byte-per-hillslope blobs: unequal shards (1001 over 3
    ranks: 334/334/333) are padded for the collective and cut back on the root; a backend that refuses
    gather gets the same result through all_gather."""
    from tiger_hlm_gpu_amd import shard
    world, total = 3, 1001
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_blob_worker, args=(r, world, port, total, force_fallback, q)) for r in range(world)]
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
