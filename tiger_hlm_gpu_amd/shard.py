"""Hillslope sharding across the GPUs of one node + the end-of-run gather.

The path is embarrassingly parallel (no channel routing exists in the reference:
``next_stream`` is loaded but never used), so each rank integrates one contiguous slice
and there is NO collective during integration.  The slice rule is the reference's
rank-0 split, src/main.cpp:275-307 (``baseChunk + (r <= remainder)``), with every rank
owning a GPU (the reference keeps rank 0 idle).  The only exchange is the gather of
final states / flags / step counters to rank 0 (RCCL over xGMI when the backend is
"nccl"; "gloo" in the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_bounds(total, world_size, rank):
    """(start, count) of rank's contiguous slice: first ``total % world_size`` ranks get
    one extra row (main.cpp:275-307)."""
    base, rem = divmod(int(total), int(world_size))
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def gather_to_root(local, total, group=None, root=0):
    """Gather per-rank slices (dim 0 = hillslopes of this rank, in shard_bounds order)
    to ``root``.  Returns the [total, ...] tensor on root, None elsewhere.  Slices may
    differ by one row, so they are padded to the largest slice for the collective."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if local.is_cuda and dist.get_backend(group) == "gloo":
        local = local.cpu()  # gloo rehearsal on a GPU box: the collective runs on host tensors
    counts = [shard_bounds(total, world, r)[1] for r in range(world)]
    cmax = max(counts)
    pad = local
    if local.shape[0] != cmax:
        pad = torch.zeros((cmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    pad = pad.contiguous()
    if rank == root:
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.gather(pad, gather_list=bufs, dst=root, group=group)
        return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
    dist.gather(pad, gather_list=None, dst=root, group=group)
    return None


def result_blob_bytes(count, n_eq=5):
    """Size of a ShardEngine.result_blob of ``count`` hillslopes (y_final | stiff | n_attempt |
    n_accept, the int32 sections padded to an even count)."""
    c8 = -(-int(count) // 2) * 2
    return int(count) * 8 * n_eq + 3 * c8 * 4


# Backends whose process groups implement gather() (torch.distributed: ProcessGroupNCCL -- RCCL on ROCm --
# as grouped send/recv, ProcessGroupGloo, ProcessGroupMPI).  Anything else (ucc, a custom backend) takes
# the all_gather route.  HLM_GATHER_COLLECTIVE=gather|all_gather overrides (tests, bring-up of a backend).
_BACKENDS_WITH_GATHER = ("nccl", "gloo", "mpi")


def gather_collective(group=None):
    """Which collective carries the end-of-run gather on this process group: "gather" or "all_gather".
    Decided from the backend's NAME (and the override) -- a pure function of things every rank agrees on,
    evaluated before any communication.  A collective that then fails raises on the rank that saw it:
    nothing is caught and nothing is retried on another route (a rank that switched collectives alone
    would leave the others waiting in the first one)."""
    import os
    forced = os.environ.get("HLM_GATHER_COLLECTIVE", "")
    if forced:
        if forced not in ("gather", "all_gather"):
            raise ValueError("HLM_GATHER_COLLECTIVE must be 'gather' or 'all_gather', not %r" % forced)
        return forced
    return "gather" if str(dist.get_backend(group)).lower() in _BACKENDS_WITH_GATHER else "all_gather"


def gather_blobs(blob, total, group=None, root=0, n_eq=5, single_rank_collective=False):
    """The end-of-run gather as ONE collective: every rank's result blob (uint8, 52 B per
    hillslope) to ``root``.  Returns the list of per-rank blobs (each cut to its own length) on
    root, None elsewhere.  Shards of unequal size (``total`` not divisible by the world size) are
    padded to the longest blob for the collective.  The collective is gather_collective(group).
    A one-rank group returns the blob itself unless ``single_rank_collective`` asks for the real
    collective (bench.py under a one-rank torchrun: RCCL's communicator and its gather run once)."""
    if not dist.is_initialized():
        return [blob]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world == 1 and not single_rank_collective:
        return [blob]
    if blob.is_cuda and dist.get_backend(group) == "gloo":
        blob = blob.cpu()
    sizes = [result_blob_bytes(shard_bounds(total, world, r)[1], n_eq) for r in range(world)]
    assert blob.numel() == sizes[rank], (blob.numel(), sizes[rank])
    smax = max(sizes)
    pad = blob
    if blob.numel() != smax:
        pad = torch.zeros(smax, dtype=blob.dtype, device=blob.device)
        pad[: blob.numel()] = blob
    if gather_collective(group) == "gather":
        if rank == root:
            bufs = [torch.empty_like(pad) for _ in range(world)]
            dist.gather(pad, gather_list=bufs, dst=root, group=group)
            return [b[:n] for b, n in zip(bufs, sizes)]
        dist.gather(pad, gather_list=None, dst=root, group=group)
        return None
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return [b[:n] for b, n in zip(bufs, sizes)] if rank == root else None
