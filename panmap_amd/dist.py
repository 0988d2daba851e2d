"""Multi-GPU exchange steps of the hot path (one process per GPU, torch.distributed; backend "nccl" is
RCCL over xGMI on MI355X, "gloo" in the CPU tests).

The only data-path exchange is the seed histogram: every rank reduces its read
shard to unique (hash,count) pairs; an all-gather of variable-length runs (sizes first, then max-padded
buffers -- RCCL has no all-gather-v) followed by a local integer merge gives every rank the identical
full histogram, so node scoring is replicated and deterministic.  Alignment records are gathered to
rank 0 as fixed 32-byte rows.
"""
import numpy as np
import torch
import torch.distributed as dist


def allgather_histograms(hash_t: torch.Tensor, count_t: torch.Tensor, n_local: int):
    """hash_t/count_t: int64 tensors (any device) holding this rank's n_local pairs (buffers may be longer).
    Returns (gathered [world, 2, max_n] int64 tensor, sizes list)."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    dev = hash_t.device
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    sizes[rank] = n_local
    dist.all_reduce(sizes)
    sizes_l = [int(x) for x in sizes.tolist()]
    mx = max(max(sizes_l), 1)
    mine = torch.zeros((2, mx), dtype=torch.int64, device=dev)
    mine[0, :n_local] = hash_t[:n_local]
    mine[1, :n_local] = count_t[:n_local]
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return torch.stack(out), sizes_l


def merge_histograms_host(parts):
    """Reference merge on the host (tests): list of (hash uint64, count int64) -> summed, hash ascending."""
    hs = np.concatenate([np.asarray(p[0], np.uint64) for p in parts]) if parts else np.zeros(0, np.uint64)
    cn = np.concatenate([np.asarray(p[1], np.int64) for p in parts]) if parts else np.zeros(0, np.int64)
    if len(hs) == 0:
        return hs, cn
    order = np.argsort(hs, kind="stable")
    hs, cn = hs[order], cn[order]
    uh, idx = np.unique(hs, return_index=True)
    return uh, np.add.reduceat(cn, idx)


def shard_bounds(n_reads: int, world: int, rank: int, paired: bool = True):
    """contiguous, pair-aligned read shards (mates stay together)"""
    unit = 2 if paired else 1
    n_units = n_reads // unit
    lo = n_units * rank // world
    hi = n_units * (rank + 1) // world
    return lo * unit, hi * unit


def gather_records(recs_u8: torch.Tensor, dst: int = 0):
    """gather fixed-size record rows ([n, 32] uint8) to rank dst; returns the list on dst, None elsewhere"""
    world = dist.get_world_size()
    rank = dist.get_rank()
    sizes = torch.zeros(world, dtype=torch.int64, device=recs_u8.device)
    sizes[rank] = recs_u8.shape[0]
    dist.all_reduce(sizes)
    mx = int(sizes.max().item())
    pad = torch.zeros((mx, recs_u8.shape[1]), dtype=torch.uint8, device=recs_u8.device)
    pad[:recs_u8.shape[0]] = recs_u8
    gl = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, gl, dst=dst)
    if rank != dst:
        return None
    return [g[:int(sizes[r].item())] for r, g in enumerate(gl)]
