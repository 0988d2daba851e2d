"""Multi-GPU exchange steps of the hot path (one process per GPU, torch.distributed; backend "nccl" is
RCCL over xGMI on MI355X, "gloo" in the CPU tests).  bench.py calls exactly these functions, so the
world-size-2 gloo tests cover the code that ships.

Two exchanges exist:
  * the seed histogram: every rank reduces its read shard to unique (hash,count) pairs; an all-gather of
    variable-length runs (sizes first, then max-padded buffers -- RCCL has no all-gather-v) followed by a local
    integer merge gives every rank the identical full histogram, so node scoring is replicated and deterministic;
  * the alignment results: fixed 32-byte records AND the CIGAR arena they point into are gathered to rank 0
    (padded gather), where each rank's `cigar_off` is rebased onto the concatenated arena -- rank 0 then holds
    everything the BAM writer needs.

`via_host=True` bounces every collective through host memory (gloo with GPU tensors: the functional two-rank
test on a one-GPU box); it is never used for a reported number.
"""
import numpy as np
import torch
import torch.distributed as dist

REC_BYTES = 32
CIGAR_OFF_COL = 6          # int32 column of pmx_aln_record::cigar_off (byte 24)
FLAGS_HAS_ALN = 4


def _coll_tensor(t, via_host):
    return t.cpu() if via_host and t.is_cuda else t


def exchange_sizes(n_local: int, device, via_host=False):
    """every rank's count, as a list (one all-reduce + one host read)"""
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = torch.zeros(world, dtype=torch.int64, device="cpu" if via_host else device)
    sizes[rank] = int(n_local)
    dist.all_reduce(sizes)
    return [int(x) for x in sizes.tolist()]


def exchange_size_pairs(a_local: int, b_local: int, device, via_host=False):
    """two counts per rank in ONE all-reduce and one host read -> (list of a, list of b)"""
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = torch.zeros((2, world), dtype=torch.int64, device="cpu" if via_host else device)
    sizes[0, rank] = int(a_local)
    sizes[1, rank] = int(b_local)
    dist.all_reduce(sizes)
    both = sizes.tolist()
    return [int(x) for x in both[0]], [int(x) for x in both[1]]


def allgather_padded(mine: torch.Tensor, via_host=False):
    """mine: this rank's max-padded buffer (same shape on every rank) -> [world, *mine.shape] on mine's device"""
    world = dist.get_world_size()
    out = torch.empty((world,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
    if via_host:
        co = torch.empty((world,) + tuple(mine.shape), dtype=mine.dtype)
        dist.all_gather_into_tensor(co.view(-1), mine.cpu().contiguous().view(-1))
        out.copy_(co)
    else:
        dist.all_gather_into_tensor(out.view(-1), mine.contiguous().view(-1))
    return out


def allgather_histograms(hash_t: torch.Tensor, count_t: torch.Tensor, n_local: int, via_host=False):
    """hash_t/count_t: int64 tensors (any device) holding this rank's n_local pairs (buffers may be longer).
    Returns (gathered [world, 2, max_n] int64 tensor, sizes list)."""
    sizes_l = exchange_sizes(n_local, hash_t.device, via_host)
    mx = max(max(sizes_l), 1)
    mine = torch.zeros((2, mx), dtype=torch.int64, device=hash_t.device)
    mine[0, :n_local] = hash_t[:n_local]
    mine[1, :n_local] = count_t[:n_local]
    return allgather_padded(mine, via_host), sizes_l


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


def _gather_padded(t: torch.Tensor, n_max: int, dst: int, via_host: bool):
    """rows of t ([n, ...]) padded to n_max rows, gathered to dst -> list of padded tensors on dst, None elsewhere"""
    world, rank = dist.get_world_size(), dist.get_rank()
    if t.shape[0] == n_max and t.is_contiguous():
        pad = t                                    # (weak scaling: every rank holds the same number of records)
    else:
        pad = torch.zeros((n_max,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[:t.shape[0]] = t
    pad = _coll_tensor(pad, via_host)
    gl = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, gl, dst=dst)
    return gl


def gather_records(recs_u8: torch.Tensor, dst: int = 0, via_host=False):
    """gather fixed-size record rows ([n, 32] uint8) to rank dst; returns the list on dst, None elsewhere"""
    sizes = exchange_sizes(recs_u8.shape[0], recs_u8.device, via_host)
    gl = _gather_padded(recs_u8, max(max(sizes), 1), dst, via_host)
    if gl is None:
        return None
    return [g[:sizes[r]] for r, g in enumerate(gl)]


def gather_alignments(recs_u8: torch.Tensor, cigars_u32: torch.Tensor, dst: int = 0, via_host=False):
    """Records ([n, 32] uint8 rows of pmx_aln_record) and the CIGAR arena they index (int32/uint32 words, 1-D) of every
    rank -> on rank dst: (records [N, 32] uint8 in rank order with cigar_off rebased, arena [W] int32 = the ranks'
    arenas back to back, per-rank record counts, per-rank arena bases); None elsewhere."""
    rank = dist.get_rank()
    n_rec, n_cig = exchange_size_pairs(recs_u8.shape[0], cigars_u32.shape[0], recs_u8.device, via_host)
    rl = _gather_padded(recs_u8, max(max(n_rec), 1), dst, via_host)
    cl = _gather_padded(cigars_u32.view(torch.int32), max(max(n_cig), 1), dst, via_host)
    if rank != dst:
        return None
    bases, acc = [], 0
    for w in n_cig:
        bases.append(acc)
        acc += w
    if acc >= 2 ** 32:
        raise OverflowError("merged CIGAR arena exceeds the 32-bit cigar_off of pmx_aln_record")
    out_r = []
    for r, g in enumerate(rl):
        g = g[:n_rec[r]].contiguous()
        if bases[r]:
            g32 = g.view(torch.int32).view(-1, REC_BYTES // 4)
            # (two's-complement add == unsigned add of the 32-bit field; rows without an alignment keep 0)
            has = (g[:, 22].to(torch.int32) & FLAGS_HAS_ALN) != 0          # low byte of `flags`
            g32[:, CIGAR_OFF_COL] += has.to(torch.int32) * ((bases[r] + 2 ** 31) % 2 ** 32 - 2 ** 31)
        out_r.append(g)
    recs = torch.cat(out_r) if out_r else torch.zeros((0, REC_BYTES), dtype=torch.uint8)
    arena = torch.cat([c[:n_cig[r]] for r, c in enumerate(cl)]) if cl else torch.zeros(0, dtype=torch.int32)
    return recs, arena, n_rec, bases
