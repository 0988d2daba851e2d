"""N>1 path on CPU: world_size-2 gloo processes run the histogram exchange step on oracle-produced shard
histograms; the merged result must equal the single-pass histogram, and both ranks must agree bit for bit."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def pmx_rec_dtype():
    return np.dtype([("rs", "<i4"), ("re", "<i4"), ("qs", "<i4"), ("qe", "<i4"), ("mapq", "u1"), ("rev", "u1"), ("proper_frag", "u1"),
                     ("mapped", "u1"), ("n_cigar", "<u2"), ("flags", "<u2"), ("cigar_off", "<u4"), ("score", "<i4")])


def _worker(rank, world, port, reads, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from panmap_amd import dist as pd
    lo, hi = pd.shard_bounds(len(reads), world, rank, paired=True)
    hs, cn = orc.histogram(reads[lo:hi], 19, 8, 3)
    g, sizes = pd.allgather_histograms(torch.from_numpy(hs.view(np.int64)), torch.from_numpy(cn), len(hs))
    parts = [(g[r, 0, :sizes[r]].numpy().view(np.uint64), g[r, 1, :sizes[r]].numpy()) for r in range(world)]
    mh, mc = pd.merge_histograms_host(parts)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), h=mh, c=mc, lo=lo, hi=hi)
    # alignment results: records whose cigar_off indexes a per-rank arena -> rank 0 gets both, offsets rebased
    n_loc = hi - lo
    rec = np.zeros(n_loc, pmx_rec_dtype())
    rec["rs"] = np.arange(lo, hi)
    rec["n_cigar"] = 1 + (np.arange(n_loc) % 3)
    rec["flags"] = 4
    rec["flags"][::5] = 0                      # no alignment: cigar_off stays 0
    rec["n_cigar"][::5] = 0
    rec["cigar_off"] = np.concatenate([[0], np.cumsum(rec["n_cigar"])[:-1]]) * (rec["flags"] != 0)
    arena = np.concatenate([np.full(int(k), (int(r) << 4), np.uint32) for r, k in zip(rec["rs"], rec["n_cigar"])] + [np.zeros(0, np.uint32)])
    gal = pd.gather_alignments(torch.from_numpy(rec.view(np.uint8).reshape(n_loc, 32).copy()), torch.from_numpy(arena.view(np.int32).copy()), 0)
    if rank == 0:
        g_recs, g_arena, n_rec, bases = gal
        m = g_recs.numpy().view(pmx_rec_dtype()).reshape(-1)
        ar = g_arena.numpy().view(np.uint32)
        assert len(m) == len(reads) and n_rec == [1500, 1502] and bases[0] == 0 and bases[1] > 0
        for i in range(len(m)):                 # every record finds ITS CIGAR words in the merged arena
            k = int(m["n_cigar"][i])
            assert m["rs"][i] == i
            if k:
                assert np.all(ar[int(m["cigar_off"][i]):int(m["cigar_off"][i]) + k] == (i << 4))
            else:
                assert m["cigar_off"][i] == 0
    else:
        assert gal is None
    recs = torch.full((hi - lo, 32), rank, dtype=torch.uint8)
    got = pd.gather_records(recs, 0)
    if rank == 0:
        assert [int(x.shape[0]) for x in got] == [pd.shard_bounds(len(reads), world, r)[1] - pd.shard_bounds(len(reads), world, r)[0] for r in range(world)]
        assert all(int(x[0, 0]) == r for r, x in enumerate(got))
    dist.barrier()
    dist.destroy_process_group()


def test_histogram_exchange_world2(pmx, oracle, tmp_path):
    g = b"".join(l.strip() for l in open(os.path.join(GOLDEN, "isolate.ref.fa"), "rb") if not l.startswith(b">"))
    concat, off = pmx.simulate_paired_reads(g, 1501, seed=9)
    reads = [bytes(concat[off[i]:off[i + 1]]) for i in range(len(off) - 1)]
    mp.spawn(_worker, args=(2, _free_port(), reads, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "rank0.npz")
    b = np.load(tmp_path / "rank1.npz")
    assert np.array_equal(a["h"], b["h"]) and np.array_equal(a["c"], b["c"])
    assert (int(a["lo"]), int(a["hi"]), int(b["lo"]), int(b["hi"])) == (0, 1500, 1500, 3002)   # pair-aligned shards
    hs, cn = oracle.histogram(reads, 19, 8, 3)
    assert np.array_equal(a["h"], hs) and np.array_equal(a["c"], cn)


def test_bench_collective_order_for_any_number_of_pipelines():
    """bench.py's Sequencer: the one order in which the host threads of a rank issue their collectives -- a permutation of the
    2n tickets, every batch's exchange before its gather, and every pipeline's own operations in rising order (a thread runs its
    batches one after the other: it must never wait for a ticket behind one of its own later operations)"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for n in (1, 2, 3, 4, 5, 7, 8, 20, 25):
        for p in (1, 2, 3, 4, 6):
            sq = bench.Sequencer(n, p)
            tick = {("H", i): sq.ticket("H", i) for i in range(n)}
            tick.update({("G", i): sq.ticket("G", i) for i in range(n)})
            assert sorted(tick.values()) == list(range(2 * n)), (n, p)
            assert all(tick[("H", i)] < tick[("G", i)] for i in range(n)), (n, p)
            pp = max(1, min(p, n))
            for q in range(pp):                      # pipeline q runs batches q, q + P, ...
                mine = [t for b in range(q, n, pp) for t in (tick[("H", b)], tick[("G", b)])]
                assert mine == sorted(mine), (n, p, q)
    assert bench.Sequencer(7, 3).ticket("G", 0) == 3 and bench.Sequencer(7, 3).ticket("H", 3) == 4      # H0 H1 H2 G0 H3 G1 ...
    assert [bench.Sequencer(5, 2).ticket("H", i) for i in range(5)] == [0, 1, 3, 5, 7]                  # the two-pipeline order of rounds 3-4
