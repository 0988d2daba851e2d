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
