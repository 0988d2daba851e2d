"""Worker of tests/test_bench_gpu.py::test_two_rank_gather_equals_single_rank (launched with torch.distributed.run, gloo
through host memory, both ranks on the one GPU): every rank aligns ITS shard of one read set, the records and CIGAR arenas
are gathered to rank 0 with panmap_amd.dist.gather_alignments, and rank 0 checks them against its own single-rank
alignment of the whole set: every field of every record and every CIGAR, bit for bit."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    import panmap_amd as pmx
    from panmap_amd import dist as pd
    golden = os.path.join(ROOT, "tests", "golden")
    g = b"".join(l.strip() for l in open(os.path.join(golden, "isolate.ref.fa"), "rb") if not l.startswith(b">"))
    concat, off = pmx.simulate_paired_reads(g, 20000, seed=77, sub_rate=0.01)
    reads = [bytes(concat[off[i]:off[i + 1]]) for i in range(len(off) - 1)]
    rng = np.random.default_rng(1)
    for i in range(0, len(reads), 6):          # indels: multi-operation CIGARs on both shards
        r = bytearray(reads[i])
        p = int(rng.integers(30, 110))
        if i % 12 == 0:
            del r[p:p + 3]
        else:
            r[p:p] = b"ACG"
        reads[i] = bytes(r)
    ctx = pmx.Context(0)
    al = pmx.Aligner(ctx, g, 150)
    lo, hi = pd.shard_bounds(len(reads), world, rank, paired=True)
    rs = pmx.ReadSet(ctx, reads[lo:hi])
    al.align_readset(rs, paired=True, revcomp_mate2=True)
    dev = torch.device("cuda", 0)
    n = hi - lo
    recs = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    al.copy_records_device(recs.data_ptr(), n)
    nw = al.cigar_words()
    cig = torch.empty(max(nw, 1), dtype=torch.int32, device=dev)
    al.copy_cigars_device(cig.data_ptr(), max(nw, 1))
    got = pd.gather_alignments(recs, cig[:nw], 0, via_host=True)
    out = {"rank": rank}
    if rank == 0:
        g_recs, g_arena, n_rec, bases = got
        m = g_recs.numpy().view(pmx.REC_DTYPE).reshape(-1)
        arena = g_arena.numpy().view(np.uint32)
        rs_all = pmx.ReadSet(ctx, reads)
        al.align_readset(rs_all, paired=True, revcomp_mate2=True)
        w, wc = al.fetch()
        ok_fields = all(np.array_equal(m[f], w[f]) for f in ("rs", "re", "qs", "qe", "mapq", "rev", "proper_frag", "mapped", "n_cigar", "flags", "score"))
        # the arenas are laid out in a different order (atomic bump allocation): compare every CIGAR through its offset
        same_cigars = True
        multi = 0
        for i in range(len(w)):
            k = int(w["n_cigar"][i])
            if k == 0:
                continue
            a = arena[int(m["cigar_off"][i]):int(m["cigar_off"][i]) + k]
            b = wc[int(w["cigar_off"][i]):int(w["cigar_off"][i]) + k]
            if not np.array_equal(a, b):
                same_cigars = False
                break
            multi += k > 1
        out.update(n_records=int(len(m)), n_expected=int(len(w)), fields_equal=bool(ok_fields), cigars_equal=bool(same_cigars),
                   multi_op_cigars=int(multi), n_rec=n_rec, bases=bases, rank1_base_nonzero=bool(bases[1] > 0),
                   rank1_has_cigars=bool(np.any(m["n_cigar"][n_rec[0]:] > 0)), flagged=int(np.sum((m["flags"] & 3) != 0)))
        print("RESULT " + json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
