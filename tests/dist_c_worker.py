"""Worker of tests/test_dist_c_gpu.py: one rank of the multi-GPU exchange THROUGH THE C ABI (pmx_dist_*).  Launched once per
rank as a plain subprocess (PMX_RANK / PMX_WORLD in the environment; the ranks share the one GPU of a box, so the exchange
runs on the library's host-directory test transport, PMX_DIST_HOST_DIR; with PMX_WORLD=1 and no directory it runs on RCCL
itself).  Every rank seeds + aligns ITS shard of one read set; rank 0 checks the merged histogram, the placement and every
gathered record / CIGAR against its own single-rank run over the whole set, bit for bit."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def shard_bounds(n_reads, world, rank):
    n_pairs = n_reads // 2
    return 2 * (n_pairs * rank // world), 2 * (n_pairs * (rank + 1) // world)


def main():
    rank, world = int(os.environ["PMX_RANK"]), int(os.environ["PMX_WORLD"])
    use_torch = not os.environ.get("PMX_WORKER_NO_TORCH")
    if use_torch:
        import torch   # (first, as bench.py does: the process then runs on the HIP / RCCL copies torch ships; without it -- the
                       # panmap command line -- on the system's)
    import panmap_amd as pmx
    golden = os.path.join(ROOT, "tests", "golden")
    pm = pmx.Panman(os.path.join(golden, "sars_20000_twilight_dipper.panman"))
    index = pmx.Index.build(pm)
    g = pm.genome("node_7618")
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
    uid_path = os.environ.get("PMX_DIST_UID_FILE")
    if world > 1 and uid_path and not os.environ.get("PMX_DIST_HOST_DIR"):
        raise SystemExit("this worker shares one GPU: set PMX_DIST_HOST_DIR")
    # rank 0 makes the id, the others read it from a file next to the transport's (as `panmap --gpus N` ships it); under the
    # host transport the id only carries the nonce of the run's file names
    host_dir = os.environ.get("PMX_DIST_HOST_DIR")
    if world > 1 and host_dir:
        import time
        uid_file = os.path.join(host_dir, "uid.tmp")
        if rank == 0:
            uid = pmx.Dist.unique_id()
            with open(uid_file + ".part", "wb") as f:
                f.write(uid)
            os.rename(uid_file + ".part", uid_file)
        else:
            t0 = time.time()
            while not os.path.exists(uid_file):
                assert time.time() - t0 < 300, "rank 0 never published the id"
                time.sleep(0.01)
            uid = open(uid_file, "rb").read()
    else:
        uid = pmx.Dist.unique_id()
    dist = pmx.Dist(ctx, uid, rank, world)
    dist.barrier()
    # PMX_SHARD_MODE (two ranks): how the reads are cut -- the size exchange, the padded histogram planes and the rebase of
    # cigar_off at their edges: "unequal" 70 / 30, "empty" rank 1 holds no read at all, "nocigar" rank 1 holds only reads
    # that map nowhere (records, but not one CIGAR word)
    mode = os.environ.get("PMX_SHARD_MODE", "equal")
    lo, hi = shard_bounds(len(reads), world, rank)
    if mode != "equal":
        assert world == 2
        if mode == "nocigar":
            junk = [bytes(rng.choice(list(b"ACGT"), 150).astype(np.uint8)) for _ in range(6000)]
            reads = reads[:30000] + junk
            cut = 30000
        else:
            cut = {"unequal": 2 * ((len(reads) // 2) * 7 // 10), "empty": len(reads)}[mode]
        lo, hi = (0, cut) if rank == 0 else (cut, len(reads))
    rs = pmx.ReadSet(ctx, reads[lo:hi])
    placer = pmx.Placer(ctx, index)
    placer.reset()
    placer.add_reads(rs)
    dist.merge_histograms(placer)
    res = placer.score(pmx.TraversalParams(), len(reads))
    hh, hc = placer.histogram()
    ref = pm.genome(int(res.best_index[4]))
    al = pmx.Aligner(ctx, ref, 150)
    al.align_readset(rs, paired=True, revcomp_mate2=True)
    n_rec, n_words = dist.gather_alignments(al, 0)
    out = {"rank": rank, "placed": pm.node_id(int(res.best_index[4])), "mode": mode, "shard": [lo, hi]}
    # the one-node form of the same step: every rank downloads ITS part to its place in buffers all ranks map (here: files
    # in the transport's directory, mapped shared) -- no record crosses between the ranks
    sh_rec = sh_cig = None
    if world > 1 and host_dir:
        rb, wb, tot_r, tot_w = dist.plan_alignments(al)
        maps = []
        for name, nbytes in (("shard_records.bin", max(tot_r, 1) * 32), ("shard_cigars.bin", max(tot_w, 1) * 4)):
            fd = os.open(os.path.join(host_dir, name), os.O_RDWR | os.O_CREAT, 0o600)
            os.ftruncate(fd, nbytes)       # (both ranks: same size)
            maps.append(np.memmap(os.path.join(host_dir, name), dtype=np.uint8, mode="r+", shape=(nbytes,)))
            os.close(fd)
        dist.fetch_shard_async(al, maps[0].ctypes.data, maps[1].ctypes.data)
        ctx.synchronize()
        maps[0].flush(); maps[1].flush()
        dist.barrier()
        out["plan"] = [rb, wb, tot_r, tot_w]
        sh_rec = np.frombuffer(maps[0], dtype=pmx.REC_DTYPE)[:tot_r]
        sh_cig = np.frombuffer(maps[1], dtype=np.uint32)[:tot_w]
    if rank == 0:
        m, arena = dist.fetch_gathered(n_rec, n_words)
        per_rank, words_per_rank = dist.rank_counts()
        # the same through the asynchronous download (a second stream of the caller's)
        async_same = None
        if use_torch:
            side = torch.cuda.Stream()
            h_r = torch.zeros((max(n_rec, 1), 32), dtype=torch.uint8).pin_memory()
            h_c = torch.zeros(max(n_words, 1), dtype=torch.int32).pin_memory()
            dist.fetch_gathered_async(h_r.data_ptr(), n_rec, h_c.data_ptr(), n_words, side.cuda_stream)
            side.synchronize()
            async_same = bool(np.array_equal(h_r.numpy().view(pmx.REC_DTYPE).reshape(-1)[:n_rec], m) and np.array_equal(h_c.numpy().view(np.uint32)[:n_words], arena))
        # single-rank run over the whole set
        rs_all = pmx.ReadSet(ctx, reads)
        p1 = pmx.Placer(ctx, index)
        p1.reset()
        p1.add_reads(rs_all)
        res1 = p1.score(pmx.TraversalParams(), len(reads))
        wh, wcnt = p1.histogram()
        al.align_readset(rs_all, paired=True, revcomp_mate2=True)
        w, wc = al.fetch()
        ok_fields = all(np.array_equal(m[f], w[f]) for f in ("rs", "re", "qs", "qe", "mapq", "rev", "proper_frag", "mapped", "n_cigar", "flags", "score"))
        same_cigars, multi = True, 0
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
        if sh_rec is not None:   # the sharded download holds the very bytes of the gathered set
            out["sharded_equals_gathered"] = bool(len(sh_rec) == len(m) and sh_rec.tobytes() == m.tobytes() and np.array_equal(sh_cig, arena))
        out.update(n_records=int(len(m)), n_expected=int(len(w)), fields_equal=bool(ok_fields), cigars_equal=bool(same_cigars),
                   multi_op_cigars=int(multi), per_rank=[int(x) for x in per_rank], words_per_rank=[int(x) for x in words_per_rank],
                   hist_equal=bool(np.array_equal(hh, wh) and np.array_equal(hc, wcnt)), async_same=async_same,
                   scores_equal=bool(res.best_score == res1.best_score and res.best_index == res1.best_index),
                   flagged=int(np.sum((m["flags"] & 3) != 0)))
    else:
        out.update(n_records=n_rec, n_words=n_words)
    # --dedup over the ranks: a read set whose second half repeats its first half (every duplicate pair straddles the two
    # shards) plus duplicates inside a shard; the merged histogram must equal the single-rank --dedup histogram
    dd = reads[:6000] + reads[100:160] + reads[:6000] + reads[200:260]
    lo, hi = shard_bounds(len(dd), world, rank)
    rs_d = pmx.ReadSet(ctx, dd[lo:hi])
    kept = dist.dedup_reads(placer, rs_d)
    placer.reset()
    placer.add_reads(rs_d, pmx.TraversalParams(dedupReads=True))
    dist.merge_histograms(placer)
    dh, dc = placer.histogram()
    out["dedup_kept"] = kept
    if rank == 0:
        rs_all = pmx.ReadSet(ctx, dd)
        p2 = pmx.Placer(ctx, index)
        p2.reset()
        p2.add_reads(rs_all, pmx.TraversalParams(dedupReads=True))
        wh, wc2 = p2.histogram()
        p2.reset()
        p2.add_reads(rs_all, pmx.TraversalParams())
        _, wc_nodedup = p2.histogram()
        out["dedup_hist_equal"] = bool(np.array_equal(dh, wh) and np.array_equal(dc, wc2))
        out["dedup_matters"] = bool(int(wc_nodedup.sum()) > int(wc2.sum()))
        out["distinct_reads"] = len(set(dd))
    dist.barrier()
    dist.close()
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
