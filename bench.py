#!/usr/bin/env python3
"""Benchmark of the panmap hot path on MI355X: reads placed + aligned per second.

One "step" = one pass of the hot path over one batch of synthetic reads resident in HBM (ASCII, as the
boundary hands them over): 2-bit pack -> syncmer/k-min-mer seeding + seed histogram -> [N>1: RCCL
all-gather + merge of the per-rank histograms] -> node scoring down the PanMAN tree -> materialise the
placed genome + build its minimizer index -> map + align every read pair -> [N>1: RCCL gather of the
alignment records AND the CIGAR arena to rank 0, panmap_amd/dist.py].

Workload (BASELINE.json configs[1]): 1M x 150 bp synthetic paired reads vs the 20,000-genome SARS-CoV-2
PanMAN on one GPU.  --gpus N: every rank takes its own shard, the seed index is replicated per GPU.
  --scaling weak   (default) 1M reads per rank (8 ranks = 8M reads)
  --scaling strong --total-reads 10000000   BASELINE configs[2] literally: 10M reads split over the ranks
  --read-len >= 500: single-end long reads (configs[3], `--reads-per-gpu 100000 --read-len 10000`)

The ONE JSON line (rank 0) carries, next to the contract keys:
  value_host_to_host   the SURVEY 8d metric: pinned host ASCII -> H2D (chunked, overlapped with packing + seeding on
                       a side stream) -> the same step -> D2H of records + CIGAR arena into pinned host memory
  roofline             dominant kernel vs the HBM peak (algorithmic bytes / its HIP-event duration)
  dp                   share of pairs that ran a ksw2 DP, DP cells per step, GCUPS
  real_reads           the repository's real example reads x8 through the same step (20 % of the pairs need a DP)
  cpu_baseline         the CPU path on ALL host cores, timed around bare C calls (no Python in the timed spans)

Launched for N>1 as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def cpu_baseline(concat, off, index_arrays, placed_genome, sample_reads, threads, paired):
    """The CPU path timed on this box's host cores on a bounded sample of the SAME workload.
    place leg = oracle restatement (oracle/oracle_place.c; the reference's placement.cpp cannot be built without
    panman/TBB/abseil), seeded on `threads` pthreads in ONE C call, then finalize + node scoring + best/tie rule;
    align leg = the reference's own aligner compiled from its sources (oracle/_ref: src/mm_align.c + vendored
    minimap2), timed around the bare align_reads_direct call with `threads` worker threads.  All ctypes marshalling
    and unpacking happens outside the timed spans."""
    from oracle import oracle as orc
    import panmap_amd as pmx
    n = min(sample_reads, len(off) - 1)
    if paired:
        n &= ~1
    sub_off = np.ascontiguousarray(off[:n + 1])
    sub_concat = np.ascontiguousarray(concat[:int(sub_off[-1])])
    orc.olib()
    orc.rlib()
    # ---- place leg
    t0 = time.perf_counter()
    uh, uc = orc.histogram_flat_mt(sub_concat, sub_off, 19, 8, 3, threads)
    t_seed = time.perf_counter() - t0
    t0 = time.perf_counter()
    kh, kl, st = orc.finalize_reads(uh, uc, 19)
    sc, _, _, _ = orc.score_nodes(index_arrays["parent"], index_arrays["offsets"], index_arrays["hash"], index_arrays["parent_count"],
                                  index_arrays["child_count"], kh, kl, st)
    orc.best_ties(index_arrays["parent"], sc)
    t_score = time.perf_counter() - t0
    # ---- align leg (R2 reverse-complemented as readFastqPaired does; marshalling not timed)
    reads = [bytes(sub_concat[sub_off[i]:sub_off[i + 1]]) for i in range(n)]
    if paired:
        reads = [r if i % 2 == 0 else pmx.reverse_complement(r) for i, r in enumerate(reads)]
    prep = orc.prepare_align_call(reads, paired)
    t0 = time.perf_counter()
    orc.run_align_call(orc.rlib().align_reads_direct, placed_genome, prep, threads)
    t_align = time.perf_counter() - t0
    res = orc.unpack_align_call(prep)
    mapped = sum(r["mapped"] for r in res)
    total = t_seed + t_score + t_align
    return dict(value=n / total, unit="reads/s", cores=threads, kind="reference",
                legs={"seed_reads_per_s": n / t_seed, "score_s_per_sample": t_score, "align_reads_per_s": n / t_align},
                sample="%d of the workload's reads, %d threads (= all host cores): place leg = oracle port of seeding (one "
                       "pthread-chunked C call, %.2fs) + node scoring (%.2fs, once per sample, single thread as a fixed cost); "
                       "align leg = the reference's minimap2 via src/mm_align.c align_reads_direct, bare C call %.2fs; "
                       "%d/%d %s mapped" % (n, threads, t_seed, t_score, t_align, mapped, len(res), "pairs" if paired else "reads"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (defaults: the first steps after start-up run ~3 % slower -- clocks, first touches -- and 20 steps of 8.5 ms cost nothing)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads-per-gpu", type=int, default=1000000)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--total-reads", type=int, default=10000000, help="--scaling strong: reads of the whole job")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--cpu-sample", type=int, default=0, help="reads of the CPU baseline sample (0 = sized for ~10-20 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-real-reads", action="store_true")
    ap.add_argument("--no-host-to-host", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="skip the two-batches-in-flight figure")
    ap.add_argument("--h2d-chunks", type=int, default=4)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # PMX_BENCH_TEST_BACKEND=gloo: functional check of the multi-rank logic on a box with fewer GPUs than ranks
    # (ranks share devices, collectives go through host memory).  Never used for a reported number.
    test_gloo = os.environ.get("PMX_BENCH_TEST_BACKEND") == "gloo"
    if test_gloo:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    # PMX_BENCH_FORCE_DIST=1: run the N>1 code path (process group, histogram exchange, record / CIGAR gather, barriers)
    # in a one-rank RCCL group -- the functional check of the nccl backend that a one-GPU box allows (tests/test_bench_gpu.py)
    dist_on = world > 1 or os.environ.get("PMX_BENCH_FORCE_DIST") == "1"
    saved_stdout_fd = None
    if dist_on:
        # stdout carries the ONE JSON line and nothing else: RCCL prints a version banner on stdout through C stdio, so
        # descriptor 1 points at stderr until the line is printed
        sys.stdout.flush()
        saved_stdout_fd = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 1000))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if test_gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import panmap_amd as pmx
    from panmap_amd import dist as pdist
    golden = os.path.join(ROOT, "tests", "golden")
    pm = pmx.Panman(os.path.join(golden, "sars_20000_twilight_dipper.panman"))
    index = pmx.Index.build(pm, k=19, s=8, t=0, l=3, open_syncmer=False, flank_mask=250)
    ctx = pmx.Context(local_rank)
    placer = pmx.Placer(ctx, index)
    ctx_stream = torch.cuda.ExternalStream(ctx.stream, device=dev) if ctx.stream else torch.cuda.current_stream(dev)

    # ---------------------------------------------------------------------------------------------- workload
    # Source genome: node_7618 of the tree (the node the repository's example sample places on; SURVEY 8d asks for
    # the splitmix64(42) mod 20000-th leaf and a splitmix64 stream -- this generator uses numpy's PCG64 and a fixed
    # node instead, see panmap_amd/synth.py); every rank draws its own shard with its own seed.
    src = pm.genome("node_7618")
    long_reads = args.read_len >= 500
    paired = not long_reads
    if args.scaling == "strong":
        lo, hi = pdist.shard_bounds(args.total_reads, world, rank, paired=paired)
        my_reads = hi - lo
    else:
        my_reads = args.reads_per_gpu
    if long_reads:
        lst = pmx.simulate_long_reads(src, my_reads, read_len=args.read_len, seed=43 + rank)
        cb, off = pmx.concat_reads(lst)
        concat = np.frombuffer(cb, np.uint8).copy()
        del lst, cb
    else:
        concat, off = pmx.simulate_paired_reads(src, my_reads // 2, read_len=args.read_len, seed=42 + rank)
    n_reads = len(off) - 1
    max_len = int(np.max(np.diff(off))) if n_reads else 0
    mean_len = int(off[-1] // max(n_reads, 1))
    total_reads = n_reads * world if args.scaling == "weak" else args.total_reads // (2 if paired else 1) * (2 if paired else 1)

    params = pmx.TraversalParams()
    state = {}
    host_times = {} if os.environ.get("PMX_BENCH_HOST_TIMES") else None   # diagnostic: serialised per-phase wall times

    def tick(name, t_prev):
        if host_times is None:
            return t_prev
        ctx.synchronize()
        t = time.perf_counter()
        host_times[name] = host_times.get(name, 0.0) + (t - t_prev) * 1e3
        return t

    def place_and_align(read_sets, n_total_reads, mean_read_len, is_paired, revcomp_mate2, all_rs=None):
        """the hot path on read sets whose reads are already packed and seeded into `placer`; `all_rs` = the read set the
        aligner runs on (the only one, or a wrapper of the whole buffer when the upload was chunked)"""
        tk = time.perf_counter()
        if dist_on:
            # exchange step: all-gather the per-rank (hash,count) histograms, merge the other ranks' parts
            n_loc = placer.histogram_entries()       # (no sort here: the one sorted histogram is made after the merge)
            h_sizes = np.asarray(pdist.exchange_sizes(n_loc, dev, via_host=test_gloo), np.int64)   # one host round trip
            mx = max(int(h_sizes.max()), 1)
            mine = torch.empty((2, mx), dtype=torch.int64, device=dev)
            placer.export_device_unsorted(mine[0].data_ptr(), mine[1].data_ptr(), mx)
            ctx.synchronize()                          # the export ran on the context's stream, the collective runs on torch's
            allh = pdist.allgather_padded(mine, via_host=test_gloo)
            torch.cuda.synchronize()
            # rank p's run sits 2*mx elements after rank p-1's in both the hash and the count plane
            placer.merge_device_parts(allh[0, 0].data_ptr(), allh[0, 1].data_ptr(), 2 * mx, h_sizes, rank)
            tk = tick("exchange", tk)
        res = placer.score(params, n_total_reads)
        tk = tick("score", tk)
        node = res.best_index[4]                      # bestLogContainmentNodeId (src/main.cpp:1771)
        ref = pm.genome(int(node))                    # getStringFromReference, every step (nothing cached)
        tk = tick("genome", tk)
        if "aligner" not in state:
            state["aligner"] = pmx.Aligner(ctx, ref, mean_read_len)
        else:
            state["aligner"].set_reference(ref, mean_read_len)   # mm_idx_str of the placed genome, every step
        aligner = state["aligner"]
        tk = tick("ref_index", tk)
        aligner.align_readset(all_rs if all_rs is not None else read_sets[0], paired=is_paired, revcomp_mate2=revcomp_mate2)
        tk = tick("align", tk)
        state["res"], state["ref"] = res, ref
        return aligner

    def gather_results(aligner, n_my_reads):
        """N>1: fixed-size records + the CIGAR arena of every rank to rank 0, cigar_off rebased (dist.gather_alignments)"""
        recs = torch.empty((n_my_reads, 32), dtype=torch.uint8, device=dev)
        aligner.copy_records_device(recs.data_ptr(), n_my_reads)
        nw = aligner.cigar_words()
        cig = torch.empty(max(nw, 1), dtype=torch.int32, device=dev)
        aligner.copy_cigars_device(cig.data_ptr(), max(nw, 1))
        state["gathered"] = pdist.gather_alignments(recs, cig[:nw], 0, via_host=test_gloo)

    # inputs resident in HBM before the timed region (ASCII + offsets, as torch tensors)
    d_concat = torch.from_numpy(concat).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    torch.cuda.synchronize()
    rs = pmx.ReadSet.wrap_device(ctx, d_concat.data_ptr(), d_off.data_ptr(), n_reads, int(concat.size), max_len, keepalive=(d_concat, d_off))

    def step():
        tk = tick("", time.perf_counter()) if host_times is not None else 0.0
        rs.pack()
        tk = tick("pack", tk)
        placer.reset()
        placer.add_reads(rs, params)
        tk = tick("seed", tk)
        aligner = place_and_align([rs], total_reads, mean_len, paired, paired)
        if dist_on:
            tk = tick("", time.perf_counter()) if host_times is not None else 0.0
            gather_results(aligner, n_reads)
            tk = tick("gather", tk)
        ctx.synchronize()

    def sync_all():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if not dist_on:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device="cpu" if test_gloo else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    kernel_ms = {"align": [], "align_dom": [], "seed": [], "score": []}
    dp_stats = []
    for _ in range(args.steps):
        step()
        for k in kernel_ms:
            kernel_ms[k].append(ctx.kernel_ms(k))
        dp_stats.append(state["aligner"].stats())
    sync_all()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    if host_times is not None and rank == 0:
        n_st = args.steps + args.warmup
        print("[bench host times, ms/step, serialised] " + " ".join(f"{k}={v / n_st:.3f}" for k, v in host_times.items() if k), file=sys.stderr)

    # sanity on the last step's output (not timed)
    recs, cig = state["aligner"].fetch()
    res = state["res"]
    mapped_frac = float(np.mean(recs["mapped"]))
    flagged = int(np.sum((recs["flags"] & 3) != 0))
    placed_id = pm.node_id(int(res.best_index[4]))
    mean_cigar = float(np.mean(recs["n_cigar"]))
    gather_ok = None
    if dist_on and rank == 0 and state.get("gathered") is not None:
        g_recs, g_arena, g_n, g_bases = state["gathered"]
        mine_r = g_recs[:n_reads].cpu().numpy().view(pmx.REC_DTYPE).reshape(-1)
        gather_ok = bool(g_recs.shape[0] == sum(g_n) and np.array_equal(mine_r["rs"], recs["rs"]) and
                         np.array_equal(g_arena[:len(cig)].cpu().numpy().view(np.uint32), cig))

    # ------------------------------------------------------------------------------- two batches in flight (one GPU only)
    # Not `value`: the same step on two independent pipelines (own context = own stream, own placer / aligner / packed read
    # set, own host thread), each running half of the steps back to back.  What a streaming job over many batches gets: the
    # latency-bound tail of one batch's align stage and the host round trips between its stages overlap with the other
    # batch's kernels.  Results per batch are the ones of `value` (same code path, checked below).
    overlapped = None
    if world == 1 and not args.no_overlap and not long_reads:
        try:
            import threading
            ctx_b = pmx.Context(local_rank)
            placer_b = pmx.Placer(ctx_b, index)
            rs_b = pmx.ReadSet.wrap_device(ctx_b, d_concat.data_ptr(), d_off.data_ptr(), n_reads, int(concat.size), max_len, keepalive=(d_concat, d_off))
            pipes = [dict(ctx=ctx, placer=placer, rs=rs, aligner=state["aligner"]), dict(ctx=ctx_b, placer=placer_b, rs=rs_b, aligner=None)]

            def pipe_step(pp):
                pp["rs"].pack()
                pp["placer"].reset()
                pp["placer"].add_reads(pp["rs"], params)
                res_p = pp["placer"].score(params, total_reads)
                ref_p = pm.genome(int(res_p.best_index[4]))
                if pp["aligner"] is None:
                    pp["aligner"] = pmx.Aligner(pp["ctx"], ref_p, mean_len)
                else:
                    pp["aligner"].set_reference(ref_p, mean_len)
                pp["aligner"].align_readset(pp["rs"], paired=paired, revcomp_mate2=paired)
                pp["node"] = int(res_p.best_index[4])

            def pipe_run(pp, n):
                for _ in range(n):
                    pipe_step(pp)
                pp["ctx"].synchronize()
            for pp in pipes:
                pipe_run(pp, 1)       # warm-up (allocations of the second pipeline)
            n_each = max(1, args.steps // 2)
            sync_all()
            t_o = time.perf_counter()
            th = [threading.Thread(target=pipe_run, args=(pp, n_each)) for pp in pipes]
            for t in th:
                t.start()
            for t in th:
                t.join()
            sync_all()
            el_o = time.perf_counter() - t_o
            rb, cb = pipes[1]["aligner"].fetch()
            same = bool(pipes[1]["node"] == int(res.best_index[4]) and np.array_equal(rb["rs"], recs["rs"]) and np.array_equal(rb["mapq"], recs["mapq"])
                        and np.array_equal(rb["n_cigar"], recs["n_cigar"]))
            overlapped = dict(value=n_reads * 2 * n_each / el_o, unit="reads/s", ms_per_step=el_o / (2 * n_each) * 1e3, steps=2 * n_each, batches_in_flight=2,
                              equals_serial_run=same,
                              note="two pipelines (context/stream + host thread each) alternate batches; every batch goes through the same step as `value`")
        except Exception as e:     # noqa: BLE001  (an optional figure must not take the bench line down)
            overlapped = dict(error=str(e)[:200])

    # ------------------------------------------------------------------------------- host -> host (SURVEY 8d metric)
    h2h = None
    if not args.no_host_to_host:
        h_concat = torch.from_numpy(concat).pin_memory()
        h_off = torch.from_numpy(off).pin_memory()
        d_concat2 = torch.empty_like(d_concat)
        d_off2 = torch.empty_like(d_off)
        copy_stream = torch.cuda.Stream(device=dev)
        n_chunks = max(1, min(args.h2d_chunks, n_reads // 2 or 1))
        unit = 2 if paired else 1
        bounds = [(n_reads // unit) * c // n_chunks * unit for c in range(n_chunks + 1)]
        out_recs = torch.empty((n_reads, 32), dtype=torch.uint8).pin_memory()
        out_cig = torch.empty(max(n_reads * 16, int(concat.size) // 4, 4096), dtype=torch.int32).pin_memory()
        d_recs = torch.empty((n_reads, 32), dtype=torch.uint8, device=dev)
        d_cig = torch.empty(out_cig.numel(), dtype=torch.int32, device=dev)

        pool = {}   # read set objects reused from step to step (pmx_readset_rewrap_device: no hipMalloc, offsets handled on the device)

        def read_set(key, d_off_ptr, n):
            if key in pool:
                pool[key].rewrap_device(d_concat2.data_ptr(), d_off_ptr, n, int(concat.size), max_len)
            else:
                pool[key] = pmx.ReadSet.wrap_device(ctx, d_concat2.data_ptr(), d_off_ptr, n, int(concat.size), max_len)
            return pool[key]

        def step_h2h():
            # offsets first (8 B per read), then the bases chunk by chunk on the copy stream; chunk c is packed and seeded
            # on the library's stream as soon as its copy has landed, while chunk c+1 is still in flight
            evs = []
            with torch.cuda.stream(copy_stream):
                d_off2.copy_(h_off, non_blocking=True)
                ev_off = torch.cuda.Event()
                ev_off.record(copy_stream)
                for c in range(n_chunks):
                    b0, b1 = int(off[bounds[c]]), int(off[bounds[c + 1]])
                    d_concat2[b0:b1].copy_(h_concat[b0:b1], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(copy_stream)
                    evs.append(ev)
            placer.reset()
            ctx_stream.wait_event(ev_off)
            parts = []
            for c in range(n_chunks):
                r0, r1 = bounds[c], bounds[c + 1]
                if r1 <= r0:
                    continue
                part = read_set(c, d_off2.data_ptr() + 8 * r0, r1 - r0)   # (needs the offsets only)
                ctx_stream.wait_event(evs[c])
                part.pack()
                placer.add_reads(part, params)
                parts.append(part)
            whole = read_set("whole", d_off2.data_ptr(), n_reads)
            whole.pack()
            aligner = place_and_align(parts, total_reads, mean_len, paired, paired, all_rs=whole)
            nw = aligner.cigar_words()
            if dist_on:
                gather_results(aligner, n_reads)
            # records + CIGAR arena into pinned host memory
            aligner.copy_records_device(d_recs.data_ptr(), n_reads)
            out_recs.copy_(d_recs, non_blocking=True)
            if nw > out_cig.numel():
                raise RuntimeError("pinned CIGAR buffer too small")
            if nw > d_cig.numel():
                raise RuntimeError("device CIGAR buffer too small")
            aligner.copy_cigars_device(d_cig.data_ptr(), max(nw, 1))
            out_cig[:nw].copy_(d_cig[:nw], non_blocking=True)
            torch.cuda.synchronize()
            return nw

        for _ in range(max(1, args.warmup)):
            step_h2h()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            nw_last = step_h2h()
        sync_all()
        el2 = max_over_ranks(time.perf_counter() - t0)
        same = bool(np.array_equal(out_recs.numpy().view(pmx.REC_DTYPE).reshape(-1)["rs"], recs["rs"]) and nw_last == len(cig))
        for p_ in pool.values():
            p_.close()
        h2h = dict(value=total_reads * args.steps / el2, ms_per_step=el2 / args.steps * 1e3, h2d_chunks=n_chunks,
                   equals_device_resident_run=same,
                   note="pinned host ASCII + offsets -> H2D in %d chunks on a copy stream (chunk c packed + seeded while chunk c+1 is in flight) "
                        "-> the same step -> D2H of the 32 B records and the CIGAR arena into pinned host memory" % n_chunks)

    # ------------------------------------------------------------------------------- real reads (one GPU only)
    real = None
    if world == 1 and not args.no_real_reads and not long_reads:
        seqs, _, _ = pmx.read_fastq_paired(os.path.join(golden, "isolate_R1.fastq.gz"), os.path.join(golden, "isolate_R2.fastq.gz"))
        rr = seqs * 8                                    # R2 already reverse-complemented by read_fastq_paired (align orientation)
        r_cb, r_off = pmx.concat_reads(rr)
        r_concat = np.frombuffer(r_cb, np.uint8).copy()
        r_mean = int(r_off[-1] // len(rr))
        rd_concat = torch.from_numpy(r_concat).to(dev)
        rd_off = torch.from_numpy(r_off).to(dev)
        rrs = pmx.ReadSet.wrap_device(ctx, rd_concat.data_ptr(), rd_off.data_ptr(), len(rr), int(r_concat.size), int(np.max(np.diff(r_off))),
                                      keepalive=(rd_concat, rd_off))

        def step_real():
            rrs.pack()
            placer.reset()
            placer.add_reads(rrs, params)     # (canonical seeds: the orientation of R2 does not matter to the place stage)
            place_and_align([rrs], len(rr), r_mean, True, False)
            ctx.synchronize()
        step_real()
        ctx.synchronize()
        t0 = time.perf_counter()
        n_real_steps = 3
        al_ms = []
        for _ in range(n_real_steps):
            step_real()
            al_ms.append(ctx.kernel_ms("align"))
        ctx.synchronize()
        el3 = time.perf_counter() - t0
        st = state["aligner"].stats()
        rrecs, _ = state["aligner"].fetch()
        real = dict(value=len(rr) * n_real_steps / el3, unit="reads/s", reads=len(rr), ms_per_step=el3 / n_real_steps * 1e3,
                    align_stage_ms=float(np.mean(al_ms)), placed_node=pm.node_id(int(state["res"].best_index[4])),
                    mapped_fraction=float(np.mean(rrecs["mapped"])), records_flagged=int(np.sum((rrecs["flags"] & 3) != 0)),
                    dp_pair_share=st["dp_pairs"] / max(st["n_items"], 1), dp_cells_per_step=st["dp_cells"],
                    gcups_align_stage=st["dp_cells"] / max(float(np.mean(al_ms)), 1e-9) / 1e6,
                    workload="tests/golden/isolate_R{1,2}.fastq.gz (2 x 51,169 real reads, mean %d bp, indels / N / adapters) x8, place + align" % r_mean)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_reads * args.steps / elapsed
        align_ms = float(np.mean(kernel_ms["align"]))
        dom_ms = float(np.mean(kernel_ms["align_dom"]))
        seed_ms = float(np.mean(kernel_ms["seed"]))
        score_ms = float(np.mean(kernel_ms["score"]))
        # Dominant kernel of the align stage, first launch over every pair of the batch (one launch per step); its duration
        # is measured with HIP events recorded on the launch stream (pmx_last_kernel_ms "align_dom").  Algorithmic HBM
        # bytes per read: 2 bit/base packed bases + 1 bit/base ambiguity words in, 32 B record +
        # 4 B per CIGAR op out (~93 B for a 150 bp read); times the reads of one launch.
        alg_bytes = n_reads * (mean_len * 3 / 8.0 + 32 + 4.0 * mean_cigar)
        if dom_ms <= 0:
            dom_ms = align_ms
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        dom_name = os.environ.get("PMX_BENCH_DOM_KERNEL", "k_align_compact (all pairs)" if not long_reads else "align stage (wave-per-read kernels)")
        # HBM traffic of that kernel per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
        # runs; profiles/r02/make_pmc_traffic.py writes the json); only valid for the workload it was collected on.
        traffic = None
        for rd in ("r02", "r01"):
            try:
                with open(os.path.join(ROOT, "profiles", rd, "pmc_traffic.json")) as fh:
                    pt = json.load(fh)
                if pt.get("reads_per_gpu") == n_reads and pt.get("read_len") == args.read_len and world == 1:
                    ent = pt.get("dominant_kernel") or pt.get("k_align_reads_tpp_round0")
                    traffic = float(ent["hbm_bytes_per_launch"])
                    dom_name = ent.get("name", dom_name)
                    break
            except (OSError, KeyError, ValueError, TypeError):
                traffic = None
        dp_cells = float(np.mean([s["dp_cells"] for s in dp_stats]))
        dp_pairs = float(np.mean([s["dp_pairs"] for s in dp_stats]))
        n_items = max(dp_stats[-1]["n_items"], 1)
        out = {
            "metric": "reads placed+aligned/sec, 10M×150bp vs 20k-genome PanMAN, 1/2/4/8 MI355X",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u8/i8 DP + u64 hash + f64 score", "data": "synthetic",
            "config": {"workload": ("%gM x %dbp synthetic %s reads %s vs SARS-CoV-2 20k PanMAN (39,999 nodes), place+align (%s%s)"
                                    % ((total_reads if args.scaling == "strong" else n_reads) / 1e6, args.read_len,
                                       "paired" if paired else "single-end long (2% sub, 1.5% ins, 1.5% del)",
                                       "in total, read-sharded over the ranks" if args.scaling == "strong" else "per GPU",
                                       "configs[3]" if long_reads else ("configs[2]" if args.scaling == "strong" else "configs[1]"),
                                       "" if world == 1 else "; seed index replicated, RCCL histogram all-gather + record/CIGAR gather")),
                       "reads_per_gpu": n_reads, "total_reads": total_reads, "read_len": args.read_len,
                       "index": "k=19,s=8,l=3,closed syncmers,flank-mask 250",
                       "aligner_preset": ("k=21,w=11,a=2,b=8,q=12,e=2,q2=24,e2=1 (src/mm_align.c:140-166)" if not long_reads else
                                          "map-hifi / map-ont branch of setup_minimap2 (src/mm_align.c:167-180)")},
            "value_host_to_host": None if h2h is None else h2h["value"],
            "host_to_host": h2h,
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": 8000.0,
                         "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic, "kernel_ms": dom_ms,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "mapping kernel (sketch, index probes, chaining, region logic, extension): integer/latency bound by "
                                 "construction (SURVEY 8d: S3/S4 are not HBM-bound), so the HBM fraction is low; `traffic` is what the "
                                 "kernel really moves (PMC), `achieved` prices only the compulsory input + output"},
            "dp": {"pair_share": dp_pairs / n_items, "cells_per_step": dp_cells,
                   "gcups_align_stage": dp_cells / max(align_ms, 1e-9) / 1e6,
                   "note": "ksw2 cells counted as q*min(t,2w+1) per DP actually run (SURVEY 8d); extensions / gap fills answered by the "
                           "proved closed-form shortcuts run no DP and count no cells; GCUPS = cells / whole align-stage time"},
            "kernels_ms": {"seed stage (k_seed_histogram, chunked)": seed_ms, "score stage (k_score_terms + k_score_chains)": score_ms,
                           "align stage (all tiers)": align_ms, "dominant align kernel": dom_ms},
            "two_batches_in_flight": overlapped,
            "real_reads": real,
            "checks": {"placed_node": placed_id, "mapped_fraction": mapped_frac, "records_flagged": flagged,
                       "unique_seeds": int(res.n_unique_seeds), "kept_seeds": int(res.readUniqueSeedCount),
                       "tiers": dp_stats[-1], "rank0_gather_has_every_cigar": gather_ok},
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU baseline is reported by the 1-GPU run only
            threads = max(1, os.cpu_count() or 1)     # all host cores, uncapped
            sample = args.cpu_sample or (int(min(n_reads, max(20000, 60000 * threads))) if not long_reads else int(min(n_reads, max(200, 150 * threads))))
            out["cpu_baseline"] = cpu_baseline(concat, off, index.arrays(), state["ref"], sample, threads, paired)
        else:
            out["cpu_baseline"] = None
    else:
        out = None
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    if out is None and saved_stdout_fd is not None:   # other ranks: whatever is still buffered goes to stderr too
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
    if out is not None:
        # the C buffers (the banner) are flushed to where descriptor 1 points now -- stderr -- before it is restored
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        if saved_stdout_fd is not None:
            os.dup2(saved_stdout_fd, 1)
            os.close(saved_stdout_fd)
            saved_stdout_fd = None
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
