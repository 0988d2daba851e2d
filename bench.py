#!/usr/bin/env python3
"""Benchmark of the panmap hot path on MI355X: reads placed + aligned per second.

One "step" = one pass of the hot path over one batch of synthetic reads whose ASCII bases + offsets are RESIDENT IN HBM when
the timed region starts, and which ends as alignment records + CIGARs in (pinned) host memory:
  2-bit packing -> syncmer / k-min-mer seeding + seed histogram -> [N>1: RCCL all-gather + merge of the per-rank histograms]
  -> node scoring down the PanMAN tree -> materialise the placed genome + build its minimizer index
  -> map + align every read pair -> [N>1: RCCL gather of the records AND the CIGAR arena to rank 0]
  -> D2H of the records + CIGAR arena on a copy stream.
`--pipelines` (default 2) batches are in flight at a time: each pipeline owns a context (= HIP stream), a placer, an
aligner, device staging, pinned output buffers and a host thread, and the pipelines take the steps alternately, so the
latency-bound parts of one batch (bail tail, host round trips, D2H) overlap the kernels of the other.  `value` counts the
reads of all K steps over the wall time of the whole timed region (barrier + synchronize on both sides, max over ranks).
`value_host_to_host` is the same step fed from pinned host memory (H2D included, double-buffered): the PCIe-inclusive rate.

Workload (default = BASELINE.json configs[2], the one the metric string is quoted on): 10M x 150 bp synthetic paired
reads vs the 20,000-genome SARS-CoV-2 PanMAN, `--scaling strong`: the 10M reads are split over the ranks (N=1: all 10M on
the one GPU, N=8: 1.25M each); the seed index is replicated per GPU.
  --scaling weak --reads-per-gpu 1000000      configs[1] per rank
  --read-len >= 500                           single-end long reads (configs[3]: `--scaling weak --reads-per-gpu 100000 --read-len 10000`)

The ONE JSON line (rank 0) carries, next to the contract keys:
  value_host_to_host     the PCIe-inclusive figure (inputs from pinned host memory), never `value`
  value_device_resident  one batch at a time, outputs left in HBM: the run the per-kernel timings come from
  pcie                   measured H2D / D2H GB/s of this box (pinned, 256 MB) and the PCIe-bound reads/s that follows
  roofline               dominant kernel vs the HBM peak (algorithmic bytes / its HIP-event duration in the resident run)
  roofline_valu          the same kernel against the VALU issue rate (wave-instructions from the PMC pass of profiles/,
                         only when that pass was collected on the sources of this build)
  roofline_by_stage      every stage (pack, seed, score, compact-seeds, compact-chain, tail) against its bound
  dp                     share of pairs that ran a ksw2 DP, DP cells per step, GCUPS
  real_reads             the repository's real example reads x8 through the same step
  cpu_baseline           the CPU path on the host cores, timed around bare C calls (no Python in the timed spans)
  checks.oracle          the GPU's results on the CPU baseline's sample against what the CPU path computed (same run)

Launched for N>1 as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def _digest(keep):
    h = hashlib.sha256()
    base = os.path.join(ROOT, "panmap_amd", "csrc")
    for dp, dn, fn in sorted(os.walk(base)):
        dn[:] = sorted(d for d in dn if d != "build")
        for f in sorted(fn):
            rel = os.path.relpath(os.path.join(dp, f), base)
            if not keep(rel):
                continue
            if f.endswith((".hip", ".hpp", ".h", ".cpp")):
                h.update(rel.encode())
                with open(os.path.join(dp, f), "rb") as fh:
                    h.update(fh.read())
    return h.hexdigest()[:16]


def source_digest():
    """digest of the sources the align stage is compiled from and launched by (align/, align_kernel*, api_align.hip, device/):
    a PMC pass under profiles/ is only quoted for an align kernel while it carries the digest of the build that runs."""
    return _digest(lambda rel: rel.startswith(("align", "device" + os.sep)) or rel == "api_align.hip")


def source_digest_place():
    """the same for the place stage's kernels (pack, seeding, scoring): place_kernels.*, api_place.hip, readset.hpp, device/"""
    return _digest(lambda rel: rel.startswith(("place_kernels", "device" + os.sep)) or rel in ("api_place.hip", "readset.hpp"))


def usable_cpus():
    """threads this process may really use: affinity mask, capped by the cgroup CPU quota when there is one"""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, p = fh.read().split()
            if q != "max":
                quota = float(q) / float(p)
    except (OSError, ValueError):
        pass
    return max(1, n), quota


def cpu_baseline(concat, off, index_arrays, placed_genome, sample_reads, threads, paired, quota):
    """The CPU path timed on this box's host cores on a bounded sample of the SAME workload.
    place leg = oracle restatement (oracle/oracle_place.c; the reference's placement.cpp cannot be built without
    panman/TBB/abseil), seeded on `threads` pthreads in ONE C call, then finalize + node scoring + best/tie rule;
    align leg = the reference's own aligner compiled from its sources (oracle/_ref: src/mm_align.c + vendored
    minimap2), timed around the bare align_reads_direct call with `threads` worker threads.  All ctypes marshalling
    and unpacking happens outside the timed spans.
    Returns (the cpu_baseline object of the JSON line, what the CPU path computed: the same-run parity check compares the
    GPU's results on the same reads with it -- SURVEY 8d "parity checks in the same run")."""
    from oracle import oracle as orc
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
    best, best_idx, ties = orc.best_ties(index_arrays["parent"], sc)
    t_score = time.perf_counter() - t0
    # ---- align leg (R2 reverse-complemented as readFastqPaired does; marshalling in C, not timed)
    prep = orc.prepare_align_call_flat(sub_concat, sub_off, paired, paired)
    t0 = time.perf_counter()
    orc.run_align_call(orc.rlib().align_reads_direct, placed_genome, prep, threads)
    t_align = time.perf_counter() - t0
    flat = orc.flatten_align_call(prep)
    mapped = int(flat["mapped"].sum())
    total = t_seed + t_score + t_align
    # the CPUs this process can really run on: the affinity mask capped by the cgroup quota (256 threads under a 16-CPU
    # quota are 16 cores' worth of work)
    eff = threads if not quota else max(1, min(threads, int(round(quota))))
    line = dict(value=n / total, unit="reads/s", cores=eff, threads_launched=threads, kind="reference",
                kind_by_leg={"align": "reference (oracle/_ref: src/mm_align.c + vendored minimap2, compiled where they lie)",
                             "place": "port (oracle/oracle_place.c restatement; placement.cpp needs panman/TBB/abseil)"},
                cgroup_cpu_quota=quota,
                legs={"seed_reads_per_s": n / t_seed, "score_s_per_sample": t_score, "align_reads_per_s": n / t_align},
                sample="%d of the workload's reads, %d threads on %d usable CPUs (affinity mask, capped by the cgroup quota): place leg = "
                       "oracle port of seeding (one pthread-chunked C call, %.2fs) + node scoring (%.2fs, once per sample, single thread "
                       "as a fixed cost); align leg = the reference's minimap2 via src/mm_align.c align_reads_direct, bare C call %.2fs; "
                       "%d/%d %s mapped" % (n, threads, eff, t_seed, t_score, t_align, mapped, len(flat["mapped"]), "pairs" if paired else "reads"))
    return line, dict(n=n, hist=(uh, uc), scores=sc, best=best, best_idx=best_idx, ties=ties, align=flat)


def oracle_checks(pmx, gpu, cpu, node_id, paired):
    """SURVEY 8d "parity checks in the same run": the GPU's results on the sample the CPU baseline ran on, against what the
    CPU path (oracle restatement / the compiled reference aligner) computed on the same reads.
    gpu = dict(hist=(hash, count), scores, result=PlacementResult, recs, cig); cpu = the second value of cpu_baseline()."""
    n = cpu["n"]
    hh, hc = gpu["hist"]
    uh, uc = cpu["hist"]
    out = {"reads": n,
           "histogram_equal": bool(np.array_equal(hh, uh) and np.array_equal(hc, uc)), "histogram_entries": int(len(uh)),
           "node_scores_bit_equal": bool(np.array_equal(np.ascontiguousarray(gpu["scores"]).view(np.uint64), np.ascontiguousarray(cpu["scores"]).view(np.uint64))),
           "nodes_scored": int(len(cpu["scores"]))}
    want_tsv = pmx.format_placement_tsv(pmx.PlacementResult(list(cpu["best"]), list(cpu["best_idx"]), cpu["ties"]), node_id)
    out["tsv_equal"] = bool(pmx.format_placement_tsv(gpu["result"], node_id) == want_tsv)
    # ---- per-read records: (pos, rs, re, qs, qe, mapq, rev, proper_frag) and the pair's / read's `mapped`, as
    # extract_align_result / align_worker_func leave them (src/mm_align.c:271-354): an unmapped read is pos = INT_MAX, rest 0
    recs, cig = gpu["recs"][:n], gpu["cig"]
    flat = cpu["align"]
    valid = (recs["mapped"] != 0) & ((recs["flags"] & 4) != 0)
    g = np.zeros((n, 8), np.int32)
    g[:, 0] = np.where(valid, recs["rs"] + 1, 2147483647)
    for j, f in enumerate(("rs", "re", "qs", "qe", "mapq", "rev", "proper_frag")):
        g[:, j + 1] = np.where(valid, recs[f].astype(np.int32), 0)
    rec_ok = np.all(g == flat["fields"], axis=1)
    g_mapped = recs["mapped"][0::2] if paired else recs["mapped"]
    map_ok = (g_mapped != 0) == (flat["mapped"] != 0)
    rec_ok &= np.repeat(map_ok, 2) if paired else map_ok
    # ---- CIGARs: operation count and a position-weighted 64-bit digest of the operations of every read
    def digests(ops, counts):
        counts = counts.astype(np.int64)
        tot = int(counts.sum())
        if tot == 0:
            return np.zeros(len(counts), np.uint64)
        first = np.cumsum(counts) - counts
        pos = np.arange(tot, dtype=np.int64) - np.repeat(first, counts)
        with np.errstate(over="ignore"):
            w = (ops[:tot].astype(np.uint64) + np.uint64(1)) * (np.uint64(0x9E3779B97F4A7C15) * (pos.astype(np.uint64) + np.uint64(1)) | np.uint64(1))
            cs = np.concatenate([np.zeros(1, np.uint64), np.cumsum(w, dtype=np.uint64)])
        return cs[first + counts] - cs[first]
    g_n = np.where(valid, recs["n_cigar"].astype(np.int64), 0)
    tot = int(g_n.sum())
    first = np.cumsum(g_n) - g_n
    idx = np.repeat(recs["cigar_off"].astype(np.int64) - first, g_n) + np.arange(tot, dtype=np.int64)
    g_ops = np.asarray(cig).view(np.uint32)[idx] if tot else np.zeros(0, np.uint32)
    cig_ok = (g_n == flat["n_cigar"].astype(np.int64)) & (digests(g_ops, g_n) == digests(flat["cigar"], flat["n_cigar"]))
    out.update({"records_equal": "%d/%d" % (int(rec_ok.sum()), n), "cigars_equal": "%d/%d" % (int(cig_ok.sum()), n),
                "records_all_equal": bool(rec_ok.all()), "cigars_all_equal": bool(cig_ok.all()),
                "cigar_ops_compared": int(flat["n_cigar"].sum()), "mapped_reference": int(flat["mapped"].sum()),
                "records_flagged_unsupported_or_overflow": int(np.sum((recs["flags"] & 3) != 0)),
                "against": "place: oracle/oracle_place.c (restatement, pinned by the reference's golden TSV and unit-test contracts); align: "
                           "oracle/_ref = the reference's src/mm_align.c + vendored minimap2 compiled where they lie; same reads, same run"})
    if not rec_ok.all():
        out["first_record_mismatches"] = [int(x) for x in np.nonzero(~rec_ok)[0][:8]]
    if not cig_ok.all():
        out["first_cigar_mismatches"] = [int(x) for x in np.nonzero(~cig_ok)[0][:8]]
    return out


class Sequencer:
    """Collectives issued from several host threads must reach the process group in ONE order on every rank.  The batches
    are numbered; batch i has a histogram exchange H_i and a result gather G_i; with P pipelines the order is H_0 .. H_{P-1},
    G_0, H_P, G_1, H_{P+1}, ... (P batches between their exchange and their gather: batch i+1 .. i+P-1 are seeded while batch
    i aligns), then the last P gathers; every rank derives it from the batch number alone."""

    def __init__(self, n_batches, n_pipes=2):
        self.n = n_batches
        self.p = max(1, min(n_pipes, n_batches))
        self.next = 0
        self.cv = threading.Condition()
        self.failed = None

    def ticket(self, kind, i):
        n, p = self.n, self.p
        if kind == "H":
            return i if i < p else 2 * i - p + 1
        return p + 2 * i if i <= n - p - 1 else n + i

    def run(self, kind, i, fn):
        t = self.ticket(kind, i)
        with self.cv:
            while self.next != t and self.failed is None:
                self.cv.wait(timeout=1.0)
            if self.failed is not None:
                raise RuntimeError("another pipeline failed: %s" % self.failed)
        try:
            return fn()
        except BaseException as e:   # noqa: BLE001  (wake the other thread, then re-raise)
            with self.cv:
                self.failed = repr(e)
                self.cv.notify_all()
            raise
        finally:
            with self.cv:
                self.next = t + 1
                self.cv.notify_all()

    def fail(self, e):
        with self.cv:
            self.failed = repr(e)
            self.cv.notify_all()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (10M reads per step: a step is 50-100 ms; 20 steps after 3 warm-up steps)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong")
    ap.add_argument("--total-reads", type=int, default=10000000, help="--scaling strong: reads of the whole job (BASELINE configs[2])")
    ap.add_argument("--reads-per-gpu", type=int, default=1000000, help="--scaling weak: reads per rank (BASELINE configs[1])")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--pipelines", type=int, default=0, help="batches in flight (0 = 2 for short reads, 1 for long reads)")
    ap.add_argument("--h2d-chunks", type=int, default=0, help="H2D chunks per batch (default 1)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="reads of the CPU baseline sample (0 = sized for ~10-20 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-real-reads", action="store_true")
    ap.add_argument("--no-resident", action="store_true", help="skip the one-batch-at-a-time figure (and the roofline objects that need it)")
    ap.add_argument("--no-host-to-host", action="store_true", help="skip the PCIe-inclusive figure (value_host_to_host)")
    ap.add_argument("--gather", choices=["sharded", "rccl"], default="sharded",
                    help="N>1: how the records reach the host -- sharded: every rank downloads its own part over its own PCIe link into "
                         "one shared-memory result set (pmx_dist_plan_alignments); rccl: everything to rank 0 over RCCL, then one download")
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

    # ---------------------------------------------------------------------------------------------- workload
    # SURVEY 8d to the letter: the source genome is the (splitmix64(42) mod 20000)-th leaf of the tree, the reads come from ONE
    # splitmix64(42) stream with a fixed draw order (panmap_amd/synth.py: simulate_paired_reads_8d).  The stream is
    # counter-based, so a rank draws exactly its stretch of the job's pairs: the job's reads do not depend on N.
    from panmap_amd import synth
    src_node = synth.source_leaf_8d(np.array([pm.parent(i) if i else 0 for i in range(pm.num_nodes)]), 42)
    src = pm.genome(src_node)
    long_reads = args.read_len >= 500
    paired = not long_reads
    unit = 2 if paired else 1
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
        first_pair = (lo // 2) if args.scaling == "strong" else rank * (my_reads // 2)
        concat, off = synth.simulate_paired_reads_8d(src, my_reads // 2, first_pair=first_pair, read_len=args.read_len, seed=42)
    n_reads = len(off) - 1
    max_len = int(np.max(np.diff(off))) if n_reads else 0
    mean_len = int(off[-1] // max(n_reads, 1))
    total_reads = n_reads * world if args.scaling == "weak" else args.total_reads // unit * unit
    # batches in flight: two; three when a rank's batch is small (a rank's share of a strong-scaling job: the fixed-latency
    # kernels of a step -- histogram finalisation, scoring, reference index, the tail tiers -- leave more of the chip idle).
    # Measured at 1.25M reads: 156 M reads/s with two, 179-187 with three, 196-213 with four, 162 with six; at 2.5M 212 / 224 /
    # 232-251; at 5M 275 / 253 / 246; at 10M three are not better (249-356 M).  Four would be the pick for small batches, but
    # four legs of forty-four ran 17x slower with four pipelines (a whole leg at 107 or 142 ms per step: 100 ms more per step
    # to the half millisecond, whatever the warm-up; ten runs in a row with three pipelines: none; profiles/r04/README.md
    # item 16).  Every context also owns hardware queues, and those are a budget too (item 14).
    n_pipes = args.pipelines or (1 if long_reads else (3 if n_reads <= 3000000 else 2))
    # (one chunk by default: with two batches in flight the upload of a batch overlaps the kernels of the other one, and a
    # batch seeded as a whole sizes its seed table once; --h2d-chunks > 1 packs + seeds range by range behind the copies)
    n_chunks = args.h2d_chunks or 1
    n_chunks = max(1, min(n_chunks, n_reads // unit or 1))
    bounds = [(n_reads // unit) * c // n_chunks * unit for c in range(n_chunks + 1)]

    params = pmx.TraversalParams()
    host_times = {} if os.environ.get("PMX_BENCH_HOST_TIMES") else None   # diagnostic: serialised per-phase wall times
    trace = [] if os.environ.get("PMX_BENCH_TRACE") else None             # diagnostic: host time stamps of the host->host phases

    # ------------------------------------------------------------------------------------------------- PCIe of this box
    def measure_pcie(nbytes=256 << 20, reps=3):
        h = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        out = {}
        for name, dst, s in (("h2d", d, h), ("d2h", h, d)):
            dst.copy_(s, non_blocking=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                dst.copy_(s, non_blocking=True)
            torch.cuda.synchronize()
            out[name + "_GBps"] = reps * nbytes / (time.perf_counter() - t0) / 1e9
        # both directions at once (the pipelined step uploads batch n+1 while it downloads batch n-1)
        h2 = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        d2 = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            with torch.cuda.stream(s1):
                d.copy_(h, non_blocking=True)
            with torch.cuda.stream(s2):
                h2.copy_(d2, non_blocking=True)
        torch.cuda.synchronize()
        out["duplex_each_GBps"] = reps * nbytes / (time.perf_counter() - t0) / 1e9
        return out

    pcie = measure_pcie()

    def measure_copy_ceiling(nbytes=1 << 30, reps=5):
        """what this box's HBM sustains for the simplest streaming kernel there is: a device-to-device copy of 1 GB
        (bytes read + bytes written per second), next to the 8 TB/s of the data sheet (SURVEY 8d "Metric")"""
        a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        b = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        b.copy_(a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        del a, b
        return 2.0 * nbytes / (ms * 1e-3) / 1e9

    copy_ceiling = measure_copy_ceiling() if rank == 0 else None

    # ------------------------------------------------------------------------------------------------------ pipelines
    h_concat = torch.from_numpy(concat).pin_memory()
    h_off = torch.from_numpy(off).pin_memory()
    sharded = dist_on and args.gather == "sharded"
    # (sharded: every rank maps the whole job's result set; rccl: rank 0 receives every rank's records)
    n_out = total_reads if (dist_on and (rank == 0 or sharded)) else n_reads
    cig_words_cap = max(n_out * 4, int(concat.size) // 4 * (world if dist_on and (rank == 0 or sharded) else 1), 4096)
    shm_keep = []

    class Pipe:
        """one batch in flight: context (= stream), placer, aligner, device staging, pinned outputs"""

        def __init__(self):
            self.ctx = pmx.Context(local_rank)
            self.placer = pmx.Placer(self.ctx, index)
            self.aligner = None
            self.stream = torch.cuda.ExternalStream(self.ctx.stream, device=dev) if self.ctx.stream else torch.cuda.current_stream(dev)
            self.copy_in = torch.cuda.Stream(device=dev)
            self.copy_out = torch.cuda.Stream(device=dev)
            # two staging slots: batch j+1 is uploaded into the other slot while batch j computes (double buffering)
            self.d_concat = [torch.empty(int(concat.size), dtype=torch.uint8, device=dev) for _ in range(2)]
            self.d_off = [torch.empty(n_reads + 1, dtype=torch.int64, device=dev) for _ in range(2)]
            self.up_ev = [None, None]
            self.dist = None
            self.out_recs = None
            self.out_cig = None
            self.pool = {}
            self.res = None
            self.ref = None
            self.gathered = None
            self.nw = 0

        def alloc_outputs(self):
            if self.out_recs is not None:
                return
            if sharded:
                # ONE result set for the node: a shared-memory segment made by rank 0, mapped by every rank and registered
                # with the HIP runtime (page-locked), so that each rank's download of ITS part is an asynchronous copy over
                # ITS OWN PCIe link.  (collective: every rank calls this for its pipelines in the same order)
                from multiprocessing import shared_memory
                nbytes = [n_out * 32, cig_words_cap * 4]
                names = [None, None]
                segs = []
                if rank == 0:
                    segs = [shared_memory.SharedMemory(create=True, size=max(b, 64)) for b in nbytes]
                    names = [sg.name for sg in segs]
                dist.broadcast_object_list(names, src=0)
                if rank != 0:
                    segs = [shared_memory.SharedMemory(name=nm) for nm in names]
                arrs = [np.ndarray((nbytes[0],), np.uint8, buffer=segs[0].buf), np.ndarray((nbytes[1],), np.uint8, buffer=segs[1].buf)]
                for a_ in arrs:
                    rc_ = torch.cuda.cudart().cudaHostRegister(a_.ctypes.data, a_.nbytes, 0)
                    if int(rc_) != 0:
                        raise RuntimeError("cudaHostRegister of the shared result set failed: %s" % rc_)
                self.out_recs = torch.from_numpy(arrs[0]).view(n_out, 32)
                self.out_cig = torch.from_numpy(arrs[1].view(np.int32))
                shm_keep.append((segs, arrs))
                dist.barrier()
            elif not dist_on or rank == 0:
                self.out_recs = torch.empty((n_out, 32), dtype=torch.uint8).pin_memory()
                self.out_cig = torch.empty(cig_words_cap, dtype=torch.int32).pin_memory()

        def read_set(self, key, d_concat, d_off_ptr, n):
            # read set objects reused from step to step (pmx_readset_rewrap_device: no hipMalloc, offsets handled on the device)
            if key in self.pool:
                self.pool[key].rewrap_device(d_concat.data_ptr(), d_off_ptr, n, int(concat.size), max_len)
            else:
                self.pool[key] = pmx.ReadSet.wrap_device(self.ctx, d_concat.data_ptr(), d_off_ptr, n, int(concat.size), max_len)
            return self.pool[key]

        def tick(self, name, t_prev):
            if host_times is None:
                return t_prev
            self.ctx.synchronize()
            t = time.perf_counter()
            host_times[name] = host_times.get(name, 0.0) + (t - t_prev) * 1e3
            return t

        def exchange_histograms(self):
            """pmx_dist_merge_histograms: RCCL all-gather of the per-rank (hash,count) histograms + integer merge on the device"""
            self.dist.merge_histograms(self.placer)

        def place_and_align(self, all_rs, n_total_reads, mean_read_len, is_paired, revcomp_mate2, seq=None, batch=0):
            """the hot path on reads already packed and seeded into the placer; `all_rs` = the read set the aligner runs on"""
            tk = time.perf_counter()
            if dist_on:
                if seq is not None:
                    seq.run("H", batch, self.exchange_histograms)
                else:
                    self.exchange_histograms()
                tk = self.tick("exchange", tk)
            res = self.placer.score(params, n_total_reads)
            tk = self.tick("score", tk)
            node = res.best_index[4]                      # bestLogContainmentNodeId (src/main.cpp:1771)
            ref = pm.genome(int(node))                    # getStringFromReference, every step (nothing cached)
            tk = self.tick("genome", tk)
            if self.aligner is None:
                self.aligner = pmx.Aligner(self.ctx, ref, mean_read_len)
            else:
                self.aligner.set_reference(ref, mean_read_len)   # mm_idx_str of the placed genome, every step
            tk = self.tick("ref_index", tk)
            self.aligner.align_readset(all_rs, paired=is_paired, revcomp_mate2=revcomp_mate2)
            tk = self.tick("align", tk)
            self.res, self.ref = res, ref

        def gather_results(self):
            """sharded: pmx_dist_plan_alignments -- the ranks agree on every rank's place in the one result set (the download
            follows, per rank); rccl: pmx_dist_gather_alignments -- fixed-size records + the CIGAR arena of every rank to rank
            0, cigar_off rebased"""
            if sharded:
                rb, wb, tr_, tw_ = self.dist.plan_alignments(self.aligner)
                self.gathered = (tr_, tw_)
            else:
                self.gathered = self.dist.gather_alignments(self.aligner, 0)       # (n_records, n_words) on rank 0

        # ---- the step with its inputs resident in HBM and its outputs left there (value_device_resident)
        def run_resident(self, rs):
            tk = self.tick("", time.perf_counter()) if host_times is not None else 0.0
            rs.pack()
            tk = self.tick("pack", tk)
            self.placer.reset()
            self.placer.add_reads(rs, params)
            if paired:
                rs.order_pairs()          # (the align stage's pair order, beside the scoring: pmx_readset_order_pairs)
            tk = self.tick("seed", tk)
            self.place_and_align(rs, total_reads, mean_len, paired, paired)
            if dist_on:
                tk = self.tick("", time.perf_counter()) if host_times is not None else 0.0
                self.gather_results()
                tk = self.tick("gather", tk)
            self.ctx.synchronize()

        # ---- the step of `value`: host memory -> host memory
        def upload(self, slot):
            """enqueue the H2D copies of one batch into a staging slot (copy-in stream): offsets first (8 B per read), then
            the bases chunk by chunk"""
            evs = []
            with torch.cuda.stream(self.copy_in):
                self.d_off[slot].copy_(h_off, non_blocking=True)
                ev_off = torch.cuda.Event()
                ev_off.record(self.copy_in)
                for c in range(n_chunks):
                    b0, b1 = int(off[bounds[c]]), int(off[bounds[c + 1]])
                    self.d_concat[slot][b0:b1].copy_(h_concat[b0:b1], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self.copy_in)
                    evs.append(ev)
            self.up_ev[slot] = (ev_off, evs)

        def run_h2h(self, seq=None, batch=0, slot=0, prefetch_next=False, resident_inputs=False):
            """one step.  resident_inputs: the batch's bases + offsets already lie in this staging slot (HBM) when the step
            starts -- the step of `value`; otherwise they are uploaded from pinned host memory first (value_host_to_host).
            Either way the records + CIGAR arena end in pinned host memory."""
            tr = [("start", time.perf_counter())] if trace is not None else None
            if resident_inputs:
                self.placer.reset()
                whole = self.read_set(("whole", slot), self.d_concat[slot], self.d_off[slot].data_ptr(), n_reads)
                whole.pack()
                self.placer.add_reads(whole, params)
                if paired:
                    whole.order_pairs()
                ev_off, evs = None, []
            else:
                if self.up_ev[slot] is None:
                    self.upload(slot)                 # the first batch of a run: nothing was prefetched
                if prefetch_next:
                    self.upload(slot ^ 1)             # the next batch of this pipeline travels while this one computes
                ev_off, evs = self.up_ev[slot]
                self.up_ev[slot] = None
                if tr is not None:
                    tr.append(("h2d_enqueued", time.perf_counter()))
                self.placer.reset()
                self.stream.wait_event(ev_off)
                # ONE read set over the whole staging buffer (word offsets from the device-resident offsets)
                whole = self.read_set(("whole", slot), self.d_concat[slot], self.d_off[slot].data_ptr(), n_reads)
            if resident_inputs:
                pass
            elif n_chunks == 1:
                self.stream.wait_event(evs[0])
                whole.pack()
                self.placer.add_reads(whole, params)
                if paired:
                    whole.order_pairs()
            else:
                # its reads are packed and seeded range by range as the chunks land (pmx_readset_pack_range /
                # pmx_place_add_reads_range)
                for c in range(n_chunks):
                    r0, r1 = bounds[c], bounds[c + 1]
                    if r1 <= r0:
                        continue
                    self.stream.wait_event(evs[c])
                    whole.pack_range(r0, r1)
                    self.placer.add_reads_range(whole, r0, r1, params)
            if tr is not None:
                tr.append(("seeded", time.perf_counter()))
            self.place_and_align(whole, total_reads, mean_len, paired, paired, seq=seq, batch=batch)
            if tr is not None:
                tr.append(("aligned", time.perf_counter()))
            al = self.aligner
            # records + CIGAR arena into pinned host memory on the pipeline's own download stream (pmx_align_fetch_async /
            # pmx_dist_fetch_gathered_async: the copies wait for the results, the next batch's kernels do not wait for the
            # copies); the pipeline goes on with its next batch, finish() waits for the last download
            if dist_on:
                if seq is not None:
                    seq.run("G", batch, self.gather_results)
                else:
                    self.gather_results()
                if sharded:
                    n_rec, nw = self.gathered
                    if nw > self.out_cig.numel() or n_rec > self.out_recs.shape[0]:
                        raise RuntimeError("shared output buffers too small")
                    self.dist.fetch_shard_async(self.aligner, self.out_recs.data_ptr(), self.out_cig.data_ptr(), self.copy_out.cuda_stream)
                else:
                    if rank != 0:
                        return 0
                    n_rec, nw = self.gathered
                    if nw > self.out_cig.numel() or n_rec > self.out_recs.shape[0]:
                        raise RuntimeError("pinned output buffers too small")
                    self.dist.fetch_gathered_async(self.out_recs.data_ptr(), self.out_recs.shape[0], self.out_cig.data_ptr(), self.out_cig.numel(), self.copy_out.cuda_stream)
            else:
                nw = al.cigar_words()
                if nw > self.out_cig.numel():
                    raise RuntimeError("pinned CIGAR buffer too small")
                al.fetch_async(self.out_recs.data_ptr(), self.out_recs.shape[0], self.out_cig.data_ptr(), self.out_cig.numel(), self.copy_out.cuda_stream)
            if tr is not None:
                tr.append(("d2h_enqueued", time.perf_counter()))
                trace.append((id(self), batch, tr))
            self.nw = nw
            return nw

        def finish(self):
            self.copy_out.synchronize()
            self.ctx.synchronize()

        def close(self):
            for p_ in self.pool.values():
                p_.close()
            self.pool = {}
            if self.dist is not None:
                self.dist.close()
                self.dist = None

    pipes = [Pipe() for _ in range(n_pipes)]
    main_pipe = pipes[0]
    if dist_on:
        # one communicator per pipeline (pmx_dist_init = ncclCommInitRank); the 128-byte ids travel over the process group
        # that the launcher's rendezvous set up.  test_gloo (ranks share a device): the library's host-directory transport.
        import tempfile
        for pi, pp in enumerate(pipes):
            box = [None, None]
            if rank == 0:
                box = [pmx.Dist.unique_id() if not test_gloo else os.urandom(pmx.Dist.ID_BYTES), tempfile.mkdtemp(prefix="pmx_dist_%d_" % pi) if test_gloo else None]
            dist.broadcast_object_list(box, src=0)
            if test_gloo:
                os.environ["PMX_DIST_HOST_DIR"] = box[1]
                pmx.reload_options()          # (the library reads its switches once: csrc/device/pmx_options.hpp)
            pp.dist = pmx.Dist(pp.ctx, box[0], rank, world)

    def sync_all():
        for pp in pipes:
            pp.ctx.synchronize()
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

    def run_batches(n_batches, resident_inputs=False):
        """n_batches steps over the pipelines: pipeline p takes the batches p, p + P, p + 2P, ... (host -> host: and uploads
        its next batch while it computes the current one); all downloads have landed when this returns"""
        seq = Sequencer(n_batches, len(pipes)) if (dist_on and len(pipes) > 1) else None
        errs = []

        def work(p):
            try:
                torch.cuda.set_device(local_rank)
                mine = list(range(p, n_batches, len(pipes)))
                for j, b in enumerate(mine):
                    pipes[p].run_h2h(seq, b, slot=j & 1, prefetch_next=(not resident_inputs) and j + 1 < len(mine), resident_inputs=resident_inputs)
                pipes[p].finish()
            except BaseException as e:   # noqa: BLE001
                errs.append(e)
                if seq is not None:
                    seq.fail(e)
        if len(pipes) == 1:
            work(0)
        else:
            th = [threading.Thread(target=work, args=(p,)) for p in range(len(pipes))]
            for t in th:
                t.start()
            for t in th:
                t.join()
        if errs:
            raise errs[0]

    # ---------------------------------------------------------------------------- value_host_to_host (PCIe-inclusive)
    for pp in pipes:
        pp.alloc_outputs()
    h2h = None
    if not args.no_host_to_host:
        run_batches(max(args.warmup, 2 * len(pipes)))   # (untimed; two steps per pipeline: the second sizes what the first one's counts asked for)
        sync_all()
        t0 = time.perf_counter()
        run_batches(args.steps)
        sync_all()
        el_h = max_over_ranks(time.perf_counter() - t0)
        h2h = dict(value=total_reads * args.steps / el_h, ms_per_step=el_h / args.steps * 1e3)
    # ------------------------------------------------------------------------------------------------------ value
    # Inputs resident in HBM when the timed region starts (every staging slot of every pipeline holds the batch: uploaded
    # here, untimed), the pipelines alternate the batches, records + CIGAR arena come back to pinned host memory.
    for pp in pipes:
        for slot in (0, 1):
            pp.d_off[slot].copy_(h_off, non_blocking=True)
            pp.d_concat[slot].copy_(h_concat, non_blocking=True)
            pp.up_ev[slot] = None
    torch.cuda.synchronize()
    run_batches(max(args.warmup, 2 * len(pipes)), resident_inputs=True)
    sync_all()
    t0 = time.perf_counter()
    run_batches(args.steps, resident_inputs=True)
    sync_all()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    if trace is not None and rank == 0:
        for pid, b, tr in sorted(trace, key=lambda x: x[2][0][1])[-2 * len(pipes) - 2:]:
            print("[bench trace] pipe %x batch %d: " % (pid & 0xffff, b) + " ".join("%s=+%.1fms" % (nm, (t - tr[0][1]) * 1e3) for nm, t in tr[1:])
                  + " (start at %.1f ms)" % ((tr[0][1] - t0) * 1e3), file=sys.stderr)
    h2h_recs = [None if pp.out_recs is None else pp.out_recs.numpy().view(pmx.REC_DTYPE).reshape(-1) for pp in pipes]
    h2h_nw = [pp.nw for pp in pipes]
    h2h_nodes = [None if pp.res is None else int(pp.res.best_index[4]) for pp in pipes]

    # ---------------------------------------------------------------- value_device_resident (one batch at a time, no PCIe)
    kernel_ms = {"align": [], "align_dom": [], "align_cseeds": [], "seed": [], "score": [], "pack": []}
    dp_stats = []
    resident = None
    d_concat = torch.from_numpy(concat).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    torch.cuda.synchronize()
    rs = pmx.ReadSet.wrap_device(main_pipe.ctx, d_concat.data_ptr(), d_off.data_ptr(), n_reads, int(concat.size), max_len, keepalive=(d_concat, d_off))
    if not args.no_resident:
        for _ in range(max(1, min(args.warmup, 2))):
            main_pipe.run_resident(rs)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            main_pipe.run_resident(rs)
            for k in kernel_ms:
                kernel_ms[k].append(main_pipe.ctx.kernel_ms(k))
            dp_stats.append(main_pipe.aligner.stats())
        sync_all()
        el_r = max_over_ranks(time.perf_counter() - t0)
        resident = dict(value=total_reads * args.steps / el_r, ms_per_step=el_r / args.steps * 1e3)
        if host_times is not None and rank == 0:
            n_st = args.steps + max(1, min(args.warmup, 2))
            print("[bench host times, ms/step, serialised] " + " ".join(f"{k}={v / n_st:.3f}" for k, v in host_times.items() if k), file=sys.stderr)
    else:
        main_pipe.run_resident(rs)
        dp_stats.append(main_pipe.aligner.stats())

    # sanity on the last resident step's output (not timed), and host->host == resident
    recs, cig = main_pipe.aligner.fetch()
    res = main_pipe.res
    placed_ref = main_pipe.ref
    mapped_frac = float(np.mean(recs["mapped"]))
    flagged = int(np.sum((recs["flags"] & 3) != 0))
    placed_id = pm.node_id(int(res.best_index[4]))
    mean_cigar = float(np.mean(recs["n_cigar"]))
    same_as_resident = None
    gather_ok = None

    def cigar_ops(r_, arena):
        """every CIGAR operation in record order (the arena is bump-allocated: its layout differs from run to run)"""
        k = r_["n_cigar"].astype(np.int64)
        tot = int(k.sum())
        if tot == 0:
            return np.zeros(0, np.uint32)
        first = np.cumsum(k) - k
        idx = np.repeat(r_["cigar_off"].astype(np.int64) - first, k) + np.arange(tot, dtype=np.int64)
        return np.asarray(arena).view(np.uint32)[idx]

    if rank == 0:
        same_as_resident = True
        want_ops = cigar_ops(recs, cig)
        for pp, r_, nw_, node_ in zip(pipes, h2h_recs, h2h_nw, h2h_nodes):
            mine = r_[:n_reads]
            ok = node_ == int(res.best_index[4]) and all(np.array_equal(mine[f], recs[f]) for f in ("rs", "re", "qs", "qe", "mapq", "rev", "proper_frag",
                                                                                                      "mapped", "n_cigar", "flags", "score"))
            ok = ok and (nw_ == len(cig) if not dist_on else nw_ >= len(cig))
            ok = ok and np.array_equal(cigar_ops(mine, pp.out_cig[:nw_].numpy()), want_ops)
            same_as_resident = bool(same_as_resident and ok)
        if dist_on:
            # rank 0 holds every rank's records, and the CIGARs of the other ranks' records are reachable through the rebased offsets
            gather_ok = bool(all(len(r_) == total_reads for r_ in h2h_recs))
            for pp, r_, nw_ in zip(pipes, h2h_recs, h2h_nw):
                k = r_["n_cigar"].astype(np.int64)
                ops = cigar_ops(r_, pp.out_cig[:nw_].numpy())
                spans = (r_["re"] - r_["rs"]).astype(np.int64)
                # reference-consuming operations (M, D, N, =, X) of every record add up to its reference span
                oplen, opc = (ops >> 4).astype(np.int64), ops & 15
                ref_len = np.where(np.isin(opc, (0, 2, 3, 7, 8)), oplen, 0)
                owner = np.repeat(np.arange(len(r_)), k)
                got = np.bincount(owner, weights=ref_len, minlength=len(r_)).astype(np.int64)
                has = (r_["flags"] & 4) != 0
                gather_ok = bool(gather_ok and int(k.sum()) == nw_ and np.array_equal(got[has], spans[has]))

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
        ctx = main_pipe.ctx
        rrs = pmx.ReadSet.wrap_device(ctx, rd_concat.data_ptr(), rd_off.data_ptr(), len(rr), int(r_concat.size), int(np.max(np.diff(r_off))),
                                      keepalive=(rd_concat, rd_off))

        def step_real():
            rrs.pack()
            main_pipe.placer.reset()
            main_pipe.placer.add_reads(rrs, params)     # (canonical seeds: the orientation of R2 does not matter to the place stage)
            main_pipe.place_and_align(rrs, len(rr), r_mean, True, False)
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
        st = main_pipe.aligner.stats()
        rrecs, _ = main_pipe.aligner.fetch()
        # the same step with two batches in flight, as `value` is measured (one pipeline per batch; the records stay in HBM):
        # the thread-per-pair passes and the DP service of the pairs that need a DP fill a fraction of the chip and last as
        # long as their slowest wave -- the other batch's kernels run beside them
        real_two = None
        if len(pipes) > 1:
            sets = [rrs] + [pmx.ReadSet.wrap_device(pp.ctx, rd_concat.data_ptr(), rd_off.data_ptr(), len(rr), int(r_concat.size), int(np.max(np.diff(r_off))),
                                                    keepalive=(rd_concat, rd_off)) for pp in pipes[1:]]
            n_each = 4

            def work_real(pi, n):
                pp, rs_ = pipes[pi], sets[pi]
                for _ in range(n):
                    rs_.pack()
                    pp.placer.reset()
                    pp.placer.add_reads(rs_, params)
                    pp.place_and_align(rs_, len(rr), r_mean, True, False)
                pp.ctx.synchronize()

            def both(n):
                th = [threading.Thread(target=work_real, args=(pi, n)) for pi in range(len(pipes))]
                for t_ in th:
                    t_.start()
                for t_ in th:
                    t_.join()
            both(2)          # (untimed; two steps per pipeline, as before the timed region of `value`)
            t0 = time.perf_counter()
            both(n_each)
            el4 = time.perf_counter() - t0
            r2, _ = pipes[-1].aligner.fetch()
            real_two = dict(value=len(rr) * n_each * len(pipes) / el4, ms_per_step=el4 / (n_each * len(pipes)) * 1e3, batches_in_flight=len(pipes),
                            equals_one_at_a_time=bool(np.array_equal(r2[["rs", "re", "qs", "qe", "mapq", "rev", "proper_frag", "mapped", "n_cigar", "flags", "score"]],
                                                                     rrecs[["rs", "re", "qs", "qe", "mapq", "rev", "proper_frag", "mapped", "n_cigar", "flags", "score"]])))
            for s_ in sets[1:]:
                s_.close()
        real = dict(value=len(rr) * n_real_steps / el3, unit="reads/s", reads=len(rr), ms_per_step=el3 / n_real_steps * 1e3,
                    align_stage_ms=float(np.mean(al_ms)), placed_node=pm.node_id(int(main_pipe.res.best_index[4])),
                    mapped_fraction=float(np.mean(rrecs["mapped"])), records_flagged=int(np.sum((rrecs["flags"] & 3) != 0)),
                    dp_pair_share=st["dp_pairs"] / max(st["n_items"], 1), dp_cells_per_step=st["dp_cells"],
                    gcups_align_stage=st["dp_cells"] / max(float(np.mean(al_ms)), 1e-9) / 1e6, tiers=st,
                    inputs="resident in HBM (one batch at a time)", batches_in_flight_run=real_two,
                    note="the sample is replicated x8 to fill the chip: every read is an 8-fold duplicate for the place stage's read collapse "
                         "(its seeding runs on an eighth of the reads); the align stage -- most of this step -- does not collapse",
                    workload="tests/golden/isolate_R{1,2}.fastq.gz (2 x 51,169 real reads, mean %d bp, indels / N / adapters) x8, place + align" % r_mean)

    # ------------------------------------------------------------------------------- the DP kernel on its own (one GPU only)
    # ksw_extd2 requests as the real reads post them (extensions of ~60-110 query bases against targets up to 128, 4 %
    # substitutions, 1 % indels; one in three a gap fill) through the grouped DP service: cells = sum of qlen * tlen
    dp_service = None
    if world == 1 and not long_reads and main_pipe.aligner is not None:
        rng_dp = np.random.default_rng(99)
        n_req = 65536
        tl = rng_dp.integers(96, 129, n_req)
        ql = np.minimum(tl, rng_dp.integers(60, 113, n_req))
        t_all = rng_dp.integers(0, 4, int(tl.sum()), dtype=np.uint8)
        t_off2 = np.concatenate([[0], np.cumsum(tl)])
        qs_, ts_ = [], []
        for i in range(n_req):
            t = t_all[t_off2[i]:t_off2[i + 1]]
            q = t[:ql[i]].copy()
            hit = rng_dp.random(len(q)) < 0.04
            q[hit] = (q[hit] + 1 + rng_dp.integers(0, 3, int(hit.sum()))) % 4
            if rng_dp.random() < 0.5:
                cut = int(rng_dp.integers(1, len(q)))
                q = np.delete(q, cut)
            qs_.append(q); ts_.append(t)
        kinds = np.array([0x08, 0x40, 0xc2], np.int32)[np.arange(n_req) % 3]
        sc = main_pipe.aligner.scoring()
        res_dp, dp_ms = main_pipe.aligner.dp_batch(qs_, ts_, -1, sc["zdrop"], sc["end_bonus"], kinds, reps=5)
        cells = float(sum(len(a) * len(b) for a, b in zip(qs_, ts_)))
        dp_service = dict(kernel="k_align_dp_group (eight lanes per request, rows as a systolic pipeline)", requests=n_req, served=int(res_dp["served"].sum()),
                          cells=cells, kernel_ms=dp_ms, gcups=cells / max(dp_ms, 1e-9) / 1e6,
                          note="ksw_extd2 cells (qlen * tlen, full matrix: the band never cuts these) / duration of the service kernel, best of 5 launches; "
                               "results equal ksw_extd2_sse of the reference (tests/test_dp_service_gpu.py)")

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_reads * args.steps / elapsed
        # bytes one step moves over PCIe (per rank; rank 0 downloads the gathered records of every rank)
        up_bytes = int(concat.size) + 8 * (n_reads + 1)
        down_bytes = 32 * n_out + 4 * max(h2h_nw)
        t_pcie = max(up_bytes / (pcie["h2d_GBps"] * 1e9), down_bytes / (pcie["d2h_GBps"] * 1e9))
        pcie.update(h2d_bytes_per_step=up_bytes, d2h_bytes_per_step=down_bytes, bound_reads_per_s=total_reads / t_pcie,
                    note="pinned host memory, 256 MB copies; bound = total reads / max(H2D time, D2H time) of one step at these rates (rank 0)")
        out = {
            "metric": "reads placed+aligned/sec, %gM×%dbp vs 20k-genome PanMAN, 1/2/4/8 MI355X" % (total_reads / 1e6, args.read_len),
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u8/i8 DP + u64 hash + f64 score", "data": "synthetic",
            "config": {"workload": ("%gM x %dbp synthetic %s reads %s vs SARS-CoV-2 20k PanMAN (39,999 nodes), place+align, inputs resident in HBM -> records in pinned host memory (%s%s)"
                                    % ((total_reads if args.scaling == "strong" else n_reads) / 1e6, args.read_len,
                                       "paired" if paired else "single-end long (2% sub, 1.5% ins, 1.5% del)",
                                       "in total, read-sharded over the ranks" if args.scaling == "strong" else "per GPU",
                                       "configs[3]" if long_reads else ("configs[2]" if args.scaling == "strong" and args.total_reads == 10000000 else
                                                                        ("configs[1]" if n_reads == 1000000 else "custom size")),
                                       "" if world == 1 else "; seed index replicated, RCCL histogram all-gather + record/CIGAR gather")),
                       "reads_per_gpu": n_reads, "total_reads": total_reads, "read_len": args.read_len,
                       "warmup_steps_run": max(args.warmup, 2 * len(pipes)),
                       "source_node": pm.node_id(src_node),
                       "generator": ("SURVEY 8d: leaf number splitmix64(42) mod 20000 of the tree, one splitmix64(42) stream, 35 draws per pair "
                                     "(start, insert ~ N(300,30) in [150,600], 0.2 % i.i.d. substitutions as geometric gaps), FR, no indels / N"
                                     if not long_reads else "numpy PCG64 seed 43 + rank: 2 % sub, 1.5 % ins, 1.5 % del"),
                       "batches_in_flight": len(pipes), "h2d_chunks": n_chunks,
                       "index": "k=19,s=8,l=3,closed syncmers,flank-mask 250",
                       "aligner_preset": ("k=21,w=11,a=2,b=8,q=12,e=2,q2=24,e2=1 (src/mm_align.c:140-166)" if not long_reads else
                                          "map-hifi / map-ont branch of setup_minimap2 (src/mm_align.c:167-180)")},
            "value_is": "inputs (ASCII bases + offsets of the batch) resident in HBM when the timed region starts -> pack -> seed -> score -> "
                        "genome -> reference index -> align -> records + CIGAR arena in pinned host memory; %d batch(es) in flight" % len(pipes),
            "equals_device_resident_run": same_as_resident,
            "value_host_to_host": None if h2h is None else h2h["value"],
            "host_to_host": None if h2h is None else dict(h2h, note="the same step fed from pinned host memory (H2D of the bases + offsets, double-buffered, "
                                                                    "overlapping the other batch's kernels): PCIe-inclusive, never `value`"),
            "value_device_resident": None if resident is None else resident["value"],
            "device_resident": None if resident is None else dict(resident, note="one batch at a time, outputs left in HBM: the run the kernel timings come from"),
            "pcie": pcie,
            "host_to_host_over_min_of_value_and_pcie_bound": None if h2h is None else h2h["value"] / min(value, pcie["bound_reads_per_s"]),
        }
        if resident is not None:
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
            dom_name = "k_align_compact16 (all pairs; seeds from k_compact_seeds16)" if not long_reads else "k_align_reads_w4 (wave per read)"
            cseeds_ms = float(np.mean(kernel_ms["align_cseeds"]))
            cseeds_ms = cseeds_ms if cseeds_ms >= 0 else None   # (fused form / long reads: no such launch)
            # HBM traffic and wave-instruction counts of that kernel per launch from the PMC passes (rocprofv3 --pmc in separate
            # runs; profiles/rNN/make_pmc_traffic.py writes the json): quoted only for the workload AND the sources it was
            # collected on (source_digest), otherwise null.
            traffic, valu = None, None
            digest = source_digest()
            digest_place = source_digest_place()
            props = torch.cuda.get_device_properties(dev)
            clock_hz = float(getattr(props, "clock_rate", 0)) * 1e3 or 2.4e9
            simds = props.multi_processor_count * 4
            pt = None
            for rnd in ("r04", "r03"):
                try:
                    with open(os.path.join(ROOT, "profiles", rnd, "pmc_traffic.json")) as fh:
                        cand = json.load(fh)
                    if cand.get("reads_per_gpu") == n_reads and cand.get("read_len") == args.read_len and world == 1:
                        pt = cand
                        break
                except (OSError, ValueError):
                    pass

            def pmc(key, stage_digest, which):
                """the PMC entry of a kernel, only when it was collected on this workload AND on the sources of that stage"""
                if pt is None or key not in pt:
                    return None
                want = pt.get("source_digest_place") if which == "place" else pt.get("source_digest")
                return pt[key] if want == stage_digest else None

            def valu_obj(ent, ms, name):
                if not ent or not ent.get("valu_wave_insts_per_launch"):
                    return None
                # MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles on a SIMD that holds >= 2 waves
                slots = simds * clock_hz / 2.0 * ms * 1e-3
                return {"bound": "valu-issue", "kernel": name, "valu_wave_insts": ent["valu_wave_insts_per_launch"],
                        "salu_wave_insts": ent.get("salu_wave_insts_per_launch"), "simds": simds, "clock_hz": clock_hz,
                        "cycles_per_wave64_valu": 2, "issue_slots": slots, "frac": ent["valu_wave_insts_per_launch"] / slots,
                        "wait_any_share_of_wave_cycles": ent.get("wait_any_share")}
            ent = pmc("dominant_kernel", digest, "align")
            if ent:
                traffic = float(ent["hbm_bytes_per_launch"])
                valu = valu_obj(ent, dom_ms, dom_name)
            # ---- every stage against the roofline that bounds it (SURVEY 8d "Binding roofline per stage" and its algorithmic
            # bytes per unit; DESIGN.md section 4 states them).  kernel_ms = HIP events on the context stream in the resident run;
            # traffic / instruction counts from the PMC passes of profiles/ when their digest matches this build.
            pack_ms = float(np.mean(kernel_ms["pack"]))
            n_changes = int(index.info.n_changes) if hasattr(index.info, "n_changes") else None

            def stage(ms, alg_bytes, bound, kernels, ent=None, launches=1, extra=None):
                if ms is None or ms <= 0:
                    return None
                ach = alg_bytes / (ms * 1e-3) / 1e9
                o = {"kernels": kernels, "bound": bound, "kernel_ms": ms, "algorithmic_bytes": alg_bytes, "achieved": ach, "peak": 8000.0,
                     "unit": "GB/s", "frac": ach / 8000.0, "frac_of_copy_ceiling": (ach / copy_ceiling) if copy_ceiling else None,
                     "traffic": (float(ent["hbm_bytes_per_launch"]) * launches) if ent else None}
                if ent and ent.get("valu_wave_insts_per_launch"):
                    v = valu_obj({k: (v * launches if isinstance(v, (int, float)) and k.endswith("per_launch") else v) for k, v in ent.items()}, ms, kernels)
                    o["valu_issue_frac"] = v["frac"]
                    o["wait_any_share"] = ent.get("wait_any_share")
                if extra:
                    o.update(extra)
                return o
            seeds_per_read = 37.6 * mean_len / 150.0          # SURVEY 8d (probe: 33.2 per 133-bp read)
            tail_ms = max(align_ms - dom_ms - (cseeds_ms or 0.0), 0.0)
            by_stage = {
                "pack (k_pack_reads)": stage(pack_ms, n_reads * (mean_len + mean_len * 3 / 8.0), "hbm", "k_pack_reads", pmc("k_pack_reads", digest_place, "place"),
                                             extra={"bytes_per_read": mean_len + mean_len * 3 / 8.0}),
                "seed (S1: syncmers + k-min-mers + histogram)": stage(
                    seed_ms, n_reads * (mean_len * 3 / 8.0 + 8.0 * seeds_per_read), "hbm", "k_seed_* (+ read dedup, table clear)", None,
                    extra={"bytes_per_read": mean_len * 3 / 8.0 + 8.0 * seeds_per_read,
                           "rmw_variant": {"bytes_per_read": mean_len * 3 / 8.0 + 16.0 * seeds_per_read,
                                           "achieved": n_reads * (mean_len * 3 / 8.0 + 16.0 * seeds_per_read) / (seed_ms * 1e-3) / 1e9,
                                           "frac": n_reads * (mean_len * 3 / 8.0 + 16.0 * seeds_per_read) / (seed_ms * 1e-3) / 1e9 / 8000.0},
                           "traffic": (float(pt["seed_stage"]["hbm_bytes_per_step"]) if pt and pt.get("seed_stage") and pt.get("source_digest_place") == digest_place else None)}),
                "score (S2: k_score_terms + k_score_chains)": stage(
                    score_ms, 28.0 * (n_changes or 0), "hbm/L2 latency (read-count independent)", "k_score_terms + k_score_chains",
                    pmc("k_score_chains", digest_place, "place"), extra={"bytes_per_seed_change": 28.0, "seed_changes": n_changes}) if n_changes else None,
                "compact-seeds (S3: sketch + index probes)": stage(
                    cseeds_ms, n_reads * (mean_len * 3 / 8.0 + 4.0 * 20.0), "valu-issue (primary), hbm (secondary)", "k_compact_seeds16",
                    pmc("k_compact_seeds", digest, "align"), extra={"bytes_per_read": mean_len * 3 / 8.0 + 80.0}) if cseeds_ms else None,
                "compact-chain (S3/S4: merge, chain, regions, extension, mapq, pairing)": stage(
                    dom_ms, alg_bytes, "valu-issue / LDS latency (primary), hbm (secondary)", dom_name, ent, extra={"bytes_per_read": alg_bytes / max(n_reads, 1)}),
                "tail (bails of the compact tier: DP service + replay / wave tier)": (
                    {"kernels": "k_align_compact_replay / k_align_dp_group / k_align_reads_*", "bound": "latency (fixed per step)", "kernel_ms": tail_ms,
                     "pairs": int(dp_stats[-1]["n_items"] - dp_stats[-1]["compact_tier_items"]), "algorithmic_bytes": 0.0, "achieved": 0.0, "peak": 8000.0,
                     "unit": "GB/s", "frac": 0.0, "traffic": None,
                     "note": "align stage minus the two compact kernels (includes the pair-order sort); priced in ms, not bytes"} if not long_reads else None),
            }
            by_stage = {k: v for k, v in by_stage.items() if v is not None}
            dp_cells = float(np.mean([s["dp_cells"] for s in dp_stats]))
            dp_pairs = float(np.mean([s["dp_pairs"] for s in dp_stats]))
            n_items = max(dp_stats[-1]["n_items"], 1)
            out.update({
                "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": 8000.0,
                             "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic, "kernel_ms": dom_ms,
                             "algorithmic_bytes_per_launch": alg_bytes, "source_digest": digest,
                             "binding": "not HBM: VALU issue / LDS latency -- see roofline_valu and roofline_by_stage; the HBM-bound stages are pack and seed",
                             "note": "mapping kernel (heap merge, chaining, region logic, extension, mapq, pairing; short reads: the sketch and the "
                                     "index probes run before it in k_compact_seeds16, see kernels_ms): integer/latency bound by "
                                     "construction (SURVEY 8d: S3/S4 are not HBM-bound), so the HBM fraction is low; `traffic` is what the "
                                     "kernel really moves (PMC; includes the seed hand-over it reads), `achieved` prices only the stage's "
                                     "compulsory input + output (SURVEY 8d: 93 B per read)"},
                "roofline_valu": valu,
                "roofline_by_stage": by_stage,
                "hbm_copy_ceiling": {"GBps": copy_ceiling, "spec_GBps": 8000.0,
                                     "note": "device-to-device copy of 1 GB on this box (read + written bytes per second): what a pure streaming kernel gets"},
                "dp": {"pair_share": dp_pairs / n_items, "cells_per_step": dp_cells,
                       "gcups_align_stage": dp_cells / max(align_ms, 1e-9) / 1e6,
                       "note": "ksw2 cells counted as q*min(t,2w+1) per DP actually run (SURVEY 8d); extensions / gap fills answered by the "
                               "proved closed-form shortcuts run no DP and count no cells; GCUPS = cells / whole align-stage time"},
                "kernels_ms": {"pack (k_pack_reads)": pack_ms, "seed stage (k_seed_histogram, chunked)": seed_ms, "score stage (k_score_terms + k_score_chains)": score_ms,
                               "align stage (all tiers)": align_ms, "dominant align kernel": dom_ms,
                               "compact tier, sketch + probes (k_compact_seeds16)": cseeds_ms,
                               "note": "HIP-event durations in the device-resident run (one batch at a time)"},
            })
        else:
            out.update({"roofline": None, "roofline_valu": None})
        out.update({
            "real_reads": real,
            "dp_service": dp_service,
            "checks": {"placed_node": placed_id, "mapped_fraction": mapped_frac, "records_flagged": flagged,
                       "unique_seeds": int(res.n_unique_seeds), "kept_seeds": int(res.readUniqueSeedCount),
                       "tiers": dp_stats[-1], "rank0_gather_has_every_cigar": gather_ok},
        })
        if not args.no_cpu_baseline and world == 1 and not dist_on:   # the CPU baseline is reported by the 1-GPU run only
            threads, quota = usable_cpus()
            sample = args.cpu_sample or (int(min(n_reads, max(20000, 60000 * threads))) if not long_reads else int(min(n_reads, max(200, 150 * threads))))
            n_s = min(sample, n_reads)
            if paired:
                n_s &= ~1
            # the GPU's results on exactly the reads of the CPU sample (one untimed step of the same code path; on the driver's
            # default run the sample is the whole batch) ...
            rs_s = pmx.ReadSet.wrap_device(main_pipe.ctx, d_concat.data_ptr(), d_off.data_ptr(), n_s, int(off[n_s]), max_len, keepalive=(d_concat, d_off))
            rs_s.pack()
            main_pipe.placer.reset()
            main_pipe.placer.add_reads(rs_s, params)
            main_pipe.place_and_align(rs_s, n_s, mean_len, paired, paired)
            main_pipe.ctx.synchronize()
            g_recs, g_cig = main_pipe.aligner.fetch()
            g_sc, _, _ = main_pipe.placer.node_outputs()
            gpu_side = dict(hist=main_pipe.placer.histogram(), scores=g_sc, result=main_pipe.res, recs=g_recs, cig=g_cig)
            # ... and the CPU path on them, timed (aligned against the genome the GPU placed the sample on; the placement itself
            # is compared through the node scores and the TSV text)
            out["cpu_baseline"], cpu_side = cpu_baseline(concat, off, index.arrays(), main_pipe.ref, n_s, threads, paired, quota)
            out["checks"]["oracle"] = oracle_checks(pmx, gpu_side, cpu_side, pm.node_id, paired)
            rs_s.close()
        else:
            out["cpu_baseline"] = None
    else:
        out = None
    for pp in pipes:
        pp.close()
    if shm_keep:
        torch.cuda.synchronize()
        for pp in pipes:
            pp.out_recs = pp.out_cig = None
        h2h_recs = None
        if dist_on:
            dist.barrier()
        for segs, arrs in shm_keep:
            for a_ in arrs:
                torch.cuda.cudart().cudaHostUnregister(a_.ctypes.data)
        for segs, arrs in shm_keep:
            del arrs
            for sg in segs:
                try:
                    sg.close()
                except BufferError:
                    pass          # (a view is still alive somewhere: the mapping goes with the process)
                if rank == 0:
                    sg.unlink()
        shm_keep.clear()
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
