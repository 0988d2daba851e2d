"""CPU tests of the ALIGN pipeline sources (panmap_amd/csrc/align/*.hpp compiled by g++ with PMX_W = 1,
tests/hostsim) against the reference's own aligner compiled from its sources (oracle/_ref).  They check the
host logic of the kernels -- sketch, seeding, chaining, region bookkeeping, ksw2 DP, mapq, pairing -- and
the thread-per-pair kernel's DP-request / replay protocol without a GPU; the HIP kernels themselves are
checked by tests/test_align_gpu.py."""
import os

import numpy as np
import pytest

import align_checks as ac
from conftest import GOLDEN


def _ref_genome():
    return b"".join(l.strip() for l in open(os.path.join(GOLDEN, "isolate.ref.fa"), "rb") if not l.startswith(b">"))


@pytest.fixture(scope="module")
def cases(pmx):
    g = _ref_genome()
    seqs, _, _ = pmx.read_fastq_paired(os.path.join(GOLDEN, "isolate_R1.fastq.gz"), os.path.join(GOLDEN, "isolate_R2.fastq.gz"))
    real = seqs[20000:26000]
    concat, off = pmx.simulate_paired_reads(g, 1500, seed=31, sub_rate=0.02)
    syn = [bytes(concat[off[i]:off[i + 1]]) for i in range(len(off) - 1)]
    syn = [r if i % 2 == 0 else pmx.reverse_complement(r) for i, r in enumerate(syn)]
    rng = np.random.default_rng(9)
    for i in range(0, 600, 2):   # indels and N runs
        r = bytearray(syn[i])
        p = int(rng.integers(15, 130))
        k = i % 6
        if k == 0:
            del r[p:p + int(rng.integers(1, 7))]
        elif k == 2:
            r[p:p] = bytes(rng.choice(list(b"ACGT"), int(rng.integers(1, 7))).astype(np.uint8))
        else:
            r[p:p + 3] = b"NNN"
        syn[i] = bytes(r)
    # mate 2 as sequenced: the orientation --refine feeds (src/placement.cpp:164-197), the mates' anchors on opposite strands
    # (regression: an adapter-dimer pair got one chain through both strands, lchain.c:197 compares unsigned)
    real_fastq = [s if i % 2 == 0 else pmx.reverse_complement(s) for i, s in enumerate(seqs[:6000])]
    return g, {"real": real, "synthetic": syn, "real_as_sequenced": real_fastq}


@pytest.mark.parametrize("name", ["real", "synthetic", "real_as_sequenced"])
def test_pipeline_sources_equal_reference(pmx, oracle, cases, name):
    g, sets = cases
    reads = sets[name]
    want = oracle.ref_align_reads_direct(g, reads, True, 8)
    got = ac.hostsim_align(g, reads, True)
    bad = ac.compare_results(got, want)
    assert not bad, bad[:10]
    assert sum(1 for x in got if x["flags"] & 3) <= 2


@pytest.mark.parametrize("name", ["real", "synthetic", "real_as_sequenced"])
def test_thread_per_pair_dp_service_replay(pmx, oracle, cases, name):
    """the tier-0 control flow: DP requests served out of line, pair replayed; pairs handed to the wave tiers
    (flag 0x8000) are excluded and must stay a small minority"""
    g, sets = cases
    reads = sets[name]
    want = oracle.ref_align_reads_direct(g, reads, True, 8)
    got = ac.hostsim_align(g, reads, True, tpp=True)
    keep = [i for i, x in enumerate(got) if not (x["flags"] & 0x8000)]
    assert len(keep) >= 0.9 * len(got), (len(keep), len(got))
    bad = ac.compare_results([got[i] for i in keep], [want[i] for i in keep])
    assert not bad, bad[:10]


@pytest.mark.parametrize("read_len,sub,seed", [(6000, 0.03, 21), (2500, 0.05, 22), (9000, 0.05, 23)])
def test_rearranged_long_reads_equal_reference(pmx, oracle, read_len, sub, seed):
    """the long-read branches (RMQ re-chaining on the restated AVL tree, inversion probe + inversion hits, divergence
    filter) on the host build of the pipeline against the compiled reference"""
    g = _ref_genome()
    reads = ac.rearranged_long_reads(pmx, g, 40, read_len, seed, sub=sub)
    want = oracle.ref_align_reads_direct(g, reads, False, 8)
    got = ac.hostsim_align(g, reads, False)
    assert not ac.compare_results(got, want)
    assert sum(1 for x in got if x["flags"] & 3) == 0


@pytest.mark.parametrize("copy_div,read_err,copy,read_len", [(0.03, 0.01, (3000, 23000), None), (0.003, 0.003, (7000, 18000), 15000)])
def test_strand_retained_secondaries_equal_reference(pmx, oracle, copy_div, read_err, copy, read_len):
    """a reference with a diverged inverted copy: the divergence estimate and the strand-retained filter decide about the
    opposite-strand secondary of every read"""
    ref, reads = ac.inverted_repeat_case(pmx, _ref_genome(), 24, 31, copy_div, read_err, copy=copy, read_len=read_len)
    want = oracle.ref_align_reads_direct(ref, reads, False, 8)
    got = ac.hostsim_align(ref, reads, False)
    assert not ac.compare_results(got, want)
    assert sum(1 for x in got if x["flags"] & 3) == 0


@pytest.mark.parametrize("tpp", [False, True])
def test_golden_fixture(pmx, tpp):
    """committed outputs of the reference aligner (tests/golden/align_golden.json.gz): no oracle/_ref needed"""
    g, cases_ = ac.golden_cases(pmx)
    for name, (reads, want) in cases_.items():
        got = ac.hostsim_align(g, reads, True, tpp=tpp)
        keep = [i for i, x in enumerate(got) if not (x["flags"] & 0x8000)]
        assert len(keep) >= 0.95 * len(got)
        bad = ac.compare_results([got[i] for i in keep], [want[i] for i in keep])
        assert not bad, (name, bad[:10])


def test_dp_shortcuts_against_the_dp_fuzz():
    """every answer ksw_shortcut gives on random extension / gap-fill problems (small alphabets, tandem repeats,
    0-3 substitutions, the odd N) must be what ksw_extd2 computes"""
    import ctypes as C
    L = ac.hostsim(False)
    L.hs_shortcut_fuzz.restype = C.c_int
    L.hs_shortcut_fuzz.argtypes = [C.c_uint64, C.c_int64, C.POINTER(C.c_int64), C.c_int]
    counts = (C.c_int64 * 3)()
    L.hs_shortcut_fuzz(12345, 400000, counts, 1)
    declined, agreed, bad = counts[0], counts[1], counts[2]
    assert bad == 0, (declined, agreed, bad)
    assert agreed > 50000, (declined, agreed)


def test_compact_tier_equals_reference(pmx, oracle, cases):
    """the compact LDS tier (align/aln_compact.hpp): every pair it finishes carries exactly the reference's result, on
    clean synthetic pairs (which it must finish nearly always), noisy ones, the real example reads and the indel / N
    cases (which it must hand on); mate 2 reverse-complemented on the fly gives the same answers"""
    g, sets = cases
    concat, off = pmx.simulate_paired_reads(g, 6000, seed=5)
    raw = [bytes(concat[off[i]:off[i + 1]]) for i in range(len(off) - 1)]
    clean = [r if i % 2 == 0 else pmx.reverse_complement(r) for i, r in enumerate(raw)]
    concat, off = pmx.simulate_paired_reads(g, 3000, seed=6, sub_rate=0.01)
    noisy = [bytes(concat[off[i]:off[i + 1]]) for i in range(len(off) - 1)]
    noisy = [r if i % 2 == 0 else pmx.reverse_complement(r) for i, r in enumerate(noisy)]
    floors = {"clean": 0.97, "noisy": 0.80, "real": 0.25, "synthetic": 0.0}
    floors["real_as_sequenced"] = 0.0
    for name, reads in (("clean", clean), ("noisy", noisy), ("real", sets["real"]), ("synthetic", sets["synthetic"]), ("real_as_sequenced", sets["real_as_sequenced"])):
        want = oracle.ref_align_reads_direct(g, reads, True, 8)
        got, done = ac.hostsim_align_compact(g, reads)
        idx = [i for i in range(len(want)) if done[i]]
        bad = ac.compare_results([got[i] for i in idx], [want[i] for i in idx])
        assert not bad, (name, bad[:10])
        assert len(idx) >= floors[name] * len(want), (name, len(idx), len(want))
    got_rc, done_rc = ac.hostsim_align_compact(g, raw, rc2=True)
    got, done = ac.hostsim_align_compact(g, clean)
    assert np.array_equal(done, done_rc) and not ac.compare_results(got_rc, got)
    # the two-kernel form (seeds through the hand-over words of k_compact_seeds, chain part from a copy of them):
    # same pairs finished, same records, with 16- and 32-bit position words
    import os
    for pos32 in (False, True):
        if pos32:
            os.environ["PMX_HS_COMPACT_POS32"] = "1"
        try:
            for reads in (clean, noisy, sets["real"], sets["real_as_sequenced"]):
                os.environ["PMX_HS_COMPACT_SPLIT"] = "1"
                split, done_s = ac.hostsim_align_compact(g, reads)
                del os.environ["PMX_HS_COMPACT_SPLIT"]
                fused, done_f = ac.hostsim_align_compact(g, reads)
                # (the chain kernel's first form keeps 48 anchors, the fused form and the second form 56: the pairs with 49 .. 56
                #  seeds -- under 3 % -- are the second form's; every pair the two-kernel form finishes, the fused one finishes alike)
                assert np.all(done_f[done_s == 1] == 1) and done_s.sum() >= 0.97 * done_f.sum(), (done_s.sum(), done_f.sum())
                idx = [i for i in range(len(done_s)) if done_s[i]]
                assert not ac.compare_results([split[i] for i in idx], [fused[i] for i in idx])
        finally:
            os.environ.pop("PMX_HS_COMPACT_SPLIT", None)
            os.environ.pop("PMX_HS_COMPACT_POS32", None)


def test_compact_tier_several_regions_per_mate(cases, pmx, oracle):
    """the compact tier's second form (align/aln_compact_multi.hpp, k_align_compact16_multi): up to four fragment chains and
    several regions per mate -- what mates that overlap on the reference produce (amplicon reads: 55 % of the real example
    pairs leave the one-region form for that reason).  Every pair it finishes carries the reference's record; it finishes
    every pair the one-region form finishes, with the same record, and most of the real pairs; short inserts (mates that
    overlap by design) and the two-kernel / 32-bit position forms included"""
    import os
    g, sets = cases
    concat, off = pmx.simulate_paired_reads(g, 4000, seed=15, mean_insert=160.0, sd_insert=25.0, sub_rate=0.004)
    raw = [bytes(concat[off[i]:off[i + 1]]) for i in range(len(off) - 1)]
    short_inserts = [r if i % 2 == 0 else pmx.reverse_complement(r) for i, r in enumerate(raw)]
    seqs, _, _ = pmx.read_fastq_paired(os.path.join(GOLDEN, "isolate_R1.fastq.gz"), os.path.join(GOLDEN, "isolate_R2.fastq.gz"))
    floors = {"real": 0.60, "synthetic": 0.0, "real_as_sequenced": 0.0, "short_inserts": 0.5, "real_all": 0.70}
    todo = dict(sets, short_inserts=short_inserts, real_all=seqs)
    try:
        for name, reads in todo.items():
            want = oracle.ref_align_reads_direct(g, reads, True, 8)
            os.environ.pop("PMX_HS_COMPACT_MULTI", None)
            one, done_one = ac.hostsim_align_compact(g, reads)
            os.environ["PMX_HS_COMPACT_MULTI"] = "1"
            got, done = ac.hostsim_align_compact(g, reads)
            idx = [i for i in range(len(want)) if done[i]]
            bad = ac.compare_results([got[i] for i in idx], [want[i] for i in idx])
            assert not bad, (name, bad[:10])
            assert len(idx) >= floors[name] * len(want), (name, len(idx), len(want))
            assert np.all(done[done_one == 1] == 1), name
            if name == "real_all":
                assert done.sum() > 2 * done_one.sum()
                continue
            for pos32 in (False, True):
                if pos32:
                    os.environ["PMX_HS_COMPACT_POS32"] = "1"
                os.environ["PMX_HS_COMPACT_SPLIT"] = "1"
                split, done_s = ac.hostsim_align_compact(g, reads)
                del os.environ["PMX_HS_COMPACT_SPLIT"]
                fused, done_f = ac.hostsim_align_compact(g, reads)
                os.environ.pop("PMX_HS_COMPACT_POS32", None)
                assert np.array_equal(done_s, done) and np.array_equal(done_f, done), (name, pos32)
                assert not ac.compare_results(split, got) and not ac.compare_results(fused, got), (name, pos32)
    finally:
        for k in ("PMX_HS_COMPACT_MULTI", "PMX_HS_COMPACT_SPLIT", "PMX_HS_COMPACT_POS32"):
            os.environ.pop(k, None)


def test_compact_tier_seed_order_merge_and_heap(pmx, oracle):
    """the compact tier orders a pair's seeds by a two-way merge when they are two strictly monotone runs without equal
    reference position words and by the reference's heap (map.c:102-166) otherwise: short inserts make the mates overlap
    (the same minimizer in both: equal words, the tie's pop order is the heap's), long ones never do -- both paths are
    taken, and every pair the tier finishes carries the reference's result"""
    import ctypes as C
    g = _ref_genome()
    L = ac.hostsim(False)
    cnt = (C.c_longlong * 4)()
    seen = {}
    for name, ins in (("overlapping", 200.0), ("apart", 420.0)):
        concat, off = pmx.simulate_paired_reads(g, 2000, seed=77, mean_insert=ins, sd_insert=15.0)
        raw = [bytes(concat[off[i]:off[i + 1]]) for i in range(len(off) - 1)]
        reads = [r if i % 2 == 0 else pmx.reverse_complement(r) for i, r in enumerate(raw)]
        want = oracle.ref_align_reads_direct(g, reads, True, 8)
        L.hs_compact_counts(cnt, 1)
        got, done = ac.hostsim_align_compact(g, reads)
        L.hs_compact_counts(cnt, 1)
        idx = [i for i in range(len(want)) if done[i]]
        assert len(idx) >= (0.6 if name == "overlapping" else 0.9) * len(want), (name, len(idx))
        bad = ac.compare_results([got[i] for i in idx], [want[i] for i in idx])
        assert not bad, (name, bad[:10])
        seen[name] = (int(cnt[3]), len(want))
    assert seen["apart"][0] >= 0.95 * seen["apart"][1], seen            # no shared minimizers: the merge orders them
    assert seen["overlapping"][0] <= 0.5 * seen["overlapping"][1], seen  # 100 shared bases: the heap decides the ties


def test_thread_per_pair_pipeline_ignores_what_its_work_arrays_held(pmx):
    """the per-pair work arrays of the general tier are slabs that earlier pairs (or earlier calls) have written: filled
    with different byte patterns before every pair (PMX_HS_POISON), the pipeline gives the same records"""
    import os
    g = _ref_genome()
    concat, off = pmx.simulate_paired_reads(g, 400, seed=5, sub_rate=0.02)
    reads = [bytes(concat[off[i]:off[i + 1]]) for i in range(len(off) - 1)]
    reads = [r if i % 2 == 0 else pmx.reverse_complement(r) for i, r in enumerate(reads)]
    base = ac.hostsim_align(g, reads, True, tpp=True)
    try:
        for pat in ("0xff", "0xa5"):
            os.environ["PMX_HS_POISON"] = pat
            assert not ac.compare_results(ac.hostsim_align(g, reads, True, tpp=True), base), pat
    finally:
        os.environ.pop("PMX_HS_POISON", None)


def test_sliced_reference_sketch_equals_the_sequential_one(pmx, sars):
    """the device index build sketches the reference in independent 32-base slices with w + k + 1 bases of run-in
    (align/aln_seed.hpp sketch_slice); the host build runs sketch.c:77-143 from end to end: same minimizers in the same
    order, on genomes and on sequences full of N runs, homopolymers and tandem repeats, for several slice lengths"""
    rng = np.random.default_rng(3)

    def rand_seq(n):
        s = rng.choice(np.frombuffer(b"ACGT", np.uint8), n)
        for _ in range(n // 300 + 1):
            p = int(rng.integers(0, n)); t = int(rng.integers(0, 4)); ln = int(rng.integers(1, 60))
            if t == 0:
                s[p:p + ln] = ord("N")
            elif t == 1:
                s[p:p + ln] = s[p]
            elif t == 2:
                u = s[p:p + int(rng.integers(2, 7))].copy()
                rep = np.tile(u, 40)[:3 * ln]
                s[p:p + len(rep)] = rep[:len(s[p:p + len(rep)])]
            else:
                s[p] = ord("N")
        return bytes(s)
    seqs = [sars.genome("node_7618")] + [rand_seq(n) for n in (50, 200, 1000, 5000, 20000, 777)] + [b"A" * 500, b"AC" * 300, b"N" * 100 + b"ACGTTGCA" * 50]
    for s in seqs:
        for w, k in ((11, 21), (5, 15), (12, 19), (10, 15)):
            x0, y0 = ac.hostsim_ref_sketch(s, w, k, 0)
            for sl in (32, 37, 128, 1000):
                x1, y1 = ac.hostsim_ref_sketch(s, w, k, sl)
                assert np.array_equal(x0, x1) and np.array_equal(y0, y1), (len(s), w, k, sl)
