"""GPU parity tests for the ALIGN stage: HIP kernel (through the C ABI) vs the reference's own aligner
compiled from its sources (oracle/_ref: src/mm_align.c + vendored minimap2).  Bit-exact on
pos / rs / re / qs / qe / mapq / rev / proper_frag / CIGAR."""
import os

import numpy as np
import pytest

import align_checks as ac
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["default_tiers", "compact_off", "dp_service_forced", "wave_per_pair_only"])
def _tier_mode(request, monkeypatch):
    """Every test of this module runs four times: with the library's own tier choice (compact LDS tier first, its bails
    to the general thread-per-pair kernel, DP leftovers to the wave-per-pair tier when they are few), with the compact
    tier off (every pair through the general thread-per-pair kernel), with the compact tier off and the DP service +
    replay rounds forced for any number of leftovers (PMX_ALIGN_TPP_MIN=0), and with every pair on the wave-per-pair
    kernels (PMX_ALIGN_NO_TPP=1): the exact-parity cases cover all execution models and both DP kernels on the GPU."""
    if request.param == "compact_off":
        monkeypatch.setenv("PMX_ALIGN_NO_COMPACT", "1")
    elif request.param == "dp_service_forced":
        monkeypatch.setenv("PMX_ALIGN_NO_COMPACT", "1")
        monkeypatch.setenv("PMX_ALIGN_TPP_MIN", "0")
    elif request.param == "wave_per_pair_only":
        monkeypatch.setenv("PMX_ALIGN_NO_TPP", "1")
    yield


def _ref_genome():
    return b"".join(l.strip() for l in open(os.path.join(GOLDEN, "isolate.ref.fa"), "rb") if not l.startswith(b">"))


def _pairs(pmx, genome, n, seed, **kw):
    concat, off = pmx.simulate_paired_reads(genome, n, seed=seed, **kw)
    reads = [bytes(concat[off[i]:off[i + 1]]) for i in range(len(off) - 1)]
    return [r if i % 2 == 0 else pmx.reverse_complement(r) for i, r in enumerate(reads)]   # readFastqPaired orientation


def test_synthetic_pairs_exact(pmx, oracle, ctx):
    g = _ref_genome()
    reads = _pairs(pmx, g, 2000, 21)
    al = pmx.Aligner(ctx, g, 150)
    got = al.align_reads(reads, paired=True)
    want = oracle.ref_align_reads_direct(g, reads, True, 8)
    assert not ac.compare_results(got, want)
    assert all(x["flags"] & 3 == 0 for x in got)
    assert sum(w["mapped"] for w in want) == len(want)


def test_revcomp_on_device_equals_host_revcomp(pmx, oracle, ctx):
    g = _ref_genome()
    concat, off = pmx.simulate_paired_reads(g, 1000, seed=22)
    raw = [bytes(concat[off[i]:off[i + 1]]) for i in range(len(off) - 1)]       # FASTQ orientation
    host_rc = [r if i % 2 == 0 else pmx.reverse_complement(r) for i, r in enumerate(raw)]
    al = pmx.Aligner(ctx, g, 150)
    a = al.align_reads(raw, paired=True, revcomp_mate2=True)
    b = al.align_reads(host_rc, paired=True, revcomp_mate2=False)
    assert not ac.compare_results(a, b)


def test_example_reads_exact(pmx, oracle, ctx):
    """the README demo reads (real errors, indels, N, variable length) against the placed genome"""
    g = _ref_genome()
    seqs, _, _ = pmx.read_fastq_paired(os.path.join(GOLDEN, "isolate_R1.fastq.gz"), os.path.join(GOLDEN, "isolate_R2.fastq.gz"))
    reads = seqs                                   # all 102,338 reads (51,169 pairs; 41,003 map in the reference)
    mean_len = int(sum(len(r) for r in reads) / len(reads))
    al = pmx.Aligner(ctx, g, mean_len)
    got = al.align_reads(reads, paired=True)
    want = oracle.ref_align_reads_direct(g, reads, True, 8)
    bad = ac.compare_results(got, want)
    assert not bad, bad[:10]
    assert sum(w["mapped"] for w in want) == 41003
    assert sum(1 for x in got if x["flags"] & 3) == 0      # no capacity / unsupported-branch report on real reads


def test_noisy_and_edge_pairs(pmx, oracle, ctx):
    g = _ref_genome()
    reads = _pairs(pmx, g, 600, 23, sub_rate=0.03)
    rng = np.random.default_rng(5)
    # indels, N runs, unrelated sequence, very short mates
    for i in range(0, 400, 2):
        r = bytearray(reads[i])
        p = int(rng.integers(20, 120))
        k = i % 8
        if k == 0:
            del r[p:p + int(rng.integers(1, 9))]
        elif k == 2:
            r[p:p] = bytes(rng.choice(list(b"ACGT"), int(rng.integers(1, 9))).astype(np.uint8))
        elif k == 4:
            r[p:p + 5] = b"NNNNN"
        elif k == 6:
            r = bytearray(bytes(rng.choice(list(b"ACGT"), 150).astype(np.uint8)))
        reads[i] = bytes(r)
    reads[401] = reads[401][:40]
    reads[403] = reads[403][:21]
    reads[405] = b"A" * 150
    al = pmx.Aligner(ctx, g, 150)
    got = al.align_reads(reads, paired=True)
    want = oracle.ref_align_reads_direct(g, reads, True, 8)
    bad = ac.compare_results(got, want)
    assert not bad, bad[:10]


def test_varied_pair_shapes_exact(pmx, oracle, ctx, sars):
    """read length, insert size (mates overlapping down to one fragment shorter than a read, and far apart), substitution
    and indel rates, both mate orientations, three genomes of the tree: the compact tier's closed forms (colinear runs,
    distinct-window sketch) and every general tier against the reference aligner"""
    rng = np.random.default_rng(77)
    al = None
    for node, read_len, mean_insert, sub_rate, indel_every, as_sequenced in (
            ("node_7618", 150, 300.0, 0.002, 0, False), ("node_7618", 150, 180.0, 0.01, 0, False), ("node_7618", 100, 150.0, 0.02, 7, False),
            ("node_1", 75, 90.0, 0.005, 0, False), ("node_9000", 125, 500.0, 0.03, 5, False), ("node_7618", 150, 320.0, 0.004, 11, True),
            ("node_9000", 100, 130.0, 0.0, 0, True), ("node_1", 150, 700.0, 0.001, 0, False)):
        g = sars.genome(node)
        concat, off = pmx.simulate_paired_reads(g, 6000, read_len=read_len, seed=int(rng.integers(1, 1 << 30)), sub_rate=sub_rate,
                                                mean_insert=max(mean_insert, float(read_len)), sd_insert=mean_insert / 8)
        reads = [bytes(concat[off[i]:off[i + 1]]) for i in range(len(off) - 1)]
        if not as_sequenced:
            reads = [r if i % 2 == 0 else pmx.reverse_complement(r) for i, r in enumerate(reads)]
        if indel_every:
            for i in range(0, len(reads), indel_every):
                r = bytearray(reads[i])
                p = int(rng.integers(10, max(11, len(r) - 10)))
                if i % 2:
                    del r[p:p + int(rng.integers(1, 6))]
                else:
                    r[p:p] = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), int(rng.integers(1, 6))))
                reads[i] = bytes(r)
        if al is None:
            al = pmx.Aligner(ctx, g, read_len)
        else:
            al.set_reference(g, read_len)
        got = al.align_reads(reads, paired=True)
        want = oracle.ref_align_reads_direct(g, reads, True, 8)
        bad = ac.compare_results(got, want)
        assert not bad, (node, read_len, mean_insert, sub_rate, indel_every, as_sequenced, bad[:5])
        assert sum(1 for x in got if x["flags"] & 3) == 0


def test_align_reads_direct_dropin(pmx, oracle, ctx):
    """same signature, same results as the reference boundary (src/mm_align.h:44-53)"""
    g = _ref_genome()
    reads = _pairs(pmx, g, 300, 24)
    got = pmx.align_reads_direct(g, reads, True, 4)
    want = oracle.ref_align_reads_direct(g, reads, True, 4)
    assert not ac.compare_results(got, want)
    odd = reads[:7]                                             # odd count in paired mode: last read ignored
    assert not ac.compare_results(pmx.align_reads_direct(g, odd, True), oracle.ref_align_reads_direct(g, odd, True))


def test_direct_boundary_takes_pre_encoded_bases(pmx, oracle):
    """seq_nt4_table (sketch.c:9-26) maps the bytes 0..3 to the bases themselves: reads that carry them (a caller's 2-bit
    encoding) align at the drop-in boundary as they do in the reference"""
    g = _ref_genome()
    reads = _pairs(pmx, g, 400, 31)
    tr = bytes.maketrans(b"ACGT", bytes([0, 1, 2, 3]))
    mixed = [r.translate(tr) if i % 3 == 0 else (r[:40] + r[40:90].translate(tr) + r[90:]) if i % 3 == 1 else r for i, r in enumerate(reads)]
    want = oracle.ref_align_reads_direct(g, mixed, True, 8)
    assert oracle.ref_align_reads_direct(g, reads, True, 8) == want
    got = pmx.align_reads_direct(g, mixed, True)
    assert not ac.compare_results(got, want)


def test_direct_boundary_is_reentrant(pmx, oracle):
    """the reference calls the aligner concurrently from TBB workers in --batch mode (src/main.cpp:1581-1611): two
    host threads inside pmx_align_reads_direct at the same time, on different read sets and references, must each
    get what a serial call gives (the thread-per-pair arena travels in the kernel arguments, not in a module global)"""
    import threading
    g = _ref_genome()
    g2 = g[2000:20000]
    sets = [(g, _pairs(pmx, g, 30000, 31)), (g2, _pairs(pmx, g2, 30000, 32)), (g, _pairs(pmx, g, 20000, 33, sub_rate=0.02))]
    serial = [pmx.align_reads_direct(ref, reads, True, 1) for ref, reads in sets]
    for w, (ref, reads) in zip(serial, sets):
        assert not ac.compare_results(w[:2000], oracle.ref_align_reads_direct(ref, reads[:4000], True, 8))
    out = [None] * len(sets)

    def work(i):
        out[i] = pmx.align_reads_direct(sets[i][0], sets[i][1], True, 1)
    for _ in range(2):
        ths = [threading.Thread(target=work, args=(i,)) for i in range(len(sets))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        for i in range(len(sets)):
            assert out[i] is not None and not ac.compare_results(out[i], serial[i]), i
            assert [x["r1"]["cigar"] for x in out[i]] == [x["r1"]["cigar"] for x in serial[i]]


def test_cigar_arena_overflow_is_redone(pmx, oracle, ctx, monkeypatch):
    """the CIGAR arena is sized optimistically; a call that overflows it is redone with the counted size, so no
    record leaves the library flagged because of the arena (api_align.hip:474)"""
    g = _ref_genome()
    reads = _pairs(pmx, g, 3000, 41)
    want = oracle.ref_align_reads_direct(g, reads, True, 8)
    monkeypatch.setenv("PMX_ALIGN_CIGAR_CAP", "100")
    al = pmx.Aligner(ctx, g, 150)
    got = al.align_reads(reads, paired=True)
    assert not ac.compare_results(got, want)
    assert all(x["flags"] & 3 == 0 for x in got)
    assert not ac.compare_results(pmx.align_reads_direct(g, reads, True, 1), want)


def test_invalid_records_never_escape_the_boundary(pmx, oracle, ctx, monkeypatch):
    """a record flagged OVERFLOW / UNSUPPORTED is invalid: forced here by capping the CIGAR operations per region in
    every tier (test hook), so that reads with indels overflow.  The drop-in must report those pairs unmapped (and say
    so through pmx_last_error), never hand out their half-built CIGARs; all other pairs stay exact."""
    g = _ref_genome()
    reads = _pairs(pmx, g, 400, 51)
    rng = np.random.default_rng(3)
    for i in range(0, 200, 2):          # a deletion and an insertion in the first 100 R1 mates: CIGARs of >= 5 operations
        r = bytearray(reads[i])
        del r[40:40 + int(rng.integers(2, 5))]
        r[100:100] = b"ACGTTGCA"[:int(rng.integers(2, 6))]
        reads[i] = bytes(r)
    want = oracle.ref_align_reads_direct(g, reads, True, 8)
    n_gapped = sum(1 for w in want if w["mapped"] and (len(w["r1"]["cigar"]) > 2 or len(w["r2"]["cigar"]) > 2))
    assert n_gapped >= 50
    monkeypatch.setenv("PMX_ALIGN_TEST_MAX_CIGAR", "2")
    got = pmx.align_reads_direct(g, reads, True, 1)
    assert b"withheld" in pmx.last_error()
    n_withheld = 0
    for gt, w in zip(got, want):
        gapped = w["mapped"] and (len(w["r1"]["cigar"]) > 2 or len(w["r2"]["cigar"]) > 2)
        if gapped:
            assert gt["mapped"] == 0 and gt["r1"]["cigar"] == [] and gt["r1"]["pos"] == 2147483647
            n_withheld += 1
        elif gt["mapped"] == 0 and w["mapped"]:
            n_withheld += 1             # a pair whose intermediate CIGAR overflowed although the final one is short
        else:
            assert not ac.compare_results([gt], [w])
    assert n_withheld >= n_gapped
    al = pmx.Aligner(ctx, g, 150)
    res = al.align_reads(reads, paired=True)
    assert all((x["flags"] & 3 == 0) or (x["mapped"] == 0 and x["r1"]["cigar"] == []) for x in res)


def test_full_size_properties(pmx, ctx):
    """BASELINE config 2 (1M reads): error-free pairs must come back as full-length matches at their
    source coordinates with proper_frag set; mate order / strand consistent."""
    g = _ref_genome()
    n_pairs = 500000
    concat, off = pmx.simulate_paired_reads(g, n_pairs, seed=42, sub_rate=0.0)
    rs = pmx.ReadSet(ctx, concat=concat, offsets=off)
    al = pmx.Aligner(ctx, g, 150)
    al.align_readset(rs, paired=True, revcomp_mate2=True)
    recs, cig = al.fetch()
    assert len(recs) == 2 * n_pairs
    assert np.all(recs["mapped"] == 1) and np.all(recs["flags"] & 3 == 0)
    # every alignment is one match run whose reference span equals its query span; nearly all are full length
    # (the reference itself clips a handful of error-free mates, e.g. 111M, when mates overlap)
    span = recs["re"] - recs["rs"]
    assert np.all(recs["n_cigar"] == 1) and np.all(span == recs["qe"] - recs["qs"])
    assert np.all(cig[recs["cigar_off"]] == (span.astype(np.uint32) << 4))
    assert np.mean(span == 150) > 0.999
    assert np.mean(recs["proper_frag"] == 1) > 0.999
    assert np.all(recs["rev"] == 0)                 # R2 was reverse-complemented into the forward strand
    # the aligned query interval equals the reference interval base for base (sampled)
    ga = np.frombuffer(g, np.uint8)
    comp = np.zeros(256, np.uint8)
    for a_, b_ in zip(b"ACGT", b"TGCA"):
        comp[a_] = b_
    rng = np.random.default_rng(0)
    for i in rng.integers(0, len(recs), 3000):
        r = recs[i]
        read = concat[off[i]:off[i + 1]]
        if i & 1:
            read = comp[read[::-1]]
        assert np.array_equal(read[r["qs"]:r["qe"]], ga[r["rs"]:r["re"]])


def test_golden_fixture_gpu(pmx, ctx):
    """HIP path vs the committed outputs of the reference aligner (tests/golden/align_golden.json.gz)"""
    g, cases_ = ac.golden_cases(pmx)
    for name, (reads, want) in cases_.items():
        mean_len = int(sum(len(r) for r in reads) / len(reads))
        al = pmx.Aligner(ctx, g, mean_len)
        got = al.align_reads(reads, paired=True)
        bad = ac.compare_results(got, want)
        assert not bad, (name, bad[:10])
        assert sum(1 for x in got if x["flags"] & 3) <= 2



@pytest.mark.gpu
def test_device_and_host_reference_index_are_the_same_index(pmx, sars, monkeypatch):
    """mm_idx_str (index.c:408-451) twice: on the device from 64-base slices (ref_index_kernels.hip) and by the host
    restatement (build_ref_index): same minimizers, same occurrence lists in the same order, same mid_occ -- on real
    genomes, on references with N runs / homopolymers / tandem repeats, and for the long-read preset (mid_occ taken
    from the index); a repeat-rich reference makes the device build hand over to the host build"""
    ctx = pmx.Context(0)
    rng = np.random.default_rng(17)

    def rand_ref(n, junk=True):
        s = rng.choice(np.frombuffer(b"ACGT", np.uint8), n)
        if junk:
            for _ in range(n // 300 + 1):
                p = int(rng.integers(0, n)); t = int(rng.integers(0, 4)); ln = int(rng.integers(1, 80))
                if t == 0:
                    s[p:p + ln] = ord("N")
                elif t == 1:
                    s[p:p + ln] = s[p]
                elif t == 2:
                    u = s[p:p + int(rng.integers(2, 7))].copy()
                    rep = np.tile(u, 60)[:3 * ln]
                    s[p:p + len(rep)] = rep[:len(s[p:p + len(rep)])]
                else:
                    s[p] = ord("n")
        return bytes(s)
    rsv = pmx.Panman(os.path.join(os.path.dirname(__file__), "golden", "rsv_4K.panman"))
    refs = [sars.genome("node_7618"), sars.genome(0), rsv.genome("MZ515733.1"), rand_ref(70), rand_ref(1000), rand_ref(30000), rand_ref(200000),
            b"ACGT" * 40, rand_ref(5000, junk=False).lower()]
    al = pmx.Aligner(ctx, refs[0], 150)
    for ref in refs:
        for mean_len in (150, 3000):          # short-read preset (mid_occ fixed) and map-ont (mid_occ from the index)
            monkeypatch.delenv("PMX_ALIGN_HOST_INDEX", raising=False)
            al.set_reference(ref, mean_len)
            dev = al.index_digest()
            monkeypatch.setenv("PMX_ALIGN_HOST_INDEX", "1")
            al.set_reference(ref, mean_len)
            host = al.index_digest()
            assert host[4] == 0
            assert dev[:4] == host[:4], (len(ref), mean_len, dev, host)
            if mean_len == 150:
                assert dev[4] == 1, len(ref)       # (the long-read preset may hand a repeat-rich reference to the host build)
    # forty copies of one 500-base unit: every minimizer occurs forty times, the mid_occ quantile is above the clamp
    monkeypatch.delenv("PMX_ALIGN_HOST_INDEX", raising=False)
    unit = rand_ref(500, junk=False)
    al.set_reference(unit * 40, 3000)
    rep = al.index_digest()
    assert rep[4] == 0 and rep[2] > 10
