"""CPU suite: the oracle (oracle/oracle_place.c) against the reference's golden vectors and its own
unit-test contracts (src/test/test_seeding.cpp, src/test/test_placement.cpp)."""
import json
import math
import os
import random

import numpy as np

from conftest import GOLDEN


def _rand_dna(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n)).encode()


def test_known_answers_from_compiled_reference(oracle):
    ka = json.load(open(os.path.join(GOLDEN, "seeding_known_answers.json")))
    syn = [x for x in oracle.rolling_syncmers(ka["sequence"].encode(), 19, 8, False, 0, False)]
    assert len(syn) == ka["n_syncmers"]
    for got, want in zip(syn[:3], ka["first"]):
        assert ("%016x" % got[0], got[1], got[3]) == tuple(want)
    assert ("%016x" % syn[-1][0], syn[-1][1], syn[-1][3]) == tuple(ka["last"])
    f, r = oracle.hash_seq(ka["hashSeq"]["kmer"].encode())
    assert "%016x" % f == ka["hashSeq"]["fwd"] and "%016x" % r == ka["hashSeq"]["rev"]


def test_hashseq_canonical_strand_invariance(oracle, pmx):
    # src/test/test_seeding.cpp:20-35
    rng = random.Random(1234)
    for _ in range(50):
        s = _rand_dna(rng, 21)
        a = oracle.hash_seq(s)
        b = oracle.hash_seq(pmx.reverse_complement(s))
        assert min(a) == min(b)


def test_rollingsyncmers_contract(oracle):
    # src/test/test_seeding.cpp:37-82
    rng = random.Random(99)
    for _ in range(10):
        seq = _rand_dna(rng, 200)
        for k in (15, 19, 31):
            for s in (6, 8):
                allk = oracle.rolling_syncmers(seq, k, s, False, 0, True)
                assert len(allk) == len(seq) - k + 1
                for h, isrev, issync, pos in allk:
                    assert 0 <= pos and pos + k <= len(seq)
                    if h != 0xFFFFFFFFFFFFFFFF:
                        assert h == min(oracle.hash_seq(seq[pos:pos + k]))
                only = oracle.rolling_syncmers(seq, k, s, False, 0, False)
                assert [x for x in allk if x[2]] == only
                assert len(only) > 0


def test_syncmers_edge_cases(oracle):
    assert oracle.rolling_syncmers(b"ACGT", 19, 8, False, 0, True) == []          # shorter than k
    seq = b"ACGTTGCATGCCGATAGCTAGNTAGGATCGATCGATTTAGCGCGATATAGCGC"
    for h, _, s, pos in oracle.rolling_syncmers(seq, 19, 8, False, 0, False):
        assert b"N" not in seq[pos:pos + 19]
    low = oracle.rolling_syncmers(seq.lower(), 19, 8, False, 0, False)
    assert low == oracle.rolling_syncmers(seq, 19, 8, False, 0, False)             # case-insensitive chash
    # open syncmers with offset t are a subset rule of their own; just exercise determinism
    a = oracle.rolling_syncmers(seq * 3, 15, 6, True, 2, False)
    assert a == oracle.rolling_syncmers(seq * 3, 15, 6, True, 2, False)


def test_read_seeds_against_definition(oracle):
    rng = random.Random(5)
    rotl = lambda h, r: ((h << (r & 63)) | (h >> (64 - (r & 63)))) & 0xFFFFFFFFFFFFFFFF if r & 63 else h
    for k, s, l in ((19, 8, 3), (15, 6, 2), (21, 8, 1), (31, 6, 3)):
        seq = _rand_dna(rng, 260)
        syn = [x for x in oracle.rolling_syncmers(seq, k, s, False, 0, False)]
        hs = [x[0] for x in syn]
        want = []
        if l == 1:
            want = hs
        else:
            for j in range(len(hs) - l + 1):
                F = R = 0
                for q in range(l):
                    F ^= rotl(hs[j + q], k * (l - 1 - q))
                    R ^= rotl(hs[j + q], k * q)
                if F != R:
                    want.append(min(F, R))
        assert list(oracle.read_seeds(seq, k, s, l)) == want
        # trimming keeps syncmers with start in [trimStart, len-trimEnd-k]
        ts, te = 17, 23
        hs2 = [x[0] for x in syn if ts <= x[3] <= len(seq) - te - k]
        if l == 1:
            assert list(oracle.read_seeds(seq, k, s, l, False, 0, ts, te)) == hs2


def _hand_index(changes):
    """single-node index: root with the given (hash,parent,child) changes."""
    h = np.array([c[0] for c in changes], np.uint64)
    pc = np.array([c[1] for c in changes], np.int16)
    cc = np.array([c[2] for c in changes], np.int16)
    return dict(parent=np.zeros(1, np.uint32), offsets=np.array([0, len(changes)], np.uint64), hash=h, parent_count=pc, child_count=cc)


def test_compute_child_metrics_hand_values(oracle):
    # src/test/test_placement.cpp:88-142: one seed, read count r=3, genome count g=2
    kept_hash = np.array([42], np.uint64)
    L = math.log1p(3.0)
    st = oracle.ReadState(1, 1, 1, 3, math.sqrt(L * L), L, 0.0)
    idx = _hand_index([(42, 0, 2)])
    sc, met, cts, wc = oracle.score_nodes(idx["parent"], idx["offsets"], idx["hash"], idx["parent_count"], idx["child_count"],
                                          kept_hash, np.array([L]), st)
    assert abs(sc[0, 0] - 0.5) < 1e-12          # logRaw  = (L/2)/|r|
    assert abs(sc[0, 1] - 1.0) < 1e-12          # cosine
    assert abs(sc[0, 2] - 1.0) < 1e-12          # containment 1/1
    assert abs(wc - 0.5) < 1e-15 and abs(sc[0, 3] - 1.0) < 1e-12
    assert abs(sc[0, 4] - 1.0) < 1e-12
    assert cts[0, 0] == 1 and cts[0, 1] == 1


def test_two_seed_case_and_live_delta_equals_scratch(oracle):
    # src/test/test_placement.cpp:144-241: accumulated deltas along a path == from-scratch metrics
    rng = np.random.default_rng(3)
    hashes = np.sort(rng.integers(1, 2 ** 62, 40, dtype=np.uint64))
    counts = rng.integers(2, 9, 40)
    kept_hash = hashes[:25]
    kept_log = np.log1p(counts[:25].astype(np.float64))
    st = oracle.ReadState(2, 25, 25, int(counts[:25].sum()), float(np.sqrt((kept_log ** 2).sum())), float(kept_log.sum()), 0.0)
    # root has seeds 0..29 with count 1..3; child changes some
    root = [(int(h), 0, int(1 + i % 3)) for i, h in enumerate(hashes[:30])]
    child = [(int(hashes[i]), int(1 + i % 3), int((1 + i % 3 + 1) % 4)) for i in range(5, 20, 3)] + [(int(hashes[35]), 0, 2)]
    child.sort()
    idx = dict(parent=np.array([0, 0], np.uint32), offsets=np.array([0, len(root), len(root) + len(child)], np.uint64),
               hash=np.array([c[0] for c in root + child], np.uint64), parent_count=np.array([c[1] for c in root + child], np.int16),
               child_count=np.array([c[2] for c in root + child], np.int16))
    sc, met, cts, wc = oracle.score_nodes(idx["parent"], idx["offsets"], idx["hash"], idx["parent_count"], idx["child_count"], kept_hash, kept_log, st)
    genome = {c[0]: c[2] for c in root}
    for c in child:
        genome[c[0]] = c[2]
    genome = {h: g for h, g in genome.items() if g > 0}
    logmap = dict(zip(kept_hash.tolist(), kept_log.tolist()))
    raw = sum(logmap[h] / g for h, g in genome.items() if h in logmap)
    cos = sum(logmap[h] * math.log1p(g) for h, g in genome.items() if h in logmap)
    gm2 = sum(math.log1p(g) ** 2 for g in genome.values())
    pres = sum(1 for h in genome if h in logmap)
    assert abs(met[1, 0] - raw) < 1e-9 and abs(met[1, 1] - cos) < 1e-9 and abs(met[1, 4] - gm2) < 1e-9
    assert cts[1, 0] == pres and cts[1, 1] == len(genome)


def test_min_read_support_and_magnitudes(oracle):
    # src/test/test_placement.cpp:243-296
    hs = np.arange(1, 11, dtype=np.uint64) * 1000
    cn = np.array([1, 1, 1, 5, 5, 5, 5, 1, 1, 1], np.int64)
    kh, kl, st = oracle.finalize_reads(hs, cn, 19)            # auto: mean over counts>=2 is 5 > 3 -> 2
    assert st.min_support == 2 and list(kh) == [4000, 5000, 6000, 7000]
    assert abs(st.log_magnitude - math.sqrt(4 * math.log1p(5) ** 2)) < 1e-12
    cn2 = np.array([1, 1, 1, 2, 3, 2, 3, 1, 1, 1], np.int64)
    _, _, st2 = oracle.finalize_reads(hs, cn2, 19)            # mean 2.5 <= 3 -> keep singletons
    assert st2.min_support == 1 and st2.n_kept == 10
    _, _, st3 = oracle.finalize_reads(hs, cn, 19, 0.0, 1)     # explicit
    assert st3.min_support == 1 and st3.n_kept == 10


def test_best_tie_rule(oracle):
    # src/placement.cpp:355-401: first score of the top cluster is kept, ties within 1e-4 relative
    parent = np.array([0, 0, 0, 1, 1], np.uint32)
    sc = np.zeros((5, 5))
    sc[:, 0] = [1.0, 2.0, 2.00001, 1.99999, 0.5]
    best, idx, ties = oracle.best_ties(parent, sc)
    assert best[0] == 2.0 and list(ties[0]) == [1, 2, 3] and idx[0] == 1
    best, idx, ties = oracle.best_ties(parent, sc, force_leaf=True)   # leaves: 2,3,4
    assert best[0] == 2.00001 and list(ties[0]) == [2, 3]
    assert idx[1] == 0xFFFFFFFF and len(ties[1]) == 0                  # all-zero metric -> no placement


def test_golden_placement_tsv_end_to_end(oracle, pmx, sars, sars_index, isolate_reads):
    """examples/check_examples.sh:52-70: byte-exact isolate.placement.tsv (oracle + host index stage)."""
    ka = json.load(open(os.path.join(GOLDEN, "seeding_known_answers.json")))["example_reads"]
    assert len(isolate_reads) == ka["n_reads"]
    out = oracle.place(isolate_reads, sars_index.arrays(), 19, 8, 3)
    st = out["state"]
    assert len(out["hist_hash"]) == ka["unique_seeds"] and st.n_kept == ka["kept_seeds"] and st.min_support == ka["min_support"]
    assert "%.2f" % st.est_coverage == ka["est_coverage_2dp"]
    assert "%.6f" % st.log_magnitude == ka["log_magnitude_6dp"] and "%.6f" % st.log_cont_den == ka["sum_log1p_6dp"]
    res = pmx.PlacementResult(list(out["best"]), list(out["best_idx"]), out["ties"])
    txt = pmx.format_placement_tsv(res, sars.node_id)
    assert txt == open(os.path.join(GOLDEN, "isolate.placement.tsv")).read()
