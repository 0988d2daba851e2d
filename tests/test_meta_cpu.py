"""--meta, CPU part: the oriented seed index (pmx_index_build_ex | PMX_INDEX_ORIENTED, host C++) against the oracle's
from-the-genome-string seedmer counts (oracle/oracle_meta.py on top of the rollingSyncmers restatement that the compiled
reference's known answers pin) -- the node side of the scores is then pinned without any product code in the checker."""
import os

import numpy as np

from conftest import GOLDEN


def test_oriented_index_equals_seedmer_counts_of_the_genomes(pmx):
    from oracle import oracle_meta as om
    pm = pmx.Panman(os.path.join(GOLDEN, "rsv_4K.panman"))
    oidx = pmx.Index.build(pm, mode=0x100, flank_mask=0)
    arr = oidx.arrays()
    rng = np.random.default_rng(2)
    nodes = [pm.find_node("MZ515733.1"), pm.find_node("node_1330")] + [int(x) for x in rng.integers(1, oidx.info.n_nodes, 4)]
    for node in nodes:
        want = om.genome_seed_counts(pm.genome(node), 19, 8, 3)
        keyed = {}
        for h, (f, r) in want.items():                            # the index keys a reverse seedmer as hash ^ ORIENT_XOR
            if f:
                keyed[h] = f
            if r:
                keyed[h ^ om.ORIENT_XOR] = r
        assert om.node_seed_counts(arr, node) == keyed, node
    # the unoriented index of the same tree folds the two orientations together
    uidx = pmx.Index.build(pm, flank_mask=0).arrays()
    for node in nodes[:3]:
        want = om.genome_seed_counts(pm.genome(node), 19, 8, 3)
        assert om.node_seed_counts(uidx, node) == {h: f + r for h, (f, r) in want.items()}


def test_seedmers_are_strand_symmetric():
    """the reverse complement of a sequence carries the same seedmer hashes, each in the other orientation, in reverse order"""
    from oracle import oracle_meta as om
    rng = np.random.default_rng(7)
    seq = bytes(rng.choice(list(b"ACGT"), 400).astype(np.uint8))
    comp = {65: 84, 67: 71, 71: 67, 84: 65}
    rc = bytes(comp[c] for c in reversed(seq))
    a, b = om.seedmers(seq, 19, 8, 3), om.seedmers(rc, 19, 8, 3)
    assert len(a) > 20 and [(h, not r) for h, r in reversed(a)] == b


def test_dust_score_contract_and_product_equals_restatement(pmx):
    """mgsr::getDust: the reference's own unit-test contract (src/test/test_mgsr.cpp:12-29 -- deterministic; a low-complexity
    sequence scores strictly higher than a varied one; "", "ACG" and 50 x N do not crash) held by the restatement
    (oracle_meta.get_dust, src/mgsr.cpp:1505-1568) and by the library's pmx_read_dust, and the two agree bit for bit on
    random reads with N, lower case, runs and windows other than 64"""
    from oracle import oracle_meta as om
    low = b"A" * 200
    high = bytes(b"ACGT"[(i * 7 + (i // 4)) % 4] for i in range(200))
    for f in (om.get_dust, pmx.read_dust):
        assert f(low) == f(low)
        assert f(low) > f(high)
        assert f(b"") == 0.0 and f(b"ACG") == 0.0 and f(b"N" * 50) == 0.0
    assert om.get_dust(low) == 100.0 and pmx.read_dust(low) == 100.0      # 64 equal triplets: 2,016 pairs, 200 * 2016 / (64 * 63)
    rng = np.random.default_rng(11)
    for i in range(400):
        n = int(rng.integers(0, 260))
        alphabet = list(b"ACGTNacgt") if i % 3 else list(b"AAAAACGT")
        seq = bytes(rng.choice(alphabet, n).astype(np.uint8))
        for w in (64, 3, 17):
            assert pmx.read_dust(seq, w) == om.get_dust(seq, w), (seq, w)


def test_vectorised_read_scores_equal_the_loop():
    from oracle import oracle_meta as om
    rng = np.random.default_rng(3)
    hashes = rng.integers(1, 1 << 62, 300, dtype=np.uint64)
    counts = {int(h): [int(rng.integers(0, 3)), int(rng.integers(0, 3))] for h in hashes[:200]}
    ns = rng.integers(0, 40, 150)
    off = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
    sh = rng.choice(hashes, int(off[-1]))
    rev = rng.integers(0, 2, int(off[-1])).astype(np.uint8)
    assert np.array_equal(om.read_scores(counts, off, sh, rev), om.read_scores_np(counts, off, sh, rev))
    assert np.array_equal(om.read_scores({}, off, sh, rev), om.read_scores_np({}, off, sh, rev))


def test_discard_rule_truncates_the_threshold():
    """src/main.cpp:1229-1240: maxScore < static_cast<int>(seedmers * discard) -- 37 seedmers at 0.5: threshold 18, so a best
    score of 18 stays (18.5 would drop it)"""
    from oracle import oracle_meta as om
    assert om.discard_rows([18, 17, 0, 19], [37, 37, 37, 37], 0.5).tolist() == [True, False, False, True]
