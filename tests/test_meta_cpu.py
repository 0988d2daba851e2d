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
