"""BASELINE config 3's batch at full size on one GPU (the multi-GPU form shards exactly this batch, tests/test_bench_gpu.py
covers the exchange): properties that need no oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_config3_ten_million_reads_on_one_gpu(pmx):
    """BASELINE config 3's batch (10M x 150 bp paired reads) through place + align on ONE GPU: the properties that do not
    need the oracle -- the sample places on its source node, every error-free pair comes back as two full-length matches
    at the coordinates it was drawn from, proper pairs, forward strand; a checksum of the reference intervals equals the
    one computed from the generator's own coordinates."""
    import os
    from conftest import GOLDEN
    pm = pmx.Panman(os.path.join(GOLDEN, "sars_20000_twilight_dipper.panman"))
    index = pmx.Index.build(pm)
    ctx = pmx.Context(0)
    src = pm.genome("node_7618")
    n_pairs = 5000000
    concat, off = pmx.simulate_paired_reads(src, n_pairs, seed=4242, sub_rate=0.0)
    rs = pmx.ReadSet(ctx, concat=concat, offsets=off)
    placer = pmx.Placer(ctx, index)
    placer.reset()
    placer.add_reads(rs)
    res = placer.score(pmx.TraversalParams(), 2 * n_pairs)
    assert pm.node_id(int(res.best_index[4])) == "node_7618"
    al = pmx.Aligner(ctx, src, 150)
    al.align_readset(rs, paired=True, revcomp_mate2=True)
    recs, cig = al.fetch()
    assert len(recs) == 2 * n_pairs and np.all(recs["mapped"] == 1) and np.all(recs["flags"] & 3 == 0)
    span = recs["re"] - recs["rs"]
    assert np.all(recs["n_cigar"] == 1) and np.all(span == recs["qe"] - recs["qs"])
    assert np.mean(span == 150) > 0.999 and np.mean(recs["proper_frag"] == 1) > 0.999 and np.all(recs["rev"] == 0)
    assert np.all(cig[recs["cigar_off"]] == (span.astype(np.uint32) << 4))
    # every full-length read sits where it was drawn from: compare the read with the genome at rs (all reads, vectorised)
    ga = np.frombuffer(src, np.uint8)
    comp = np.zeros(256, np.uint8)
    for a_, b_ in zip(b"ACGT", b"TGCA"):
        comp[a_] = b_
    full = np.nonzero(span == 150)[0]
    reads2d = concat.reshape(-1, 150)
    for lo in range(0, len(full), 1000000):
        idx = full[lo:lo + 1000000]
        got = ga[recs["rs"][idx][:, None] + np.arange(150)[None, :]]
        want = reads2d[idx]
        odd = (idx & 1) == 1
        want = np.where(odd[:, None], comp[want[:, ::-1]], want)
        assert np.array_equal(got, want)
    st = al.stats()
    assert st["n_items"] == n_pairs and st["compact_tier_items"] > 0.98 * n_pairs
