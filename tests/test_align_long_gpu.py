"""GPU parity tests for long reads (BASELINE config 4: single-end 10 kb reads with substitutions and indels, the
map-hifi / map-ont branches of setup_minimap2, src/mm_align.c:167-180): the wave-per-read general tier against the
committed outputs of the reference aligner (tests/golden/align_golden_long.json.gz, made by
tests/golden/make_align_golden_long.py from oracle/_ref) and, when the compiled reference is present, against a
live run of it.  Bit-exact on pos / rs / re / qs / qe / mapq / rev / CIGAR; no flagged record."""
import gzip
import json
import os
import sys

import numpy as np
import pytest

import align_checks as ac
from conftest import GOLDEN

pytestmark = pytest.mark.gpu

sys.path.insert(0, GOLDEN)
import make_align_golden_long as mg  # noqa: E402


def _expected():
    return json.loads(gzip.open(os.path.join(GOLDEN, "align_golden_long.json.gz")).read())


@pytest.mark.parametrize("name", list(mg.SETS))
def test_long_reads_equal_reference_fixture(pmx, ctx, name):
    g = mg.genome()
    reads = mg.inputs(pmx, name)
    al = pmx.Aligner(ctx, g, int(np.mean([len(r) for r in reads])))
    got = al.align_reads(reads, paired=False)
    assert sum(1 for x in got if x["flags"] & 3) == 0
    exp = _expected()[name]
    assert len(got) == len(exp)
    bad = []
    for i, (x, row) in enumerate(zip(got, exp)):
        r = x["r1"]
        mine = [int(x["mapped"]), r["pos"], r["rs"], r["re"], r["qs"], r["qe"], r["mapq"], r["rev"], len(r["cigar"]), mg.cigar_crc(r["cigar"])]
        if mine != row[:10] or (len(row) > 10 and r["cigar"] != row[10]):
            bad.append((i, mine, row[:10]))
    assert not bad, bad[:5]
    st = al.stats()
    assert st["dp_calls"] > 10 * len(reads) and st["dp_cells"] > 0      # long reads are DP work, not shortcuts


def test_long_reads_live_reference_and_small_traceback_tier(pmx, oracle, ctx, monkeypatch):
    """a fresh set against the compiled reference; then the same with a 64 KB traceback area in the first launch, so that
    reads are re-run by the full-capacity launch: same records"""
    g = mg.genome()
    reads = pmx.simulate_long_reads(g, 300, read_len=6000, seed=91)
    want = oracle.ref_align_reads_direct(g, reads, False, 8)
    al = pmx.Aligner(ctx, g, 6000)
    got = al.align_reads(reads, paired=False)
    assert not ac.compare_results(got, want) and all(x["flags"] & 3 == 0 for x in got)
    monkeypatch.setenv("PMX_ALIGN_TB_KB", "64")    # (64 KB: a 300 x 300 DP no longer fits, its read goes to the second launch)
    got2 = al.align_reads(reads, paired=False)
    assert not ac.compare_results(got2, want) and all(x["flags"] & 3 == 0 for x in got2)
    assert al.stats()["general_tier_items"] > 0


@pytest.mark.parametrize("read_len,sub,seed", [(6000, 0.03, 11), (9000, 0.05, 12), (2500, 0.05, 13), (14000, 0.02, 14), (1500, 0.03, 15)])
def test_rearranged_long_reads_equal_reference(pmx, oracle, ctx, read_len, sub, seed):
    """reads with a deletion / insertion / inversion / duplication each: the RMQ re-chaining pass, the inversion probe and
    inversion hits, the divergence filter (both long-read presets) -- exact, and nothing flagged"""
    g = mg.genome()
    reads = ac.rearranged_long_reads(pmx, g, 150, read_len, seed, sub=sub)
    want = oracle.ref_align_reads_direct(g, reads, False, 8)
    al = pmx.Aligner(ctx, g, int(np.mean([len(r) for r in reads])))
    got = al.align_reads(reads, paired=False)
    assert not ac.compare_results(got, want)
    assert sum(1 for x in got if x["flags"] & 3) == 0


@pytest.mark.parametrize("copy_div,read_err,copy,read_len", [(0.03, 0.01, (3000, 23000), None), (0.003, 0.003, (7000, 18000), 15000)])
def test_strand_retained_secondaries_equal_reference(pmx, oracle, ctx, copy_div, read_err, copy, read_len):
    """a reference with a diverged inverted copy of 20 kb: every read has an opposite-strand secondary that mm_select_sub keeps
    as strand_retained; mm_est_err (with the device's pow) and mm_filter_strand_retained decide whether it stays"""
    ref, reads = ac.inverted_repeat_case(pmx, mg.genome(), 100, 41, copy_div, read_err, copy=copy, read_len=read_len)
    want = oracle.ref_align_reads_direct(ref, reads, False, 8)
    al = pmx.Aligner(ctx, ref, int(np.mean([len(r) for r in reads])))
    got = al.align_reads(reads, paired=False)
    assert not ac.compare_results(got, want)
    assert sum(1 for x in got if x["flags"] & 3) == 0


def test_slab_budget_caps_the_grid_not_the_results(pmx, ctx, monkeypatch):
    """every resident wave of the wave-per-read tier owns a ~20 MB slab; the launch takes the grid its memory budget pays for
    (16,000 reads of 10 kb used to ask for 318 GB).  With the budget forced down to 100 MB -- a handful of waves -- the
    records are the same"""
    g = mg.genome()
    reads = pmx.simulate_long_reads(g, 120, read_len=6000, seed=77)
    al = pmx.Aligner(ctx, g, 6000)
    want = al.align_reads(reads, paired=False)
    monkeypatch.setenv("PMX_ALIGN_SLAB_MB", "100")
    al2 = pmx.Aligner(ctx, g, 6000)
    got = al2.align_reads(reads, paired=False)
    assert not ac.compare_results(got, want) and all(x["flags"] & 3 == 0 for x in got)


def test_config4_full_size_properties(pmx, ctx):
    """BASELINE config 4 at full size (100k x 10 kb): every read maps, CIGARs are consistent with the reported intervals,
    alignments cover the read and agree with where the read was drawn from"""
    g = mg.genome()
    n = 100000
    rng = np.random.default_rng(4)
    reads = pmx.simulate_long_reads(g, n, read_len=10000, seed=143)
    al = pmx.Aligner(ctx, g, 10000)
    rs = pmx.ReadSet(ctx, reads)
    al.align_readset(rs, paired=False)
    recs, cig = al.fetch()
    assert len(recs) == n and np.all(recs["flags"] & 3 == 0)
    assert np.mean(recs["mapped"]) > 0.9999
    m = recs[recs["mapped"] == 1]
    lens = np.array([len(r) for r in reads])[recs["mapped"] == 1]
    assert np.all(m["qe"] - m["qs"] > 0.95 * lens) and np.all((m["re"] - m["rs"] > 9000) & (m["re"] - m["rs"] < 11000))
    assert np.mean(m["mapq"] == 60) > 0.99
    assert 0.45 < np.mean(m["rev"]) < 0.55
    for i in rng.integers(0, len(m), 400):      # CIGAR lengths add up to the intervals
        r = m[i]
        ops = cig[r["cigar_off"]:r["cigar_off"] + r["n_cigar"]]
        ln, op = ops >> 4, ops & 0xf
        assert set(op.tolist()) <= {0, 1, 2}
        assert int(ln[(op == 0) | (op == 1)].sum()) == r["qe"] - r["qs"] and int(ln[(op == 0) | (op == 2)].sum()) == r["re"] - r["rs"]
