"""GPU parity of the compact align tier's other compiled forms against the reference's own aligner (oracle/_ref): the
fused kernel (sketch + probes inside, PMX_ALIGN_COMPACT_FUSED), the 32-bit position layout (references longer than
32,767 bases: k_compact_seeds32 / k_align_compact32, forced on the 29.9 kb genome by PMX_ALIGN_COMPACT_POS32), both
together, and the resident strided grids (PMX_ALIGN_COMPACT_WAVES / PMX_ALIGN_CSEED_WAVES).  The default form -- two
kernels, 16-bit position words, one workgroup per 64 pairs -- is what every test of test_align_gpu.py runs.
(Added at the end of round 3 when the GPU budget was spent: the file sorts last so that the driver's run of the suite
reaches every other test first.  The same sources pass in all these forms in the host build, tests/test_align_host.py.)"""
import os

import pytest

import align_checks as ac
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _ref_genome():
    return b"".join(l.strip() for l in open(os.path.join(GOLDEN, "isolate.ref.fa"), "rb") if not l.startswith(b">"))


def _pairs(pmx, genome, n, seed, **kw):
    concat, off = pmx.simulate_paired_reads(genome, n, seed=seed, **kw)
    reads = [bytes(concat[off[i]:off[i + 1]]) for i in range(len(off) - 1)]
    return [r if i % 2 == 0 else pmx.reverse_complement(r) for i, r in enumerate(reads)]   # readFastqPaired orientation


FORMS = {
    "fused": {"PMX_ALIGN_COMPACT_FUSED": "1"},
    "pos32": {"PMX_ALIGN_COMPACT_POS32": "1"},
    "pos32_fused": {"PMX_ALIGN_COMPACT_POS32": "1", "PMX_ALIGN_COMPACT_FUSED": "1"},
    "resident_grids": {"PMX_ALIGN_COMPACT_WAVES": "7", "PMX_ALIGN_CSEED_WAVES": "16"},
    # the general tiers of the pairs the seeds kernel gave up on, on the context's second stream beside the chain kernels
    "early_tail": {"PMX_ALIGN_EARLY_TAIL": "1"},
}


@pytest.mark.parametrize("form", sorted(FORMS))
def test_compact_tier_forms_equal_reference(pmx, oracle, ctx, form, monkeypatch):
    for k, v in FORMS[form].items():
        monkeypatch.setenv(k, v)
    g = _ref_genome()
    clean = _pairs(pmx, g, 3000, 41)
    noisy = _pairs(pmx, g, 1500, 42, sub_rate=0.01)
    al = pmx.Aligner(ctx, g, 150)
    with_n = [r if i % 3 else r[:40] + b"NN" + r[42:] for i, r in enumerate(_pairs(pmx, g, 12000, 43))]   # enough bails for the thread-per-pair passes
    for name, reads, floor in (("clean", clean, 0.95), ("noisy", noisy, 0.5), ("with_n", with_n, 0.2)):
        got = al.align_reads(reads, paired=True)
        want = oracle.ref_align_reads_direct(g, reads, True, 8)
        bad = ac.compare_results(got, want)
        assert not bad, (form, name, bad[:5])
        assert all(x["flags"] & 3 == 0 for x in got), (form, name)
        st = al.stats()
        assert st["compact_tier_items"] >= floor * (len(reads) // 2), (form, name, st)


def test_compact_second_form_on_example_reads(pmx, oracle, ctx, monkeypatch):
    """k_align_compact16_multi (several regions per mate): on the real example pairs -- mates that overlap on the reference --
    the compact tier finishes more than twice as many pairs with it as without (PMX_ALIGN_NO_MULTI), the records are the
    reference's either way, and with 32-bit position words"""
    g = _ref_genome()
    seqs, _, _ = pmx.read_fastq_paired(os.path.join(GOLDEN, "isolate_R1.fastq.gz"), os.path.join(GOLDEN, "isolate_R2.fastq.gz"))
    reads = seqs[:40000]
    mean_len = int(sum(len(r) for r in reads) / len(reads))
    want = oracle.ref_align_reads_direct(g, reads, True, 8)
    al = pmx.Aligner(ctx, g, mean_len)
    finished = {}
    for mode in ("multi", "no_multi", "multi_pos32"):
        if mode == "no_multi":
            monkeypatch.setenv("PMX_ALIGN_NO_MULTI", "1")
        if mode == "multi_pos32":
            monkeypatch.setenv("PMX_ALIGN_COMPACT_POS32", "1")
        got = al.align_reads(reads, paired=True)
        monkeypatch.delenv("PMX_ALIGN_NO_MULTI", raising=False)
        monkeypatch.delenv("PMX_ALIGN_COMPACT_POS32", raising=False)
        bad = ac.compare_results(got, want)
        assert not bad, (mode, bad[:10])
        st = al.stats()
        finished[mode] = st["compact_tier_items"]
        assert st["n_items"] == len(reads) // 2
    assert finished["multi"] > 2 * finished["no_multi"] and finished["multi"] >= 0.7 * (len(reads) // 2), finished
    assert finished["multi_pos32"] == finished["multi"]
    # the edit counts (score_reads_vs_reference, what --refine sums): the second form's equal the general tiers' and the reference's
    rs = pmx.ReadSet(ctx, reads)
    want_score = oracle.ref_score_reads(g, reads, True)
    assert al.score_reads(rs, True, False) == want_score
    monkeypatch.setenv("PMX_ALIGN_NO_MULTI", "1")
    assert al.score_reads(rs, True, False) == want_score
    monkeypatch.delenv("PMX_ALIGN_NO_MULTI", raising=False)
    rs.close()
