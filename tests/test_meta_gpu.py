"""--meta on the device (pmx_meta_*, panmap_amd/csrc/api_meta.hip) against the restatement oracle/oracle_meta.py, and the
reference's own e2e expectation for this mode: src/test/e2e/run_e2e.sh:182-204 -- 700 reads tiled over MZ515733.1 plus 300
over node_1330 of rsv_4K must come out as exactly two haplotypes, MZ515733.1 in (0.55, 0.82) and node_1330 in (0.18, 0.45),
summing to 1.  (The README demo's 5-haplotype SARS reads are absent from the reference checkout, .MISSING_LARGE_BLOBS, so its
golden abundance file cannot be reproduced; parity of this mode is otherwise unpinned -- see the oracle's header.)"""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _fasta(path):
    return "".join(l.strip() for l in open(path) if not l.startswith(">")).upper()


def _tile(g, n, L=150):
    """the e2e script's read generator: n reads of L bases at a fixed step"""
    step = max(1, (len(g) - L) // n)
    out, c, i = [], 0, 0
    while c < n and i + L <= len(g):
        out.append(g[i:i + L].encode())
        c += 1
        i += step
    return out


@pytest.fixture(scope="module")
def rsv_meta(pmx):
    pm = pmx.Panman(os.path.join(GOLDEN, "rsv_4K.panman"))
    ctx = pmx.Context(0)
    meta = pmx.Meta.build(ctx, pm)
    return pm, ctx, meta


def test_reference_e2e_mixture_70_30(pmx, rsv_meta):
    pm, ctx, meta = rsv_meta
    a, b = _fasta(os.path.join(GOLDEN, "MZ515733.1.fa")), _fasta(os.path.join(GOLDEN, "rsv_4K.panman.random.node_1330.fa"))
    reads = _tile(a, 700) + _tile(b, 300)
    meta.set_reads(reads)
    meta.score(top_oc=1000)
    haps = meta.em()
    text = pmx.format_abundance(haps, meta.index.node_id)
    lines = [l.split("\t") for l in text.splitlines()]
    assert len(lines) == 2, text                                        # "exactly 2 haplotypes"
    got = {ids: float(p) for ids, p in lines}
    assert 0.55 < got["MZ515733.1"] < 0.82 and 0.18 < got["node_1330"] < 0.45, text
    assert 0.99 < sum(got.values()) < 1.01
    info = meta.em_info()
    assert 1 <= info["rounds"] <= 5 and info["iterations"] >= 2
    # single-source samples come back as one haplotype
    meta.set_reads(_tile(a, 500))
    meta.score()
    haps = meta.em()
    assert meta.index.node_id(haps[0][0]) == "MZ515733.1" and haps[0][1] > 0.95


def test_scores_and_overlap_equal_the_restatement(pmx, rsv_meta):
    """every (read, candidate) parsimony score equals the tree-walking restatement, for reads of both strands, reads with
    substitutions and reads that share seedmers; the overlap coefficients (through the place stage) equal it too"""
    from oracle import oracle_meta as om
    pm, ctx, meta = rsv_meta
    rng = np.random.default_rng(5)
    a, b = _fasta(os.path.join(GOLDEN, "MZ515733.1.fa")), _fasta(os.path.join(GOLDEN, "rsv_4K.panman.random.node_1330.fa"))
    reads = _tile(a, 120) + _tile(b, 80)
    reads = [pmx.reverse_complement(r) if i % 3 == 0 else r for i, r in enumerate(reads)]
    mutated = []
    for r in reads[:60]:
        q = bytearray(r)
        for p in rng.integers(0, len(q), 2):
            q[p] = b"ACGT"[(b"ACGT".index(q[p]) + 1) % 4] if q[p] in b"ACGT" else q[p]
        mutated.append(bytes(q))
    reads = reads + mutated + reads[:10]                                  # duplicates: merged with a multiplicity
    meta.set_reads(reads)
    n_nodes = meta.index.info.n_nodes
    cands = np.unique(np.concatenate([rng.integers(0, n_nodes, 40), [0, n_nodes - 1, pm.find_node("MZ515733.1"), pm.find_node("node_1330")]])).astype(np.uint32)
    meta.score(candidates=cands)
    assert np.array_equal(meta.candidates(), cands)
    got = meta.scores()
    off, h, rev = meta.read_seedmers()
    ns, mult = meta.read_info()
    assert mult.max() >= 2 and got.shape == (len(ns), len(cands))
    # the reads' seedmer lists and multiplicities against the from-the-string restatement (no product code in it)
    want_reads = {}
    for r in reads:
        sm = tuple(om.seedmers(r, 19, 8, 3))
        if sm:
            want_reads[sm] = want_reads.get(sm, 0) + 1
    got_reads = {tuple((int(h[q]), bool(rev[q])) for q in range(off[i], off[i + 1])): int(mult[i]) for i in range(len(ns))}
    assert got_reads == want_reads and np.array_equal(ns, np.diff(off))
    read_hashes = set(h.tolist())
    o_arr, u_arr = meta.index_oriented.arrays(), meta.index.arrays()
    for j, node in enumerate(cands.tolist()):
        counts = om.node_seed_counts(o_arr, node, read_hashes)
        want = om.read_scores(counts, off, h, rev)
        assert np.array_equal(got[:, j].astype(np.int64), want), (node, np.nonzero(got[:, j] != want)[0][:5])
    assert got.max() > 20 and (got == 0).any()
    oc = meta.overlap_coefficients()
    for node in cands.tolist()[::6]:
        assert abs(oc[node] - om.overlap_coefficient(u_arr, node, read_hashes)) < 1e-12, node


def test_em_equals_the_restatement(pmx, rsv_meta):
    """the device EM (fixed-order FP64 reductions) and the numpy restatement agree to 1e-9 on every proportion, on the
    mixture and on a three-way mixture with a small component"""
    from oracle import oracle_meta as om
    pm, ctx, meta = rsv_meta
    a, b = _fasta(os.path.join(GOLDEN, "MZ515733.1.fa")), _fasta(os.path.join(GOLDEN, "rsv_4K.panman.random.node_1330.fa"))
    c = pm.genome(pm.find_node("node_1330") // 2).decode()
    for reads in (_tile(a, 700) + _tile(b, 300), _tile(a, 500) + _tile(b, 300) + _tile(c, 60)):
        meta.set_reads(reads)
        meta.score(top_oc=50)
        haps = meta.em()
        sc = meta.scores().astype(np.int64)
        ns, mult = meta.read_info()
        # columns as the device merges them: equal score columns are one column (lowest candidate first)
        cols, seen = [], {}
        for j in range(sc.shape[1]):
            key = sc[:, j].tobytes()
            if key not in seen:
                seen[key] = j
                cols.append(j)
        rows = sc.max(axis=1) > 0
        kept, props = om.square_em(sc[rows][:, cols], ns[rows], mult[rows])
        cands = meta.candidates()
        want = sorted(((float(p), int(cands[cols[k]])) for k, p in zip(kept, props)), reverse=True)
        got = [(p, node) for node, p, _ in haps]
        assert [n for _, n in want] == [n for _, n in got]
        assert np.allclose([p for p, _ in want], [p for p, _ in got], rtol=0, atol=1e-9)


def _merged_columns(sc):
    """columns as the device merges them: equal score columns are one column (lowest candidate first)"""
    cols, seen = [], {}
    for j in range(sc.shape[1]):
        key = sc[:, j].tobytes()
        if key not in seen:
            seen[key] = j
            cols.append(j)
    return cols


def test_config5_sars_five_haplotypes(pmx):
    """BASELINE configs[4] on its own tree: a 5-haplotype mixture against the SARS-CoV-2 20k PanMAN through pmx_meta_*.
    The README demo's reads (examples/data/reads/sars20000_5hap_*) are absent from the reference checkout, so the mixture is
    drawn here: 200,000 paired 150-bp reads, 50/20/15/10/5 %, from the five leaves that lead the reference's golden
    abundance file (examples/expected/meta_abundance/example.mgsr.abundance.out -> tests/golden/), run as the demo command
    runs (--em-delta-threshold 0.00001).
      * read side: the seedmer lists + multiplicities of a 20,000-read slice equal the from-the-string restatement;
      * scores: ALL distinct reads x 64 candidates (the five sources + 59 random nodes) equal scores computed from the
        candidates' GENOME STRINGS (oracle_meta.genome_seed_counts: no index, no product code on the checker's side);
      * overlap coefficients of the five sources equal the from-the-string value;
      * EM on those 64 candidates equals the numpy restatement to 1e-9;
      * the full run (top-oc 1000 -> ~2,700 candidates) reports exactly the five sources, in the order of their shares, each
        within 0.03 of its share -- and, like the reference's golden file, within 0.03 of the golden proportions."""
    from oracle import oracle_meta as om
    from panmap_amd import _lib
    pm = pmx.Panman(os.path.join(GOLDEN, "sars_20000_twilight_dipper.panman"))
    ctx = pmx.Context(0)
    meta = pmx.Meta.build(ctx, pm)
    golden = [l.rstrip("\n").split("\t") for l in open(os.path.join(GOLDEN, "example.mgsr.abundance.out"))]
    names = [g[0] for g in golden[:5]]
    shares = [0.50, 0.20, 0.15, 0.10, 0.05]
    src = [pm.find_node(n) for n in names]
    n_reads = 200000
    parts, offs, base = [], [np.zeros(1, np.int64)], 0
    for i, (node, sh) in enumerate(zip(src, shares)):
        c, o = pmx.simulate_paired_reads(pm.genome(node), int(n_reads * sh) // 2, seed=10 + i)
        parts.append(c)
        offs.append(np.asarray(o[1:], np.int64) + base)
        base += int(o[-1])
    concat, offsets = np.concatenate(parts), np.concatenate(offs)
    assert len(offsets) - 1 == n_reads
    meta.set_reads(concat=concat, offsets=offsets)
    off, h, rev = meta.read_seedmers()
    ns, mult = meta.read_info()
    assert meta.n_reads > 50000 and int(mult.sum()) <= n_reads and np.array_equal(ns, np.diff(off))
    # ---- read side, a slice of the raw reads (every tenth): its seedmer lists are among the merged reads'
    got_reads = {}
    hl, rl = h.tolist(), rev.tolist()
    for i in range(len(ns)):
        got_reads[tuple(zip(hl[off[i]:off[i + 1]], map(bool, rl[off[i]:off[i + 1]])))] = int(mult[i])
    seen = {}
    for r in range(0, n_reads, 10):
        sm = tuple(om.seedmers(bytes(concat[offsets[r]:offsets[r + 1]]), 19, 8, 3))
        if sm:
            seen[sm] = seen.get(sm, 0) + 1
    assert len(seen) > 5000
    for sm, c in seen.items():
        assert got_reads.get(sm, 0) >= c, sm[:2]
    # ---- scores of all reads x 64 candidates, from the candidates' genome strings
    rng = np.random.default_rng(55)
    n_nodes = meta.index.info.n_nodes
    cands = np.unique(np.concatenate([np.array(src), rng.integers(0, n_nodes, 59)])).astype(np.uint32)
    meta.score(candidates=cands)
    assert np.array_equal(meta.candidates(), cands)
    got = meta.scores()
    assert got.shape == (len(ns), len(cands))
    read_hashes = set(hl)
    for j, node in enumerate(cands.tolist()):
        counts = {hh: c for hh, c in om.genome_seed_counts(pm.genome(node), 19, 8, 3).items() if hh in read_hashes}
        want = om.read_scores_np(counts, off, h, rev)
        assert np.array_equal(got[:, j].astype(np.int64), want), (node, np.nonzero(got[:, j] != want)[0][:5])
    assert got.max() >= 30
    oc = meta.overlap_coefficients()
    for node in src:
        gs = om.genome_seed_counts(pm.genome(node), 19, 8, 3)
        assert abs(oc[node] - sum(1 for hh in gs if hh in read_hashes) / len(gs)) < 1e-12, node
    # ---- EM on the 64 candidates against the numpy restatement (the demo's stop rule)
    mp = _lib.MetaParams(em_delta_threshold=1e-5)
    haps = meta.em(mp)
    sc = got.astype(np.int64)
    cols = _merged_columns(sc)
    rows = sc.max(axis=1) > 0
    kept, props = om.square_em(sc[rows][:, cols], ns[rows], mult[rows], delta_threshold=1e-5)
    want = sorted(((float(p), int(cands[cols[k]])) for k, p in zip(kept, props)), reverse=True)
    got_h = [(p, node) for node, p, _ in haps]
    assert [n for _, n in want] == [n for _, n in got_h]
    assert np.allclose([p for p, _ in want], [p for p, _ in got_h], rtol=0, atol=1e-9)
    # ---- the full run, as the demo command
    meta.score(top_oc=1000)
    assert len(meta.candidates()) > 500 and set(src) <= set(meta.candidates().tolist())
    haps = meta.em(_lib.MetaParams(em_delta_threshold=1e-5))
    text = pmx.format_abundance(haps, pm.node_id)
    lines = [l.split("\t") for l in text.splitlines()]
    assert [l[0].split(",")[0] for l in lines] == names, text          # exactly the five sources, by share
    for (ids, p), sh, g in zip(lines, shares, golden):
        assert abs(float(p) - sh) <= 0.03, text
        assert abs(float(p) - float(g[1])) <= 0.03, (text, golden)      # the reference's golden file has the same shape
    assert abs(sum(float(p) for _, p in lines) - 1.0) < 1e-4
    info = meta.em_info()
    assert 1 <= info["rounds"] <= 5 and info["iterations"] < 1000
    meta.close()
