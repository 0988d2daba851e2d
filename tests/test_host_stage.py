"""CPU suite: host logic of the product (PanMAN reader, genome materialiser, index stage, FASTQ
readers) and the C-ABI surface.  No device compute."""
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def _fa(path):
    return b"".join(l.strip() for l in open(path, "rb") if not l.startswith(b">"))


def test_abi_exports_every_declared_symbol(pmx):
    header = open(os.path.join(ROOT, "include", "panmap_amd.h")).read()
    declared = set(re.findall(r"\b(pmx_[a-z0-9_]+)\s*\(", header))
    declared -= {"pmx_ctx", "pmx_place", "pmx_index", "pmx_panman", "pmx_readset", "pmx_aligner"}
    assert declared, "no declarations parsed"
    assert not pmx._lib.MISSING
    for name in declared:
        assert hasattr(pmx.lib, name), name
    assert set(pmx._lib.SIGNATURES) >= declared


def test_no_gpu_fails_loudly(pmx):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pmx.PmxError) as e:
        pmx.Context(0)
    assert "NO_DEVICE" in str(e.value)


def test_product_does_not_touch_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "panmap_amd")):
        if "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle/" not in txt and "liboracle" not in txt and "libpanmap_ref" not in txt, os.path.join(dirpath, f)


def test_node_genomes_match_reference_fixtures(pmx, sars):
    assert sars.num_nodes == 39999 and sars.num_blocks == 1
    i = sars.find_node("node_7618")
    assert i == 15189
    assert sars.genome(i) == _fa(os.path.join(GOLDEN, "isolate.ref.fa"))
    rsv = pmx.Panman(os.path.join(GOLDEN, "rsv_4K.panman"))           # multi-block, gap lists, inverted blocks
    assert rsv.genome("MZ515733.1") == _fa(os.path.join(GOLDEN, "MZ515733.1.fa"))
    assert rsv.genome("node_1330") == _fa(os.path.join(GOLDEN, "rsv_4K.panman.random.node_1330.fa"))
    assert rsv.find_node("no_such_node") == -1


def test_index_invariants(pmx, sars, sars_index, oracle):
    # src/test/test_index.cpp:41-78: offsets monotone, parentCount == running count along the path,
    # only changed counts stored, changes sorted by hash within a node
    a = sars_index.arrays()
    info = sars_index.info
    assert (info.k, info.s, info.t, info.l) == (19, 8, 0, 3)
    assert info.n_nodes == 39999 and info.n_changes == 2422076
    off = a["offsets"].astype(np.int64)
    assert off[0] == 0 and np.all(np.diff(off) >= 0) and off[-1] == info.n_changes
    assert off[1] == 8889
    assert np.all(a["parent_count"] != a["child_count"])
    assert np.all(a["parent"][1:] < np.arange(1, info.n_nodes))
    rng = np.random.default_rng(11)
    for leaf in rng.integers(0, info.n_nodes, 6):
        path = []
        x = int(leaf)
        while True:
            path.append(x)
            if x == 0:
                break
            x = int(a["parent"][x])
        counts = {}
        for nd in reversed(path):
            h = a["hash"][off[nd]:off[nd + 1]]
            assert np.all(h[1:] > h[:-1])
            for hh, pc, cc in zip(h.tolist(), a["parent_count"][off[nd]:off[nd + 1]].tolist(), a["child_count"][off[nd]:off[nd + 1]].tolist()):
                assert counts.get(hh, 0) == pc
                if cc:
                    counts[hh] = cc
                else:
                    counts.pop(hh, None)
        assert all(v > 0 for v in counts.values())


def test_index_flank0_l1_equals_direct_extraction(pmx, sars, oracle):
    # src/test/test_index.cpp:80-110: delta reconstruction == direct extractSeeds(genome) at l=1, flank 0
    idx = pmx.Index.build(sars, k=15, s=8, t=0, l=1, open_syncmer=False, flank_mask=0)
    a = idx.arrays()
    off = a["offsets"].astype(np.int64)
    for nd in (0, 15189, 25000, 39998):
        path = []
        x = nd
        while True:
            path.append(x)
            if x == 0:
                break
            x = int(a["parent"][x])
        counts = {}
        for p in reversed(path):
            for hh, cc in zip(a["hash"][off[p]:off[p + 1]].tolist(), a["child_count"][off[p]:off[p + 1]].tolist()):
                if cc:
                    counts[hh] = cc
                else:
                    counts.pop(hh, None)
        g = sars.genome(nd)
        want = {}
        for h, _, _, _ in oracle.rolling_syncmers(g, 15, 8, False, 0, False):
            want[h] = want.get(h, 0) + 1
        assert counts == want


def test_incremental_and_from_scratch_producers_agree(pmx, sars):
    """two independent index producers (incremental DFS vs re-seeding every node's genome and diffing against
    the parent) give the same index when no flank mask is in play; with the 250-base hard mask the reference's
    incremental rule (src/index_single_mode.cpp) keeps syncmers whose mask status changed only because the
    genome ends moved, which the from-scratch definition does not: the root and all nodes whose flanks did not
    move must still agree"""
    a = pmx.Index.build(sars, k=19, s=8, t=0, l=3, flank_mask=0, mode=2, max_nodes=250).arrays()
    b = pmx.Index.build(sars, k=19, s=8, t=0, l=3, flank_mask=0, mode=1, max_nodes=250).arrays()
    for key in a:
        assert np.array_equal(a[key], b[key]), key
    a = pmx.Index.build(sars, flank_mask=250, mode=2, max_nodes=20).arrays()
    b = pmx.Index.build(sars, flank_mask=250, mode=1, max_nodes=20).arrays()
    for key in a:
        assert np.array_equal(a[key], b[key]), key      # the first 20 nodes of the DFS have equal-length genomes


def _path_to_root(parent, nd):
    path = []
    x = int(nd)
    while True:
        path.append(x)
        if x == 0:
            return path[::-1]
        x = int(parent[x])


def _reconstruct(a, off, nd, check_parent_counts=False):
    counts = {}
    for p in _path_to_root(a["parent"], nd):
        sl = slice(off[p], off[p + 1])
        for hh, pc, cc in zip(a["hash"][sl].tolist(), a["parent_count"][sl].tolist(), a["child_count"][sl].tolist()):
            if check_parent_counts:
                assert counts.get(hh, 0) == pc
            if cc:
                counts[hh] = cc
            else:
                counts.pop(hh, None)
    return counts


def _direct(oracle, genome, k, s):
    want = {}
    for h, _, _, _ in oracle.rolling_syncmers(genome, k, s, False, 0, False):
        want[h] = want.get(h, 0) + 1
    return want


def test_rsv_incremental_producer_equals_from_scratch_and_direct_extraction(pmx, oracle):
    """rsv_4K (src/test/data; multi-block, gap lists, inverted block insertions) at the reference's own test
    setting K=15, S=8, L=1, flank 0 (src/test/test_index.cpp:16): the incremental producer walks inverted
    blocks in genome order and must equal the from-scratch producer array for array; offsets / parent counts
    are consistent along root paths (:41-78) and the reconstructed multiset of sampled nodes equals the direct
    extraction of their genomes (:80-110)"""
    rsv = pmx.Panman(os.path.join(GOLDEN, "rsv_4K.panman"))
    a = pmx.Index.build(rsv, k=15, s=8, t=0, l=1, flank_mask=0, mode=2).arrays()
    b = pmx.Index.build(rsv, k=15, s=8, t=0, l=1, flank_mask=0, mode=1).arrays()
    for key in a:
        assert np.array_equal(a[key], b[key]), key
    off = a["offsets"].astype(np.int64)
    n = len(off) - 1
    assert off[0] == 0 and np.all(np.diff(off) >= 0) and off[-1] == len(a["hash"])
    assert np.all(a["parent_count"] != a["child_count"])
    rng = np.random.default_rng(5)
    for nd in [0, rsv.find_node("MZ515733.1"), rsv.find_node("node_1330")] + rng.integers(0, n, 5).tolist():
        assert _reconstruct(a, off, nd, check_parent_counts=True) == _direct(oracle, rsv.genome(nd), 15, 8)


def test_injected_block_inversions_reconstruct_like_direct_extraction(pmx, oracle):
    """src/test/test_index.cpp:145-264: pure inversions of the largest forward block injected at 25 leaves; the
    incremental index of the mutated tree reconstructs every injected and control leaf to the direct extraction
    of its (changed) genome, and equals the from-scratch index"""
    rsv = pmx.Panman(os.path.join(GOLDEN, "rsv_4K.panman"))
    n = rsv.num_nodes
    parents = np.array([rsv.parent(i) for i in range(n)])
    is_leaf = np.ones(n, bool)
    is_leaf[parents[1:]] = False
    leaves = np.flatnonzero(is_leaf)
    np.random.default_rng(7).shuffle(leaves)
    injected, before = [], {}
    for nd in leaves[:25].tolist():
        before[nd] = rsv.genome(nd)
        assert pmx.lib.pmx_panman_test_invert_block(rsv._h, nd, 15) >= 0
        injected.append(nd)
    controls = leaves[25:45].tolist()
    # one of these leaves' largest forward block is a run of N, which is its own reverse complement
    assert sum(rsv.genome(nd) != before[nd] for nd in injected) >= 24
    a = pmx.Index.build(rsv, k=15, s=8, t=0, l=1, flank_mask=0, mode=2).arrays()
    b = pmx.Index.build(rsv, k=15, s=8, t=0, l=1, flank_mask=0, mode=1).arrays()
    for key in a:
        assert np.array_equal(a[key], b[key]), key
    off = a["offsets"].astype(np.int64)
    for nd in injected + controls:
        assert _reconstruct(a, off, nd) == _direct(oracle, rsv.genome(nd), 15, 8), nd


def test_fastq_readers(pmx, tmp_path):
    fq1 = tmp_path / "a_R1.fastq"
    fq2 = tmp_path / "a_R2.fastq"
    fq1.write_text("@r1 x\nACGTN\n+\nIIIII\n@r2\nGGGTT\n+\nIIIII\n")
    fq2.write_text("@r1\nAACCg\n+\nABCDE\n@r2\nTTTTT\n+\nIIIII\n")
    assert pmx.extract_read_sequences(str(fq1), str(fq2)) == [b"ACGTN", b"AACCg", b"GGGTT", b"TTTTT"]
    seqs, quals, names = pmx.read_fastq_paired(str(fq1), str(fq2))
    assert seqs == [b"ACGTN", b"gGGTT", b"GGGTT", b"AAAAA"]        # lower-case left as-is, then reversed
    assert quals[1] == b"EDCBA" and names[0] == b"r1"
    fq2.write_text("@r1\nAACCG\n+\nABCDE\n")
    with pytest.raises(ValueError):
        pmx.extract_read_sequences(str(fq1), str(fq2))
    fa = tmp_path / "x.fa"
    fa.write_text(">s1 desc\nACGT\nACGT\n>s2\nTT\n")
    n, s, q = pmx.read_fastx(str(fa))
    assert s == [b"ACGTACGT", b"TT"] and n == [b"s1", b"s2"]


def test_native_fastx_reader_matches_python_reader(pmx, tmp_path):
    """pmx_fastx_read / pmx_fastx_read_paired (C++, kseq conventions of src/seeding.cpp:231-269) against the Python
    mirror on the repository's real FASTQ pair and on hand-made edge cases"""
    import gzip
    from conftest import GOLDEN
    r1, r2 = os.path.join(GOLDEN, "isolate_R1.fastq.gz"), os.path.join(GOLDEN, "isolate_R2.fastq.gz")
    fx = pmx.read_fastq_paired_native(r1, r2)
    seqs, quals, names = pmx.read_fastq_paired(r1, r2)
    assert fx.n == len(seqs) and fx.lists() == (seqs, quals, names)
    assert int(fx.off[-1]) == sum(len(x) for x in seqs)
    # edge cases: CRLF, multi-line records, lower case, tab in the header, gzip / plain, FASTA mates (no qualities -> 'I')
    a = tmp_path / "e_R1.fastq"
    b = tmp_path / "e_R2.fastq.gz"
    a.write_bytes(b"@r1 x y\r\nACGTN\r\n+\r\nIIIII\r\n@r2\tz\nGG\nGTT\n+\nII\nIII\n@r3\nacgT\n+r3\n!!!!\n")
    with gzip.open(b, "wb") as f:
        f.write(b"@r1\nAACCg\n+\nABCDE\n@r2\nTTTTT\n+\nIIIII\n@r3\nNNAC\n+\n1234\n")
    fx = pmx.read_fastq_paired_native(str(a), str(b))
    assert fx.lists() == pmx.read_fastq_paired(str(a), str(b))
    assert fx.lists()[0] == [b"ACGTN", b"gGGTT", b"GGGTT", b"AAAAA", b"acgT", b"GTNN"]
    assert fx.lists()[1][5] == b"4321" and fx.lists()[2] == [b"r1", b"r1", b"r2", b"r2", b"r3", b"r3"]
    fa = tmp_path / "m.fa"
    fa.write_text(">s1 desc\nACGT\nACGT\n>s2\nTT\n\n>s3\n")
    one = pmx.read_fastx_native(str(fa))
    assert one.lists()[0] == [b"ACGTACGT", b"TT", b""] and one.lists()[2] == [b"s1", b"s2", b"s3"]
    assert pmx.read_fastq_paired_native(str(fa)).lists()[1] == [b"IIIIIIII", b"II", b""]     # single-end FASTA: 'I' qualities
    # a record whose quality is shorter than its sequence ends the file (kseq_read returns -2 there)
    t = tmp_path / "t.fastq"
    t.write_text("@ok\nACGT\n+\nIIII\n@bad\nACGTACGT\n+\nIII\n")
    assert pmx.read_fastx_native(str(t)).lists()[0] == [b"ACGT"]
    # mate-count mismatch and unreadable file are errors
    c = tmp_path / "c_R2.fastq"
    c.write_text("@r1\nAACCG\n+\nABCDE\n")
    with pytest.raises(pmx.PmxError):
        pmx.read_fastq_paired_native(str(a), str(c))
    with pytest.raises(pmx.PmxError):
        pmx.read_fastx_native(str(tmp_path / "missing.fq"))


def test_native_fastx_reader_reads_pipes(pmx, tmp_path):
    """a FIFO (process substitution, /dev/stdin) cannot be rewound or reopened: the reader opens the path once and takes the
    one-read fast path for regular files only, as the reference's gzopen + kseq does (src/seeding.cpp:231-269) -- the first
    record of a plain FASTQ from a pipe must not be lost, and a gzip stream from a pipe must inflate"""
    import gzip
    import threading
    plain = b"".join(b"@p%d\nACGTACGTAC\n+\nIIIIIIIIII\n" % i for i in range(3000))
    for payload, data in (("plain", plain), ("gzip", gzip.compress(plain))):
        fifo = tmp_path / ("in_%s.fq" % payload)
        os.mkfifo(fifo)

        def feed(path=fifo, blob=data):
            with open(path, "wb") as f:
                f.write(blob)
        th = threading.Thread(target=feed)
        th.start()
        fx = pmx.read_fastx_native(str(fifo))
        th.join()
        seqs, quals, names = fx.lists()
        assert len(seqs) == 3000 and names[0] == b"p0" and names[-1] == b"p2999" and set(seqs) == {b"ACGTACGTAC"}, payload


def test_synthetic_reads_are_deterministic_and_fr(pmx, sars):
    g = sars.genome("node_7618")
    c1, o1 = pmx.simulate_paired_reads(g, 500, seed=42)
    c2, o2 = pmx.simulate_paired_reads(g, 500, seed=42)
    assert np.array_equal(c1, c2) and np.array_equal(o1, o2) and len(o1) == 1001
    c3, _ = pmx.simulate_paired_reads(g, 500, seed=42, sub_rate=0.0)
    r1 = bytes(c3[o1[0]:o1[1]]); r2 = bytes(c3[o1[1]:o1[2]])
    assert r1 in g and pmx.reverse_complement(r2) in g


@pytest.mark.parametrize("name", ["sars_20000_twilight_dipper.panman", "rsv_4K.panman"])
def test_parallel_index_build_equals_serial(pmx, name, monkeypatch):
    """the reference's contract src/test/test_index.cpp:112-139 (4-thread build == 1-thread build): the chunked producer
    (root state copied, path to the chunk's first node replayed; src/index_single_mode.cpp:2291-2470) gives the serial
    producer's arrays, element for element -- on the SARS tree and on rsv_4K (inverted block insertions), for the place
    stage's index with its 250-base flank mask and for the oriented (--meta) index"""
    pm = pmx.Panman(os.path.join(GOLDEN, name))
    for kw in (dict(), dict(mode=0x100, flank_mask=0)):
        monkeypatch.setenv("PMX_INDEX_THREADS", "1")
        one = pmx.Index.build(pm, **kw).arrays()
        monkeypatch.setenv("PMX_INDEX_THREADS", "5")
        many = pmx.Index.build(pm, **kw).arrays()
        for k in ("parent", "offsets", "hash", "parent_count", "child_count"):
            assert np.array_equal(one[k], many[k]), (kw, k)


def test_parallel_fastq_scan_equals_serial(pmx, tmp_path, monkeypatch):
    """the native reader cuts a large four-line FASTQ at record starts and parses the pieces side by side
    (src/placement.cpp:97-161 does the same for uncompressed input): same arrays as the one-thread parse -- quality lines
    that begin with '@', CRLF line ends and names with comments included -- and a file whose records are wrapped falls back
    to the general parser"""
    rng = np.random.default_rng(4)
    path = tmp_path / "big.fastq"
    with open(path, "wb") as f:
        for i in range(70000):
            n = int(rng.integers(40, 160))
            seq = bytes(rng.choice(list(b"ACGTN"), n).astype(np.uint8))
            qual = b"@" + bytes(rng.integers(33, 74, n - 1).astype(np.uint8)) if i % 7 == 0 else bytes(rng.integers(35, 74, n).astype(np.uint8))
            eol = b"\r\n" if i % 1000 == 0 else b"\n"
            f.write(b"@read%d comment %d" % (i, i) + eol + seq + eol + b"+" + eol + qual + eol)
    assert os.path.getsize(path) > (8 << 20)

    def read(threads, p=path):
        monkeypatch.setenv("PMX_FASTX_THREADS", str(threads))
        fx = pmx.read_fastx_native(str(p))
        return (np.array(fx.off), np.array(fx.name_off), bytes(fx.seq), bytes(fx.qual), bytes(fx.names_concat))
    one, many = read(1), read(7)
    assert len(one[0]) == 70001
    for a, b in zip(one, many):
        assert np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b
    wrapped = tmp_path / "wrapped.fastq"
    with open(wrapped, "wb") as f:
        for i in range(60000):
            seq = bytes(rng.choice(list(b"ACGT"), 120).astype(np.uint8))
            f.write(b"@w%d\n%s\n%s\n+\n%s\n%s\n" % (i, seq[:60], seq[60:], b"I" * 60, b"I" * 60))
    w1, w7 = read(1, wrapped), read(7, wrapped)
    assert len(w1[0]) == 60001 and all((np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b) for a, b in zip(w1, w7))


def test_one_table_of_switches(pmx, monkeypatch):
    """every PMX_* switch of the library sits in one table (csrc/device/pmx_options.hpp), read once and re-read on request;
    no library source calls getenv for a switch outside it"""
    import re
    text = pmx.describe_options()
    names = [l.split(" ")[0] for l in text.splitlines()]
    assert len(names) == len(set(names)) >= 50 and all(n.startswith("PMX_") for n in names)
    assert all(re.match(r"PMX_\w+ \[(env|test|ab|tune|diag)\] \S", l) for l in text.splitlines())
    monkeypatch.setenv("PMX_INDEX_THREADS", "3")          # (conftest re-reads the table)
    assert "PMX_INDEX_THREADS [env]" in pmx.describe_options() and "(= 3)" in [l for l in pmx.describe_options().splitlines() if l.startswith("PMX_INDEX_THREADS")][0]
    monkeypatch.delenv("PMX_INDEX_THREADS")
    assert "(= 3)" not in pmx.describe_options()
    src = os.path.join(ROOT, "panmap_amd", "csrc")
    for dp, dn, fn in os.walk(src):
        if os.path.basename(dp) in ("build", "cli"):
            continue
        for f in fn:
            if f.endswith((".hip", ".hpp", ".cpp", ".h")):
                for m in re.finditer(r'getenv\("(PMX_\w+)"\)', open(os.path.join(dp, f)).read()):
                    assert m.group(1) == "PMX_C_DUMP", (f, m.group(1))      # (hostsim-only debug print in aln_compact.hpp)
