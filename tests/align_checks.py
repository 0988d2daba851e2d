"""Helpers shared by the align tests: run the reference aligner (oracle/_ref), the CPU unit-test build
(tests/hostsim) and the HIP path, and compare per-read results field by field."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class Rec(C.Structure):
    _fields_ = [("rs", C.c_int32), ("re", C.c_int32), ("qs", C.c_int32), ("qe", C.c_int32),
                ("mapq", C.c_uint8), ("rev", C.c_uint8), ("proper_frag", C.c_uint8), ("mapped", C.c_uint8),
                ("n_cigar", C.c_uint16), ("flags", C.c_uint16), ("cigar_off", C.c_uint32), ("score", C.c_int32)]


_hs = {}


def hostsim(tpp=False):
    """tpp=False: the pipeline as the wave-per-pair kernels run it (PMX_W = 1); tpp=True: the thread-per-pair
    kernel's control flow, with its DP-request / replay rounds emulated on the host."""
    if tpp not in _hs:
        subprocess.run(["make", "-C", os.path.join(HERE, "hostsim")], check=True, stdout=subprocess.DEVNULL)
        L = C.CDLL(os.path.join(HERE, "hostsim", "libhostsim_tpp.so" if tpp else "libhostsim.so"))
        L.hs_align.restype = C.c_int
        L.hs_align.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int, C.POINTER(Rec),
                               C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_int]
        _hs[tpp] = L
    return _hs[tpp]


def records_to_results(recs, cig, n_reads, paired):
    """-> list of dicts shaped like oracle.call_align_reads_direct output."""
    def one(r):
        if not r.mapped or not (r.flags & 4):
            return dict(pos=2147483647 if True else 0, rs=0, re=0, qs=0, qe=0, mapq=0, rev=0, proper_frag=0, cigar=[])
        return dict(pos=r.rs + 1, rs=r.rs, re=r.re, qs=r.qs, qe=r.qe, mapq=r.mapq, rev=r.rev, proper_frag=r.proper_frag,
                    cigar=[int(x) for x in cig[r.cigar_off:r.cigar_off + r.n_cigar]])
    out = []
    if paired:
        for i in range(n_reads // 2):
            a, b = recs[2 * i], recs[2 * i + 1]
            out.append(dict(mapped=int(a.mapped), r1=one(a), r2=one(b), flags=a.flags | b.flags))
    else:
        for i in range(n_reads):
            out.append(dict(mapped=int(recs[i].mapped), r1=one(recs[i]), r2=None, flags=recs[i].flags))
    return out


def hostsim_align(reference: bytes, reads, paired, verbose=0, tpp=False):
    n = len(reads)
    arr = (C.c_char_p * n)(*reads)
    lens = (C.c_int * n)(*[len(r) for r in reads])
    recs = (Rec * max(n, 1))()
    cap = max(64, sum(len(r) for r in reads))
    cig = np.zeros(cap, np.uint32)
    used = C.c_int64()
    rc = hostsim(tpp).hs_align(reference, len(reference), n, arr, lens, int(paired), recs, cig.ctypes.data, cap, C.byref(used), verbose)
    assert rc == 0, rc
    return records_to_results(recs, cig, n, paired)


def hostsim_align_compact(reference: bytes, reads, rc2=False):
    """the compact tier (align/aln_compact.hpp) on the host -> (results, done flags per pair)"""
    L = hostsim(False)
    L.hs_align_compact.restype = C.c_int
    L.hs_align_compact.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int, C.POINTER(Rec),
                                   C.c_void_p, C.c_void_p]
    n = len(reads)
    arr = (C.c_char_p * n)(*reads)
    lens = (C.c_int * n)(*[len(r) for r in reads])
    recs = (Rec * max(n, 1))()
    cig = np.zeros(max(n, 1), np.uint32)
    done = np.zeros(max(n // 2, 1), np.int8)
    rc = L.hs_align_compact(reference, len(reference), n, arr, lens, int(rc2), recs, cig.ctypes.data, done.ctypes.data)
    assert rc == 0, rc
    return records_to_results(recs, cig, n, True), done[:n // 2]


def hostsim_ref_sketch(reference: bytes, w: int, k: int, slice_len: int = 0):
    """minimizers of a reference: slice_len == 0 -> the sequential sketch the host index build runs, > 0 -> the
    concatenation of independent slices of that many bases (what the device index build runs) -> (x, y) arrays"""
    L = hostsim(False)
    L.hs_ref_sketch.restype = C.c_int64
    L.hs_ref_sketch.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
    cap = len(reference) + 16
    x = np.zeros(cap, np.uint64)
    y = np.zeros(cap, np.uint64)
    n = L.hs_ref_sketch(reference, len(reference), w, k, slice_len, x.ctypes.data, y.ctypes.data, cap)
    assert 0 <= n <= cap
    return x[:n].copy(), y[:n].copy()


def rearranged_long_reads(pmx, genome: bytes, n: int, read_len: int, seed: int, sub=0.03, ins=0.01, dele=0.01):
    """long reads that make the long-read branches of mm_map_frag / mm_align1 run: a large deletion, a novel insertion,
    an inversion, a tandem duplication or two deletions in every read (several chains -> the RMQ re-chaining pass,
    map.c:296-305; z-drops across the breakpoints -> the inversion probe and mm_align1_inv, align.c:74-86, 835-885;
    opposite-strand secondaries -> mm_est_err / mm_filter_strand_retained), on top of substitutions and short indels"""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        st = int(rng.integers(0, len(genome) - read_len))
        frag = bytearray(genome[st:st + read_len])
        kind = int(rng.integers(5))
        p = int(rng.integers(read_len // 5, 4 * read_len // 5))
        m = int(rng.integers(100, min(3000, read_len // 2)))
        if kind == 0:
            del frag[p:p + m]
        elif kind == 1:
            frag[p:p] = bytes(rng.choice(list(b"ACGT"), m).astype(np.uint8))
        elif kind == 2:
            frag[p:p + m] = pmx.reverse_complement(bytes(frag[p:p + m]))
        elif kind == 3:
            frag[p:p] = frag[max(0, p - m):p]
        else:
            del frag[p:p + m]
            q = int(rng.integers(0, max(1, len(frag) - 200)))
            del frag[q:q + int(rng.integers(50, 800))]
        u = rng.random(len(frag))
        v = rng.random(len(frag))
        nb = rng.integers(0, 4, (2, len(frag)))
        r = bytearray()
        for i, b in enumerate(frag):
            if u[i] < dele:
                continue
            if u[i] < dele + ins:
                r.append(b"ACGT"[nb[0, i]])
            r.append(b"ACGT"[nb[1, i]] if v[i] < sub else b)
        r = bytes(r)
        out.append(pmx.reverse_complement(r) if rng.random() < 0.5 else r)
    return out


def inverted_repeat_case(pmx, genome: bytes, n: int, seed: int, copy_div: float, read_err: float, copy=(3000, 23000), read_len=None):
    """-> (reference, reads): the genome followed by a diverged reverse-complemented copy of a stretch of it (`copy`), and long
    reads drawn from around that stretch -- every read has a second, weaker chain on the OTHER strand that overlaps its primary on the read:
    mm_select_sub keeps it as strand_retained, and mm_est_err + mm_filter_strand_retained (esterr.c:30-64, hit.c:277-290)
    decide whether it stays (a diverged copy goes, a short near-identical one stays)"""
    rng = np.random.default_rng(seed)

    def mutate(s, rate):
        a = np.frombuffer(s, np.uint8).copy()
        hit = rng.random(len(a)) < rate
        a[hit] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(hit.sum()))]
        return a.tobytes()

    ref = genome + pmx.reverse_complement(mutate(genome[copy[0]:copy[1]], copy_div))
    reads = []
    for _ in range(n):
        if read_len is None:   # reads inside the copied stretch: the secondary is as long as the primary, only more diverged
            ln = int(rng.integers(9000, 14000))
            st = int(rng.integers(copy[0], copy[1] - ln))
        else:                  # reads that cover the whole copied stretch: the secondary is shorter than the primary
            ln = read_len
            st = int(rng.integers(max(0, copy[1] - ln), min(copy[0], len(genome) - ln)))
        r = mutate(genome[st:st + ln], read_err)
        reads.append(r if rng.random() < 0.5 else pmx.reverse_complement(r))
    return ref, reads


def golden_cases(pmx):
    """(genome, {name: (reads, expected results)}) of tests/golden/align_golden.json.gz; the inputs are regenerated
    exactly as tests/golden/make_align_golden.py made them"""
    import gzip
    import json
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_align_golden", os.path.join(HERE, "golden", "make_align_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    genome, sets = mod.inputs(pmx)
    exp = json.loads(gzip.open(os.path.join(HERE, "golden", "align_golden.json.gz")).read())
    out = {}
    for name, reads in sets.items():
        want = []
        for row in exp[name]:
            d = {"mapped": row[0]}
            for m, r in (("r1", row[1]), ("r2", row[2])):
                d[m] = dict(pos=r[0], rs=r[1], re=r[2], qs=r[3], qe=r[4], mapq=r[5], rev=r[6], proper_frag=r[7], cigar=r[8])
            want.append(d)
        out[name] = (reads, want)
    return genome, out


def cigar_str(c):
    return "".join("%d%s" % (x >> 4, "MIDNSHP=X"[x & 0xf]) for x in c)


def compare_results(got, want, label="", max_report=5):
    """Exact comparison of pos/rs/re/qs/qe/mapq/rev/proper_frag/CIGAR per read; unmapped -> only `mapped`."""
    bad = []
    for i, (g, w) in enumerate(zip(got, want)):
        if g["mapped"] != w["mapped"]:
            bad.append((i, "mapped", g["mapped"], w["mapped"]))
            continue
        if not w["mapped"]:
            continue
        for m in ("r1", "r2"):
            if w[m] is None:
                continue
            for f in ("pos", "rs", "re", "qs", "qe", "mapq", "rev", "proper_frag"):
                if g[m][f] != w[m][f]:
                    bad.append((i, m + "." + f, g[m][f], w[m][f]))
            if g[m]["cigar"] != w[m]["cigar"]:
                bad.append((i, m + ".cigar", cigar_str(g[m]["cigar"]), cigar_str(w[m]["cigar"])))
    return bad


def smoke_align(ctx, pm, genome, reads):
    """used by __graft_entry__.smoke(): a few hundred pairs through the HIP aligner vs the reference aligner."""
    import panmap_amd as pmx
    from oracle import oracle as orc
    if not hasattr(pmx, "Aligner"):
        return
    sub = reads[:400]
    sub_al = [r if i % 2 == 0 else pmx.reverse_complement(r) for i, r in enumerate(sub)]
    al = pmx.Aligner(ctx, genome, int(np.mean([len(r) for r in sub])))
    got = al.align_reads(sub_al, paired=True)
    want = orc.ref_align_reads_direct(genome, sub_al, True)
    bad = compare_results(got, want)
    assert not bad, bad[:5]
    print("smoke: align OK (%d pairs, %d mapped)" % (len(want), sum(w["mapped"] for w in want)))
    # the DP kernel on its own: a few requests of each call shape through the grouped service vs ksw_extd2_sse of the reference
    rng = np.random.default_rng(3)
    sc = al.scoring()
    mat = orc.simple_mat(sc["a"], sc["b"], sc["sc_ambi"])
    qs, ts, fl = [], [], []
    for i in range(96):
        t = [int(x) for x in rng.integers(0, 4, int(rng.integers(20, 129)))]
        q = [int(c) if rng.random() > 0.06 else int((c + 1) % 4) for c in t[:int(rng.integers(10, len(t) + 1))]]
        if i % 5 == 0 and len(q) > 12:
            del q[6:9]
        qs.append(q); ts.append(t); fl.append((0x08, 0x40, 0xc2)[i % 3])
    res, _ = al.dp_batch(qs, ts, -1, sc["zdrop"], sc["end_bonus"], fl)
    n_ok = 0
    for i in range(len(qs)):
        w = orc.ref_ksw_extd2(qs[i], ts[i], mat, sc["q"], sc["e"], sc["q2"], sc["e2"], -1, sc["zdrop"], sc["end_bonus"], fl[i])
        if not res[i]["served"]:
            assert w["n_cigar"] > 20, ("DP request not served", i)
            continue
        n_ok += 1
        for f in ("max", "zdropped", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "score", "n_cigar", "reach_end"):
            assert int(res[i][f]) == w[f], ("DP kernel differs from ksw_extd2_sse", i, f, int(res[i][f]), w[f])
        assert [int(x) for x in res[i]["cigar"][:w["n_cigar"]]] == w["cigar"], ("DP kernel CIGAR differs", i)
    print("smoke: DP service OK (%d requests equal ksw_extd2_sse)" % n_ok)
