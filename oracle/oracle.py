"""ctypes wrappers of the parity checkers -- TEST INFRASTRUCTURE ONLY.

  liboracle.so            oracle_place.c, the CPU restatement of the place stage
  _ref/libpanmap_ref.so   the reference's own aligner (src/mm_align.c + vendored minimap2),
                          compiled from /root/reference by oracle/Makefile

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libpanmap_ref.so")


def build(quiet=True):
    subprocess.run(["make", "-C", HERE, "all"], check=True, stdout=subprocess.DEVNULL if quiet else None)


_o = None


def olib():
    global _o
    if _o is None:
        if not os.path.exists(ORACLE_SO):
            build()
        L = C.CDLL(ORACLE_SO)
        vp, i64, i32, dbl = C.c_void_p, C.c_int64, C.c_int, C.c_double
        L.orc_chash.restype = C.c_uint64
        L.orc_chash.argtypes = [i32]
        L.orc_hash_seq.argtypes = [C.c_char_p, i32, vp, vp]
        L.orc_homopolymer_hash.restype = C.c_uint64
        L.orc_homopolymer_hash.argtypes = [i32, i32]
        L.orc_rolling_syncmers.restype = i64
        L.orc_rolling_syncmers.argtypes = [C.c_char_p, i64, i32, i32, i32, i32, i32, vp, vp, vp, vp]
        L.orc_read_seeds.restype = i64
        L.orc_read_seeds.argtypes = [C.c_char_p, i64, i32, i32, i32, i32, i32, i32, i32, vp]
        L.orc_hist_new.restype = vp
        L.orc_hist_free.argtypes = [vp]
        L.orc_hist_add.argtypes = [vp, C.c_uint64, i64]
        L.orc_hist_add_read.argtypes = [vp, C.c_char_p, i64, i32, i32, i32, i32, i32, i32, i32, i64]
        L.orc_hist_add_read_q.argtypes = [vp, C.c_char_p, C.c_char_p, i64, i64, i32, i32, i32, i32, i32, i32, i32, i32]
        L.orc_hist_add_reads.argtypes = [vp, vp, vp, i64, i64, i32, i32, i32, i32, i32, i32, i32]
        L.orc_hist_merge.argtypes = [vp, vp]
        L.orc_hist_build_mt.restype = vp
        L.orc_hist_build_mt.argtypes = [vp, vp, i64, i32, i32, i32, i32, i32, i32, i32, i32]
        L.orc_hist_size.restype = i64
        L.orc_hist_size.argtypes = [vp]
        L.orc_hist_export_sorted.argtypes = [vp, vp, vp]
        L.orc_finalize_reads.restype = i64
        L.orc_finalize_reads.argtypes = [vp, vp, i64, i32, dbl, i32, vp, vp, vp]
        L.orc_score_nodes.argtypes = [i64, vp, vp, vp, vp, vp, i64, vp, vp, dbl, dbl, vp, vp, vp, vp]
        L.orc_best_ties.argtypes = [i64, vp, vp, i32, vp, vp, vp, i64, vp]
        _o = L
    return _o


class ReadState(C.Structure):
    _fields_ = [("min_support", C.c_int64), ("n_unique_in", C.c_int64), ("n_kept", C.c_int64),
                ("total_freq", C.c_int64), ("log_magnitude", C.c_double), ("log_cont_den", C.c_double),
                ("est_coverage", C.c_double)]


def hash_seq(s: bytes):
    f, r = C.c_uint64(), C.c_uint64()
    rc = olib().orc_hash_seq(s, len(s), C.byref(f), C.byref(r))
    if rc != 0:
        raise ValueError("Kmer contains non canonical base")
    return f.value, r.value


def rolling_syncmers(seq: bytes, k, s, open_syncmer=False, t=0, return_all=True):
    """seeding::rollingSyncmers: list of (hash, isReverse, isSyncmer, startPos)."""
    n = max(len(seq) - k + 1, 0)
    if n == 0:
        return []
    h = np.zeros(n, np.uint64); r = np.zeros(n, np.uint8); sy = np.zeros(n, np.uint8); p = np.zeros(n, np.int64)
    m = olib().orc_rolling_syncmers(seq, len(seq), k, s, int(open_syncmer), t, int(return_all), h.ctypes.data, r.ctypes.data,
                                    sy.ctypes.data, p.ctypes.data)
    return [(int(h[i]), bool(r[i]), bool(sy[i]), int(p[i])) for i in range(m)]


def read_seeds(seq: bytes, k, s, l, open_syncmer=False, t=0, trim_start=0, trim_end=0):
    out = np.zeros(max(len(seq), 1), np.uint64)
    n = olib().orc_read_seeds(seq, len(seq), k, s, l, int(open_syncmer), t, trim_start, trim_end, out.ctypes.data)
    return out[:n].copy()


def histogram(reads, k, s, l, open_syncmer=False, t=0, trim_start=0, trim_end=0, dedup=False, quals=None, min_q=0):
    """(hash asc, count) of all read seeds.  dedup: each distinct sequence counted once
    (src/placement.cpp:1619-1620); otherwise counts are additive per read.  quals + min_q > 0: the
    quality-filtered branch (src/placement.cpp:1386-1527; no dedup there)."""
    L = olib()
    h = C.c_void_p(L.orc_hist_new())
    if quals is not None and min_q > 0:
        for r, q in zip(reads, quals):
            L.orc_hist_add_read_q(h, r, q, len(r), len(q), k, s, l, int(open_syncmer), t, trim_start, trim_end, min_q)
        it = []
    else:
        it = set(reads) if dedup else reads
    for r in it:
        L.orc_hist_add_read(h, r, len(r), k, s, l, int(open_syncmer), t, trim_start, trim_end, 1)
    n = L.orc_hist_size(h)
    hs, cn = np.zeros(n, np.uint64), np.zeros(n, np.int64)
    L.orc_hist_export_sorted(h, hs.ctypes.data, cn.ctypes.data)
    L.orc_hist_free(h)
    return hs, cn


def histogram_flat_mt(concat, off, k, s, l, n_threads, open_syncmer=False, t=0, trim_start=0, trim_end=0):
    """(hash asc, count) of the reads of one flat buffer, seeded on n_threads threads in ONE C call."""
    L = olib()
    concat = np.ascontiguousarray(concat, np.uint8); off = np.ascontiguousarray(off, np.int64)
    h = C.c_void_p(L.orc_hist_build_mt(concat.ctypes.data, off.ctypes.data, len(off) - 1, k, s, l, int(open_syncmer), t, trim_start, trim_end, n_threads))
    n = L.orc_hist_size(h)
    hs, cn = np.zeros(n, np.uint64), np.zeros(n, np.int64)
    L.orc_hist_export_sorted(h, hs.ctypes.data, cn.ctypes.data)
    L.orc_hist_free(h)
    return hs, cn


def finalize_reads(hs, cn, k, mask_fraction=0.0, min_support=-1):
    n = len(hs)
    hs = np.ascontiguousarray(hs, np.uint64); cn = np.ascontiguousarray(cn, np.int64)
    kh, kl = np.zeros(max(n, 1), np.uint64), np.zeros(max(n, 1), np.float64)
    st = ReadState()
    nk = olib().orc_finalize_reads(hs.ctypes.data, cn.ctypes.data, n, k, mask_fraction, min_support, kh.ctypes.data, kl.ctypes.data, C.byref(st))
    return kh[:nk].copy(), kl[:nk].copy(), st


def score_nodes(parent, offsets, ch_hash, ch_par, ch_child, kept_hash, kept_log, st: ReadState):
    n = len(parent)
    parent = np.ascontiguousarray(parent, np.uint32); offsets = np.ascontiguousarray(offsets, np.uint64)
    ch_hash = np.ascontiguousarray(ch_hash, np.uint64)
    ch_par = np.ascontiguousarray(ch_par, np.int16); ch_child = np.ascontiguousarray(ch_child, np.int16)
    kept_hash = np.ascontiguousarray(kept_hash, np.uint64); kept_log = np.ascontiguousarray(kept_log, np.float64)
    met, cts, sc = np.zeros((n, 5)), np.zeros((n, 2), np.int64), np.zeros((n, 5))
    wc = C.c_double()
    olib().orc_score_nodes(n, parent.ctypes.data, offsets.ctypes.data, ch_hash.ctypes.data, ch_par.ctypes.data, ch_child.ctypes.data,
                           len(kept_hash), kept_hash.ctypes.data, kept_log.ctypes.data, st.log_magnitude, st.log_cont_den,
                           C.byref(wc), met.ctypes.data, cts.ctypes.data, sc.ctypes.data)
    return sc, met, cts, wc.value


def best_ties(parent, scores, force_leaf=False, cap=0):
    n = len(parent)
    cap = max(cap, 2 * n + 2)     # (the tied list takes up to two entries per visited node before it is de-duplicated)
    parent = np.ascontiguousarray(parent, np.uint32); scores = np.ascontiguousarray(scores, np.float64)
    best = np.zeros(5); idx = np.zeros(5, np.uint32); ties = np.zeros((5, cap), np.uint32); nt = np.zeros(5, np.int64)
    olib().orc_best_ties(n, parent.ctypes.data, scores.ctypes.data, int(force_leaf), best.ctypes.data, idx.ctypes.data, ties.ctypes.data, cap,
                         nt.ctypes.data)
    return best, idx, [ties[m, :nt[m]].copy() for m in range(5)]


def place(reads, idx_arrays, k, s, l, open_syncmer=False, t=0, trim_start=0, trim_end=0, mask_fraction=0.0, min_support=-1,
          force_leaf=False, dedup=False, quals=None, min_q=0):
    """Whole place stage on the CPU oracle.  Returns dict with everything the GPU path reports."""
    hs, cn = histogram(reads, k, s, l, open_syncmer, t, trim_start, trim_end, dedup, quals, min_q)
    kh, kl, st = finalize_reads(hs, cn, k, mask_fraction, min_support)
    sc, met, cts, wc = score_nodes(idx_arrays["parent"], idx_arrays["offsets"], idx_arrays["hash"], idx_arrays["parent_count"],
                                   idx_arrays["child_count"], kh, kl, st)
    best, bidx, ties = best_ties(idx_arrays["parent"], sc, force_leaf)
    return dict(hist_hash=hs, hist_count=cn, kept_hash=kh, kept_log=kl, state=st, scores=sc, metrics=met, counts=cts, wc_den=wc,
                best=best, best_idx=bidx, ties=ties)


# ----------------------------------------------------------------------- reference aligner
class ReadAlign(C.Structure):
    _fields_ = [("pos", C.c_int32), ("rs", C.c_int32), ("re", C.c_int32), ("qs", C.c_int32), ("qe", C.c_int32),
                ("mapq", C.c_uint8), ("rev", C.c_uint8), ("proper_frag", C.c_uint8),
                ("n_cigar", C.c_int32), ("cigar", C.POINTER(C.c_uint32)), ("md", C.c_char_p)]


class AlignPairResult(C.Structure):
    _fields_ = [("r1", ReadAlign), ("r2", ReadAlign), ("mapped", C.c_int)]


_r = None


def rlib():
    global _r
    if _r is None:
        if not os.path.exists(REF_SO):
            build()
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO + " (reference tree absent and no prebuilt copy)")
        L = C.CDLL(REF_SO)
        L.align_reads_direct.restype = None
        L.align_reads_direct.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p),
                                         C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(AlignPairResult), C.c_bool, C.c_int]
        _r = L
    return _r


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def _unpack(ra: ReadAlign):
    cig = [ra.cigar[i] for i in range(ra.n_cigar)] if ra.cigar else []
    if ra.cigar:
        _libc.free(C.cast(ra.cigar, C.c_void_p))
    return dict(pos=ra.pos, rs=ra.rs, re=ra.re, qs=ra.qs, qe=ra.qe, mapq=ra.mapq, rev=ra.rev, proper_frag=ra.proper_frag, cigar=cig)


def prepare_align_call(reads, paired: bool):
    """marshal the arguments of align_reads_direct once (outside any timed span)"""
    n = len(reads)
    arr = (C.c_char_p * n)(*reads)
    quals = (C.c_char_p * n)(*[b"I" * len(r) for r in reads])
    names = (C.c_char_p * n)(*[b"r%d" % i for i in range(n)])
    lens = (C.c_int * n)(*[len(r) for r in reads])
    n_res = n // 2 if paired else n
    res = (AlignPairResult * max(n_res, 1))()
    return dict(n=n, arr=arr, quals=quals, names=names, lens=lens, n_res=n_res, res=res, paired=paired)


def prepare_align_call_flat(concat, off, paired: bool, revcomp_mate2: bool):
    """prepare_align_call from one flat read buffer without a Python object per read (oracle_alnflat.c): mate 2 reverse-
    complemented on the way when asked (the orientation readFastqPaired hands to align_reads_direct)"""
    L = olib()
    L.orc_align_prepare_reads.restype = None
    L.orc_align_prepare_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    concat = np.ascontiguousarray(concat, np.uint8); off = np.ascontiguousarray(off, np.int64)
    n = len(off) - 1
    base = int(off[0])
    buf = np.zeros(int(off[-1]) - base + n + 1, np.uint8)
    arr = (C.c_char_p * max(n, 1))()
    lens = (C.c_int * max(n, 1))()
    L.orc_align_prepare_reads(concat.ctypes.data, off.ctypes.data, n, int(bool(paired and revcomp_mate2)), buf.ctypes.data, C.addressof(arr), C.addressof(lens))
    # qualities / names are accepted but unused by the minimap2 backend (src/mm_align.h:44-53): every entry points at one
    # shared string (pointer arrays filled through numpy, no Python object per read)
    q = C.create_string_buffer(b"I" * (int(np.max(np.diff(off))) if n else 1))
    nm = C.create_string_buffer(b"r")
    quals = (C.c_char_p * max(n, 1))()
    names = (C.c_char_p * max(n, 1))()
    fill_q = np.full(max(n, 1), C.addressof(q), np.uint64)      # (named: a temporary would be freed before memmove reads it)
    fill_n = np.full(max(n, 1), C.addressof(nm), np.uint64)
    C.memmove(C.addressof(quals), fill_q.ctypes.data, 8 * max(n, 1))
    C.memmove(C.addressof(names), fill_n.ctypes.data, 8 * max(n, 1))
    del fill_q, fill_n
    n_res = n // 2 if paired else n
    res = (AlignPairResult * max(n_res, 1))()
    return dict(n=n, arr=arr, quals=quals, names=names, lens=lens, n_res=n_res, res=res, paired=paired, _keep=(buf, q, nm, concat, off))


def run_align_call(fn, reference: bytes, prep, n_threads=1):
    """the bare C call"""
    fn(reference, b"ref", prep["n"], prep["arr"], prep["quals"], prep["names"], prep["lens"], prep["res"], prep["paired"], n_threads)


def unpack_align_call(prep):
    res, paired = prep["res"], prep["paired"]
    return [dict(mapped=res[i].mapped, r1=_unpack(res[i].r1), r2=_unpack(res[i].r2) if paired else None) for i in range(prep["n_res"])]


def flatten_align_call(prep):
    """the results of run_align_call as flat numpy arrays (oracle_alnflat.c), CIGARs freed -- for comparisons at 10^7 reads,
    where unpack_align_call's list of dicts is too slow:
    dict(fields int32 [n_reads][8] = pos rs re qs qe mapq rev proper_frag, n_cigar int32 [n_reads], mapped u8 [n_res],
         cigar u32 arena in record order)"""
    L = olib()
    L.orc_align_cigar_total.restype = C.c_int64
    L.orc_align_cigar_total.argtypes = [C.c_void_p, C.c_int64, C.c_int]
    L.orc_align_flatten.restype = C.c_int64
    L.orc_align_flatten.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    res, paired, n_res, n = prep["res"], prep["paired"], prep["n_res"], prep["n"]
    assert C.sizeof(AlignPairResult) == 104
    tot = int(L.orc_align_cigar_total(C.addressof(res), n_res, int(paired)))
    n_reads = n_res * (2 if paired else 1)
    fields = np.zeros((max(n_reads, 1), 8), np.int32)
    n_cig = np.zeros(max(n_reads, 1), np.int32)
    mapped = np.zeros(max(n_res, 1), np.uint8)
    arena = np.zeros(max(tot, 1), np.uint32)
    used = int(L.orc_align_flatten(C.addressof(res), n_res, int(paired), fields.ctypes.data, n_cig.ctypes.data, mapped.ctypes.data, arena.ctypes.data, tot))
    assert used == tot, (used, tot)
    return dict(fields=fields[:n_reads], n_cigar=n_cig[:n_reads], mapped=mapped[:n_res], cigar=arena[:tot], paired=paired)


def call_align_reads_direct(fn, reference: bytes, reads, paired: bool, n_threads=1):
    """Call an align_reads_direct-compatible entry point (the reference's or the product's)."""
    prep = prepare_align_call(reads, paired)
    run_align_call(fn, reference, prep, n_threads)
    return unpack_align_call(prep)


def ref_align_reads_direct(reference: bytes, reads, paired: bool, n_threads=1):
    return call_align_reads_direct(rlib().align_reads_direct, reference, reads, paired, n_threads)


def ref_score_reads(reference: bytes, reads, paired: bool) -> int:
    """score_reads_vs_reference of the compiled reference (src/mm_align.c:144-199): minus the total edit distance of the
    reads' first regions against `reference` (the alignment score of one --refine candidate, src/placement.cpp:489-514)"""
    L = rlib()
    L.score_reads_vs_reference.restype = C.c_int64
    L.score_reads_vs_reference.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int, C.c_bool]
    n = len(reads)
    arr = (C.c_char_p * n)(*reads)
    lens = (C.c_int * n)(*[len(r) for r in reads])
    return int(L.score_reads_vs_reference(reference, n, arr, lens, 0, bool(paired)))


# ------------------------------------------------------------------ the reference's DP kernel on its own (A9)
class _KswExtz(C.Structure):   # ksw_extz_t (src/3rdparty/minimap2/ksw2.h:27-36)
    _fields_ = [("max_zdropped", C.c_uint32), ("max_q", C.c_int), ("max_t", C.c_int), ("mqe", C.c_int), ("mqe_t", C.c_int), ("mte", C.c_int),
                ("mte_q", C.c_int), ("score", C.c_int), ("m_cigar", C.c_int), ("n_cigar", C.c_int), ("reach_end", C.c_int),
                ("cigar", C.POINTER(C.c_uint32))]


def simple_mat(a: int, b: int, sc_ambi: int):
    """ksw_gen_simple_mat(5, ...) (src/3rdparty/minimap2/options.c / ksw2.h): the 5 x 5 score matrix minimap2 aligns with"""
    a, b, sc_ambi = abs(a), -abs(b), -abs(sc_ambi)
    m = [[(a if i == j else b) if i < 4 and j < 4 else sc_ambi for j in range(5)] for i in range(5)]
    return (C.c_int8 * 25)(*[v for row in m for v in row])


def ref_ksw_extd2(query, target, mat, q, e, q2, e2, w, zdrop, end_bonus, flag):
    """ksw_extd2_sse of the compiled reference (src/3rdparty/minimap2/ksw2_extd2_sse.c:28-400) on nt4 codes.
    -> dict of the ksw_extz_t fields, cigar as a list"""
    L = rlib()
    fn = L.ksw_extd2_sse
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int8, C.POINTER(C.c_int8), C.c_int8, C.c_int8, C.c_int8, C.c_int8,
                   C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_KswExtz)]
    ez = _KswExtz()
    qb, tb = bytes(bytearray(query)), bytes(bytearray(target))
    fn(None, len(qb), qb, len(tb), tb, 5, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, C.byref(ez))
    out = dict(max=ez.max_zdropped & 0x7fffffff, zdropped=ez.max_zdropped >> 31, max_q=ez.max_q, max_t=ez.max_t, mqe=ez.mqe, mqe_t=ez.mqe_t,
               mte=ez.mte, mte_q=ez.mte_q, score=ez.score, n_cigar=ez.n_cigar, reach_end=ez.reach_end,
               cigar=[int(ez.cigar[i]) for i in range(ez.n_cigar)])
    if ez.cigar:
        C.CDLL(None).free(ez.cigar)
    return out
