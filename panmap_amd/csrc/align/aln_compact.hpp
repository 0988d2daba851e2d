// ALIGN stage, COMPACT tier: the short-read pair pipeline with its whole per-pair work state in LDS and registers.
//
// The general pipeline (aln_map.hpp) keeps minimap2's data shapes -- 16-byte anchors, 16-byte chain cells, 108-byte
// region records, byte-per-base sequences -- about 10 KB of work state per pair, which in the thread-per-pair kernel
// lives in an HBM arena and is what that kernel moves (23-35 GB per launch of 500k pairs against 93 MB of input and
// output, profiles/r01).  This tier restates the SAME decisions for the pairs that make up a short-read batch on a small
// genome, with the state packed to 13 bytes per anchor:
//   * reads stay 2 bit/base where the host left them (no decode pass); mismatches against the reference come from
//     XOR + popcount on 32-base words, once per mate and diagonal;
//   * a minimizer is 8 bytes while it waits for its index probe, a seed 6 bytes (reference position word + query word),
//     an anchor 6 bytes (strand | reference position, query position | segment | flags), a chain cell 2 bytes
//     (score | predecessor), the merge heap 1 byte per entry;
//   * 48 anchors per pair (99.2 % of 150 bp pairs; mean 40): 624 bytes per pair, interleaved word-wise across the 64
//     lanes of a wave in LDS (conflict-free), 39 KB per wave;
//   * regions are a handful of scalars in registers: at most one region per mate is followed.
// Everything outside that envelope -- an ambiguous base, a seed that occurs twice in the reference, more than 48 seeds,
// a third chain, two regions on one mate, an extension or gap fill the closed forms (aln_ksw.hpp, ksw_shortcut_*) do
// not answer -- makes the pair BAIL: nothing is written and the general thread-per-pair tier (which posts DP requests,
// splits regions, ...) runs it.  A pair this tier finishes gets exactly the record the general pipeline computes; the
// parity tests run both against the compiled reference aligner.
//
// Reference behaviour restated here (file:line under src/3rdparty/minimap2 unless noted):
//   sketch.c:77-143 (through sketch_core), index.c:81-99, seed.c:98-131, map.c:102-166 (heap merge, ksort.h:43-59),
//   lchain.c:113-230 + :9-76 (chain fill / backtrack), hit.c:8-40, 54-94, 345-400 (region coordinates, per-mate split),
//   align.c:355-502, 575-833 (end fixing, bad-seed filters, extension / fill driver), :240-289 (alignment statistics),
//   hit.c:301-322, 421-466 (filter, mapq), pe.c:76-177 (pairing), src/mm_align.c:271-354 (record).
#pragma once
#include "aln_align.hpp"
#include "aln_chain.hpp"
#include "aln_ksw.hpp"
#include "aln_seed.hpp"
#include "aln_compact_defs.hpp"
#include "aln_types.hpp"

namespace pmx {
namespace aln {

#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) uint32_t c_u32;
typedef __attribute__((address_space(3))) uint16_t c_u16;
typedef __attribute__((address_space(3))) uint8_t c_u8;
#define PMX_C_STRIDE 64
#else
typedef uint32_t c_u32;
typedef uint16_t c_u16;
typedef uint8_t c_u8;
#define PMX_C_STRIDE 1
#endif

// one lane's 156 words; word i of lane l sits at base + i * 64 + l (device), so a wave touching word i of its 64
// pairs touches 256 contiguous bytes of LDS
struct CMem {
    c_u32* base;
    PMX_HD c_u32& w(int i) const { return base[i * PMX_C_STRIDE]; }
    PMX_HD c_u16& h(int i) const { return ((c_u16*)(base + (i >> 1) * PMX_C_STRIDE))[i & 1]; }
    PMX_HD c_u8& b(int i) const { return ((c_u8*)(base + (i >> 2) * PMX_C_STRIDE))[i & 3]; }
    // regions (never live together when they overlap):
    //   words   0..47   P  seed: reference position word      | halves 0..47 F chain cells | SX per-mate anchor x
    //   words  48..71   Q  seed: query word (u16)             | Z backtrack sort keys       | SY per-mate anchor y
    //   words  72..119  AX anchor x                           | M minimizers of one read (u64 x 40: words 72..151)
    //   words 120..143  AY anchor y (u16)
    //   words 144..155  HI merge heap (u8)                    | V chain members in walk order
    PMX_HD c_u32& P(int i) const { return w(i); }
    PMX_HD c_u16& F(int i) const { return h(i); }
    PMX_HD c_u32& SX(int i) const { return w(i); }
    PMX_HD c_u16& Q(int i) const { return h(96 + i); }
    PMX_HD c_u16& Z(int i) const { return h(96 + i); }
    PMX_HD c_u16& SY(int i) const { return h(96 + i); }
    PMX_HD c_u32& AX(int i) const { return w(72 + i); }
    PMX_HD c_u16& AY(int i) const { return h(240 + i); }
    PMX_HD c_u8& HI(int i) const { return b(576 + i); }
    PMX_HD c_u8& V(int i) const { return b(576 + i); }
    PMX_HD uint64_t M(int i) const { return (uint64_t)w(72 + 2 * i) | (uint64_t)w(73 + 2 * i) << 32; }
    PMX_HD void setM(int i, uint64_t v) const { w(72 + 2 * i) = (uint32_t)v; w(73 + 2 * i) = (uint32_t)(v >> 32); }
};

// query word of a seed / anchor: bits 0..9 position (seed: position << 1 | strand), 10 segment, 11 tandem, 12 ignore
#define PMX_CQ_SEG 0x400u
#define PMX_CQ_TANDEM 0x800u
#define PMX_CQ_IGNORE 0x1000u
#define PMX_CQ_LONG_JOIN 0x2000u

PMX_HD uint64_t c_bitrev64(uint64_t x) {
#if defined(__clang__)
    return __builtin_bitreverse64(x);
#else
    x = ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0f0f0f0f0f0f0f0fULL) | ((x & 0x0f0f0f0f0f0f0f0fULL) << 4);
    return __builtin_bswap64(x);
#endif
}

// A read as the host packed it (k_pack_reads: base j of word k at bits 2(j%32), A C G T = 0 1 2 3), and how the
// aligner sees it: orientation 0 = the read handed to align_reads_direct (mate 2 reverse-complemented on the fly when
// it arrives in FASTQ orientation: flip), orientation 1 = its reverse complement (what a reverse-strand region aligns).
struct CRead {
    const uint64_t* w;
    int len;
    bool flip;
};
// 32 bases [start, start + 32) of a packed sequence of `len` bases; positions outside [0, len) read as 0
PMX_HD uint64_t c_chunk(const uint64_t* w, int len, int start) {
    const int nw = (len + 31) >> 5;
    const int k = start >> 5;                 // arithmetic shift: floor for negative starts
    const int sh = (start & 31) * 2;
    uint64_t lo = 0, hi = 0;
    if (k >= 0 && k < nw) lo = w[k];
    if (k + 1 >= 0 && k + 1 < nw) hi = w[k + 1];
    uint64_t v = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
    const int valid = len - start;            // bases of this chunk that lie before the end
    if (valid < 32) v = valid <= 0 ? 0 : (v & ((1ULL << (2 * valid)) - 1ULL));
    return v;
}
PMX_HD uint64_t c_revcomp_chunk(uint64_t x) {   // the 32 bases in reverse order, complemented
    x = c_bitrev64(x);
    x = ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
    return ~x;
}
PMX_HD uint64_t c_read_chunk(const CRead& r, int orient, int pos) {
    if ((orient != 0) == r.flip) return c_chunk(r.w, r.len, pos);
    return c_revcomp_chunk(c_chunk(r.w, r.len, r.len - 32 - pos));   // bases pos.. = complement of stored len-1-pos, len-2-pos, ..
}
PMX_HD uint32_t c_read_base(const CRead& r, int orient, int i) { return (uint32_t)(c_read_chunk(r, orient, i) & 3ULL); }

// mismatch masks of one mate (in the orientation it aligns in) against the reference along ONE diagonal:
// bit 2j of word c = query base 32c + j differs from reference base 32c + j + diag (or the reference base is ambiguous)
struct CDiag {
    uint64_t mm[PMX_C_NW];
    uint64_t amb[PMX_C_NW];
};
PMX_HD uint64_t c_range_mask(int c, int lo, int hi) {
    int a = lo - 32 * c, b = hi - 32 * c;
    if (a < 0) a = 0;
    if (b > 32) b = 32;
    if (b <= a) return 0;
    const uint64_t upto_b = b == 32 ? ~0ULL : ((1ULL << (2 * b)) - 1ULL);
    const uint64_t below_a = (1ULL << (2 * a)) - 1ULL;   // a < 32 here
    return upto_b & ~below_a & 0x5555555555555555ULL;
}
PMX_HD int c_count(const uint64_t* m, int lo, int hi) {
    int n = 0;
#pragma unroll
    for (int c = 0; c < PMX_C_NW; ++c) n += __builtin_popcountll(m[c] & c_range_mask(c, lo, hi));
    return n;
}
PMX_HD int c_first(const uint64_t* m, int lo, int hi) {   // smallest position in [lo, hi) whose bit is set, -1 if none
    int r = -1;
#pragma unroll
    for (int c = PMX_C_NW - 1; c >= 0; --c) {
        const uint64_t v = m[c] & c_range_mask(c, lo, hi);
        if (v) r = 32 * c + (__builtin_ctzll(v) >> 1);
    }
    return r;
}
PMX_HD int c_last(const uint64_t* m, int lo, int hi) {
    int r = -1;
#pragma unroll
    for (int c = 0; c < PMX_C_NW; ++c) {
        const uint64_t v = m[c] & c_range_mask(c, lo, hi);
        if (v) r = 32 * c + ((63 - __builtin_clzll(v)) >> 1);
    }
    return r;
}

// what the tier hands back for one pair
struct CMate {
    int32_t rs, re, qs, qe, dp_max;
    uint32_t cigar;       // the single CIGAR operation (n_cigar == 1 in this tier)
    uint8_t mapq, rev, proper_frag, has_aln;
};
struct CResult {
    int mapped;
    CMate m[2];
};
#define PMX_C_DONE 0
#define PMX_C_BAIL 1

// one region of one mate while it is aligned (the fields of mm_reg1_t this tier can reach)
struct CReg {
    int32_t cnt, score, rev, qs, qe, rs, re, mlen, blen, dp_score, dp_max, has_p, mapq, proper_frag, m_len;
};

// 64-bit anchor x as the reference holds it (strand in bit 63, one reference sequence: rid = 0)
PMX_HD uint64_t c_x64(uint32_t ax) { return (uint64_t)(ax >> 31) << 63 | (uint64_t)(ax & 0x7fffffffu); }

// mm_cal_fuzzy_len + mm_reg_set_coor (hit.c:8-40) on a per-mate anchor list
PMX_HD void c_reg_set_coor(const CMem& m, int base, CReg& r, int32_t qlen, int span) {
    const int32_t x0 = (int32_t)(m.SX(base) & 0x7fffffffu), y0 = (int32_t)(m.SY(base) & 0x3ffu);
    const int32_t xl = (int32_t)(m.SX(base + r.cnt - 1) & 0x7fffffffu), yl = (int32_t)(m.SY(base + r.cnt - 1) & 0x3ffu);
    r.rs = x0 + 1 > span ? x0 + 1 - span : 0;
    r.re = xl + 1;
    if (!r.rev) { r.qs = y0 + 1 - span; r.qe = yl + 1; }
    else { r.qs = qlen - (yl + 1); r.qe = qlen - (y0 + 1 - span); }
    r.mlen = r.blen = span;
    int32_t px = x0, py = y0;
    for (int i = 1; i < r.cnt; ++i) {
        const int32_t x = (int32_t)(m.SX(base + i) & 0x7fffffffu), y = (int32_t)(m.SY(base + i) & 0x3ffu);
        const int tl = x - px, ql = y - py;
        r.blen += tl > ql ? tl : ql;
        r.mlen += tl > span && ql > span ? span : tl < ql ? tl : ql;
        px = x; py = y;
    }
}

// mm_fix_bad_ends (align.c:464-502) on a per-mate list (r.as == 0)
PMX_HD void c_fix_bad_ends(const CMem& m, int base, const CReg& r, int span, int bw, int min_match, int32_t* as, int32_t* cnt) {
    *as = 0;
    *cnt = r.cnt;
    if (r.cnt < 3) return;
    int32_t mm_ = span, l = span;
    for (int32_t i = 1; i < r.cnt - 1; ++i) {
        if (m.SY(base + i) & PMX_CQ_LONG_JOIN) break;
        const int32_t lr = (int32_t)(m.SX(base + i) & 0x7fffffffu) - (int32_t)(m.SX(base + i - 1) & 0x7fffffffu);
        const int32_t lq = (int32_t)(m.SY(base + i) & 0x3ffu) - (int32_t)(m.SY(base + i - 1) & 0x3ffu);
        const int32_t mn = lr < lq ? lr : lq, mx = lr > lq ? lr : lq;
        if (mx - mn > l >> 1) *as = i;
        l += mn;
        mm_ += mn < span ? mn : span;
        if (l >= bw << 1 || (mm_ >= min_match && mm_ >= bw) || mm_ >= r.mlen >> 1) break;
    }
    *cnt = r.cnt - *as;
    mm_ = l = span;
    for (int32_t i = r.cnt - 2; i > *as; --i) {
        if (m.SY(base + i + 1) & PMX_CQ_LONG_JOIN) break;
        const int32_t lr = (int32_t)(m.SX(base + i + 1) & 0x7fffffffu) - (int32_t)(m.SX(base + i) & 0x7fffffffu);
        const int32_t lq = (int32_t)(m.SY(base + i + 1) & 0x3ffu) - (int32_t)(m.SY(base + i) & 0x3ffu);
        const int32_t mn = lr < lq ? lr : lq, mx = lr > lq ? lr : lq;
        if (mx - mn > l >> 1) *cnt = i + 1 - *as;
        l += mn;
        mm_ += mn < span ? mn : span;
        if (l >= bw << 1 || (mm_ >= min_match && mm_ >= bw) || mm_ >= r.mlen >> 1) break;
    }
}

// base readers for the (1c) probe of ksw_shortcut_ext_decide: position i of the extension = query base qa + qstep * i
// of the mate (orientation `orient`) / reference base ta + qstep * i
struct CQryFn {
    const CRead& r;
    int orient, at, step;
    PMX_HD uint32_t operator()(int i) const { return c_read_base(r, orient, at + step * i); }
};
struct CRefFn {
    const RefIndex& ri;
    int at, step;
    PMX_HD uint32_t operator()(int i) const { return ri.seq[at + step * i]; }
};

// mm_align1 (align.c:575-833) for a region that is its mate's only one (as == 0, n_a == cnt): every extension / fill is
// answered by the closed forms or the pair bails.  All three lie on the diagonal of the first kept anchor, so one
// mismatch mask per mate serves them and the statistics pass.
PMX_HD int c_align1(const CMem& m, int base, const Opt& o, const RefIndex& ri, const CRead& rd, int qlen, CReg& r) {
    const int span = o.k;
    const int32_t rev = r.rev;
    const int32_t ref_len = ri.len;
    const int bw = (int)(o.bw * 1.5 + 1.);
    int bw_long = (int)(o.bw_long * 1.5 + 1.);
    if (bw_long < bw) bw_long = bw;
    int32_t as1, cnt1;
    c_fix_bad_ends(m, base, r, span, o.bw, o.min_chain_score * 2, &as1, &cnt1);
    {   // mm_filter_bad_seeds / _alt (align.c:391-462) act only when two or more anchor steps change the diagonal by more
        // than 10 (their min_gap; the second filter's 30 is implied): not followed here
        int n_long = 0;
        for (int i = 1; i < cnt1; ++i) {
            const int gap = ((int32_t)(m.SY(base + as1 + i) & 0x3ffu) - (int32_t)(m.SY(base + as1 + i - 1) & 0x3ffu)) -
                            ((int32_t)(m.SX(base + as1 + i) & 0x7fffffffu) - (int32_t)(m.SX(base + as1 + i - 1) & 0x7fffffffu));
            if (gap < -10 || gap > 10) ++n_long;
        }
        if (n_long > 1) return PMX_C_BAIL;
    }
    auto ax = [&](int i) { return (int32_t)(m.SX(base + i) & 0x7fffffffu); };
    auto ay = [&](int i) { return (int32_t)(m.SY(base + i) & 0x3ffu); };
    int32_t rs = ax(as1) - (o.k >> 1), qs = ay(as1) - (o.k >> 1);                           // mm_adjust_minier, non-HPC
    int32_t re = ax(as1 + cnt1 - 1) - (o.k >> 1), qe = ay(as1 + cnt1 - 1) - (o.k >> 1);
    int32_t l, rs0, re0, qs0, qe0, rs1, qs1, re1, qe1;
    // region to align (align.c:636-691); the region is the only one of its list: no neighbouring anchors to stop at
    rs0 = ax(0) + 1 - span;
    qs0 = ay(0) + 1 - span;
    if (rs0 < 0) rs0 = 0;
    rs1 = qs1 = 0;
    if (qs > 0 && rs > 0) {
        l = qs < o.max_gap ? qs : o.max_gap;
        qs1 = qs1 > qs - l ? qs1 : qs - l;
        qs0 = qs0 < qs1 ? qs0 : qs1;
        l += l * o.a > o.q ? (l * o.a - o.q) / o.e : 0;
        l = l < o.max_gap ? l : o.max_gap;
        l = l < rs ? l : rs;
        rs1 = rs1 > rs - l ? rs1 : rs - l;
        rs0 = rs0 < rs1 ? rs0 : rs1;
        rs0 = rs0 < rs ? rs0 : rs;
    } else { rs0 = rs; qs0 = qs; }
    re0 = ax(r.cnt - 1) + 1;
    qe0 = ay(r.cnt - 1) + 1;
    re1 = ref_len; qe1 = qlen;
    if (qe < qlen && re < ref_len) {
        l = qlen - qe < o.max_gap ? qlen - qe : o.max_gap;
        qe1 = qe1 < qe + l ? qe1 : qe + l;
        qe0 = qe0 > qe1 ? qe0 : qe1;
        l += l * o.a > o.q ? (l * o.a - o.q) / o.e : 0;
        l = l < o.max_gap ? l : o.max_gap;
        l = l < ref_len - re ? l : ref_len - re;
        re1 = re1 < re + l ? re1 : re + l;
        re0 = re0 > re1 ? re0 : re1;
    } else { re0 = re; qe0 = qe; }
    if (re0 <= rs0 || qs0 < 0 || qe0 > qlen) return PMX_C_BAIL;

    // one mismatch mask for the whole mate along the diagonal of the first kept anchor
    const int diag = rs - qs;
    CDiag D;
#pragma unroll
    for (int c = 0; c < PMX_C_NW; ++c) {
        const uint64_t qw = c_read_chunk(rd, rev, 32 * c);
        const uint64_t tw = c_chunk(ri.pk, ri.len, 32 * c + diag);
        const uint64_t x = qw ^ tw;
        D.amb[c] = c_chunk(ri.pk_amb, ri.len, 32 * c + diag);
        D.mm[c] = ((x | x >> 1) & 0x5555555555555555ULL) | D.amb[c];
    }
    if (c_count(D.amb, qs0, qe0) != 0) return PMX_C_BAIL;   // ambiguous reference base: the closed forms do not apply

    const int a = o.mat[0], b = -o.mat[1];
    const int g1 = o.q + o.e, g2 = o.q2 + o.e2;
    const int gmin = g1 < g2 ? g1 : g2, gmax = g1 > g2 ? g1 : g2;
    Ez ez;
    uint32_t cig0 = 0;
    int32_t m_total = 0;    // the CIGAR: one run of M
    r.has_p = 0; r.dp_score = 0;

    if (qs > 0 && rs > 0) {   // left extension (align.c:704-722): both sequences read backwards from (qs, rs)
        const int ql = qs - qs0, tl = rs - rs0;
        const int zdrop = o.zdrop;   // (split_inv regions never reach this tier)
        if (!ksw_shortcut_applicable(ql, tl, a, b, gmin, bw) || !ksw_shortcut_is_ext(ql, tl, a, b, gmax, zdrop, PMX_EZ_EXTZ_ONLY)) return PMX_C_BAIL;
        const int d = c_count(D.mm, qs0, qs);
        int pf = INT32_MAX, pm = -1;
        if (d > 0) { pm = qs - 1 - c_first(D.mm, qs0, qs); pf = qs - 1 - c_last(D.mm, qs0, qs); }
        CQryFn qf{rd, rev, qs - 1, -1};
        CRefFn tf{ri, rs - 1, -1};
        if (!ksw_shortcut_ext_decide(ql, tl, d, pf, pm, qf, tf, a, b, (int8_t)o.q, (int8_t)o.e, (int8_t)o.q2, (int8_t)o.e2, zdrop, o.end_bonus, ez, &cig0))
            return PMX_C_BAIL;
        if (ez.n_cigar > 0) { r.has_p = 1; m_total += (int32_t)(cig0 >> 4); r.dp_score += (int32_t)ez.max; }
        rs1 = rs - (ez.reach_end ? ez.mqe_t + 1 : ez.max_t + 1);
        qs1 = qs - (ez.reach_end ? qs - qs0 : ez.max_q + 1);
    } else { rs1 = rs; qs1 = qs; }
    re1 = rs; qe1 = qs;

    // gap filling (align.c:727-797): reads of this tier are shorter than min_ksw_len, so the walk closes exactly one fill,
    // at the last kept anchor (flagged anchors are only skipped on the way)
    if (cnt1 > 1) {
        if (qlen >= o.min_ksw_len) return PMX_C_BAIL;
        if (m.SY(base + as1 + cnt1 - 1) & PMX_CQ_LONG_JOIN) return PMX_C_BAIL;
        re1 = re; qe1 = qe;
        const int ql = qe - qs, tl = re - rs;
        if (!ksw_shortcut_applicable(ql, tl, a, b, gmin, bw_long) || !ksw_shortcut_is_fill(ql, tl, PMX_EZ_APPROX_MAX)) return PMX_C_BAIL;
        const int d = c_count(D.mm, qs, qe);
        if (!ksw_shortcut_fill_decide(ql, d, 0, a, b, gmin, PMX_EZ_APPROX_MAX, ez, &cig0)) return PMX_C_BAIL;
        // (the fill is gap-free with at most three mismatches: mm_test_zdrop cannot fire, see align1)
        if (!(4 * (o.a + o.b) <= o.zdrop && 4 * (o.a + o.b) <= o.zdrop_inv)) return PMX_C_BAIL;
        r.has_p = 1;
        m_total += (int32_t)(cig0 >> 4);
        r.dp_score += ez.score;
        rs = re; qs = qe;
    }

    if (qe < qe0 && re < re0) {   // right extension (align.c:799-815)
        const int ql = qe0 - qe, tl = re0 - re;
        if (!ksw_shortcut_applicable(ql, tl, a, b, gmin, bw) || !ksw_shortcut_is_ext(ql, tl, a, b, gmax, o.zdrop, PMX_EZ_EXTZ_ONLY)) return PMX_C_BAIL;
        const int d = c_count(D.mm, qe, qe0);
        int pf = INT32_MAX, pm = -1;
        if (d > 0) { pf = c_first(D.mm, qe, qe0) - qe; pm = c_last(D.mm, qe, qe0) - qe; }
        CQryFn qf{rd, rev, qe, 1};
        CRefFn tf{ri, re, 1};
        if (!ksw_shortcut_ext_decide(ql, tl, d, pf, pm, qf, tf, a, b, (int8_t)o.q, (int8_t)o.e, (int8_t)o.q2, (int8_t)o.e2, o.zdrop, o.end_bonus, ez, &cig0))
            return PMX_C_BAIL;
        if (ez.n_cigar > 0) { r.has_p = 1; m_total += (int32_t)(cig0 >> 4); r.dp_score += (int32_t)ez.max; }
        re1 = re + (ez.reach_end ? ez.mqe_t + 1 : ez.max_t + 1);
        qe1 = qe + (ez.reach_end ? qe0 - qe : ez.max_q + 1);
    }

    r.rs = rs1; r.re = re1;
    if (!rev) { r.qs = qs1; r.qe = qe1; }
    else { r.qs = qlen - qe1; r.qe = qlen - qs1; }
    if (!r.has_p || m_total != qe1 - qs1 || m_total != re1 - rs1 || m_total <= 0) return PMX_C_BAIL;
    r.m_len = m_total;

    // mm_update_extra (align.c:240-289) for one run of M over [qs1, qe1): between mismatches the running score only rises,
    // so the maximum is taken at the end of every matching stretch
    {
        double s = 0.0, mx = 0.0;
        int pos = qs1, n_diff = 0;
        while (pos < qe1) {
            const int nx = c_first(D.mm, pos, qe1);
            const int stop = nx < 0 ? qe1 : nx;
            if (stop > pos) { s += (double)a * (stop - pos); mx = mx > s ? mx : s; }
            if (nx >= 0) {
                ++n_diff;
                s -= b;
                if (s < 0) s = 0;
                else mx = mx > s ? mx : s;
            }
            pos = stop + 1;
        }
        r.blen = m_total;
        r.mlen = m_total - n_diff;
        r.dp_max = (int32_t)(mx + .499);
    }
    return PMX_C_DONE;
}

// The pair.  `in` = the two mates as packed by the host; `out` is only meaningful when PMX_C_DONE is returned.
#if defined(__HIP_DEVICE_COMPILE__)
#define PMX_C_STAMP(k) do { if (prof) { const unsigned long long t_ = (unsigned long long)clock64(); prof[k] += t_ - prof_t; prof_t = t_; } } while (0)
#else
#define PMX_C_STAMP(k) ((void)0)
#endif
// prof: NULL, or 8 per-lane cycle accumulators (sketch, probes, merge, chain fill, backtrack, regions, align + mapq, pairing)
PMX_HDN int compact_map_pair(const CMem& m, const Opt& o, const RefIndex& ri, const CRead* rd, const uint32_t* const* amb, CResult& out,
                             unsigned long long* prof = nullptr) {
    out.mapped = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long prof_t = prof ? (unsigned long long)clock64() : 0ULL;
#endif
    const int k = o.k, w = o.w;
    const int qlen0 = rd[0].len, qlen1 = rd[1].len, qlen_sum = qlen0 + qlen1;
    if (qlen0 > PMX_C_MAXLEN || qlen1 > PMX_C_MAXLEN || qlen0 <= 0 || qlen1 <= 0 || w < 1 || w > 12 || 2 * k + 11 > 64 || k > 255 || !o.is_sr_like)
        return PMX_C_BAIL;
    for (int s = 0; s < 2; ++s)   // an ambiguous base anywhere: general tier
        for (int c = 0; c < (rd[s].len + 31) >> 5; ++c)
            if (amb[s][c]) return PMX_C_BAIL;

    // ---------------------------------------------------------------- minimizers -> index probes -> seeds (P, Q)
    int n_s = 0;
    {
        bool have_prev = false;
        uint64_t prev_key = 0;
        int pending = -1;                 // seed of the previous read's last minimizer (its right neighbour is not known yet)
        for (int s = 0; s < 2; ++s) {
            const CRead& r = rd[s];
            int n_m = 0;
            bool ovf = false;
            uint64_t cw = 0;
            int ck = -1;
            auto base_at = [&](int i) {
                const int j = r.flip ? r.len - 1 - i : i;
                if ((j >> 5) != ck) { ck = j >> 5; cw = r.w[ck]; }
                const int c = (int)(cw >> (2 * (j & 31))) & 3;
                return r.flip ? 3 - c : c;
            };
            auto push = [&](uint64_t x, uint64_t y) {
                if (n_m < PMX_C_MCAP) m.setM(n_m++, (x >> 8) << 11 | ((uint32_t)y & 0x7ffu));
                else ovf = true;
            };
            if (w <= 8) sketch_core<8>(r.len, w, k, 0, base_at, push);
            else sketch_core<12>(r.len, w, k, 0, base_at, push);
            if (ovf) return PMX_C_BAIL;
            PMX_C_STAMP(0);
            const int sum = s ? qlen0 : 0;
            for (int i = 0; i < n_m; ++i) {
                const uint64_t mi = m.M(i);
                const uint64_t key = mi >> 11;
                const uint32_t ylow = (uint32_t)mi & 0x7ffu;
                bool tandem = have_prev && key == prev_key;
                if (i == 0 && tandem && pending >= 0) m.Q(pending) |= PMX_CQ_TANDEM;   // ... of the previous read's last one
                if (i + 1 < n_m && key == m.M(i + 1) >> 11) tandem = true;
                // mm_idx_get (index.c:81-99)
                uint32_t slot = (uint32_t)mix64(key) & ri.ht_mask;
                HtEnt e = ri.ht[slot];
                while (e.key != key && e.key != UINT64_MAX) { slot = (slot + 1) & ri.ht_mask; e = ri.ht[slot]; }
                const uint32_t cnt = e.key == key ? e.cnt : 0u;
                if (i == 0) pending = -1;
                if (cnt > 1) return PMX_C_BAIL;       // a repeated minimizer: general tier
                if (cnt == 1) {
                    if (n_s >= PMX_C_CAP) return PMX_C_BAIL;
                    const uint64_t pv = ri.pos[e.off];
                    if (pv >> 32) return PMX_C_BAIL;
                    m.P(n_s) = (uint32_t)pv;
                    m.Q(n_s) = (c_u16)((ylow + ((uint32_t)sum << 1)) | (s ? PMX_CQ_SEG : 0u) | (tandem ? PMX_CQ_TANDEM : 0u));
                    if (i == n_m - 1) pending = n_s;
                    ++n_s;
                } else if (i == n_m - 1) pending = -1;
                prev_key = key;
                have_prev = true;
            }
            PMX_C_STAMP(1);
        }
    }
    if (n_s == 0) return PMX_C_DONE;   // no anchors: unmapped

    // ---------------------------------------------------------------- heap merge -> anchors (AX, AY), map.c:102-166
    const int n = n_s;
    {
        for (int i = 0; i < n; ++i) m.HI(i) = (c_u8)i;
        auto heapdown = [&](int i, int sz) {   // ks_heapdown with "less" = larger reference position word (min-heap)
            const uint32_t tmp = m.HI(i);
            const uint32_t tk = m.P((int)tmp);
            int kk;
            while ((kk = (i << 1) + 1) < sz) {
                uint32_t ce = m.HI(kk);
                uint32_t ckey = m.P((int)ce);
                if (kk != sz - 1) {
                    const uint32_t c1 = m.HI(kk + 1);
                    const uint32_t k1 = m.P((int)c1);
                    if (ckey > k1) { ++kk; ce = c1; ckey = k1; }
                }
                if (ckey > tk) break;
                m.HI(i) = (c_u8)ce;
                i = kk;
            }
            m.HI(i) = (c_u8)tmp;
        };
        for (int q = (n >> 1) - 1; q >= 0; --q) heapdown(q, n);
        int sz = n, n_for = 0, n_rev = 0;
        while (sz > 0) {
            const int si = (int)m.HI(0);
            const uint32_t r = m.P(si);
            const uint32_t qy = m.Q(si);
            const uint32_t rpos = r >> 1, qp = qy & 0x3ffu;
            const uint32_t fl = qy & (PMX_CQ_SEG | PMX_CQ_TANDEM);
            if ((r & 1u) == (qp & 1u)) {
                m.AX(n_for) = rpos;
                m.AY(n_for) = (c_u16)((qp >> 1) | fl);
                ++n_for;
            } else {
                ++n_rev;
                m.AX(n - n_rev) = 0x80000000u | rpos;
                m.AY(n - n_rev) = (c_u16)((uint32_t)(qlen_sum - ((int)(qp >> 1) + 1 - k) - 1) | fl);
            }
            const uint32_t last = m.HI(sz - 1);
            --sz;
            if (sz > 0) { m.HI(0) = (c_u8)last; heapdown(0, sz); }
        }
        for (int j = 0; j < n_rev >> 1; ++j) {   // the reverse-strand block was filled back to front
            const int p = n - 1 - j, q2 = n - n_rev + j;
            const uint32_t tx = m.AX(p); const uint32_t ty = m.AY(p);
            m.AX(p) = m.AX(q2); m.AY(p) = m.AY(q2);
            m.AX(q2) = tx; m.AY(q2) = (c_u16)ty;
        }
    }

    PMX_C_STAMP(2);
    // ---------------------------------------------------------------- chain fill (lchain.c:148-230) -> F
    int max_chain_gap_ref;
    if (o.max_gap_ref > 0) max_chain_gap_ref = o.max_gap_ref;
    else if (o.max_frag_len > 0) {
        max_chain_gap_ref = o.max_frag_len - qlen_sum;
        if (max_chain_gap_ref < o.max_gap) max_chain_gap_ref = o.max_gap;
    } else max_chain_gap_ref = o.max_gap;
    const int bw = o.bw, max_skip = o.max_chain_skip, min_cnt = o.min_cnt, min_sc = o.min_chain_score;
    const int32_t max_drop = bw;
    {
        int32_t max_dist_x = max_chain_gap_ref, max_dist_y = o.max_gap;
        if (max_dist_x < bw) max_dist_x = bw;
        if (max_dist_y < bw) max_dist_y = bw;
        const float gp = o.chn_pen_gap, sp = o.chn_pen_skip;
        int st = 0, max_ii = -1;
        uint32_t ax_st = m.AX(0);
        uint64_t x_mi = 0;
        int32_t f_mi = 0;
        for (int i = 0; i < n; ++i) {
            const uint32_t axi = m.AX(i);
            const uint32_t ayi = m.AY(i);
            const uint64_t xi = c_x64(axi);
            const uint32_t rpi = axi & 0x7fffffffu;
            const int32_t qi = (int32_t)(ayi & 0x3ffu), sidi = (int32_t)(ayi >> 10 & 1u);
            int32_t max_f = k, n_skip = 0, mj = -1;
            while (st < i && (((axi ^ ax_st) >> 31) != 0u || xi > c_x64(ax_st) + (uint64_t)max_dist_x)) {
                ++st;
                ax_st = st < i ? (uint32_t)m.AX(st) : axi;
            }
            uint64_t mark = 0;   // bit (j - st): anchor j is the predecessor of an anchor already visited for this i
            int32_t ej = st - 1;
            bool stop = false;
            for (int32_t j = i - 1; j >= st && !stop; --j) {
                const uint32_t axj = m.AX(j);
                const uint32_t ayj = m.AY(j);
                const uint32_t fj = m.F(j);
                const int32_t sc0 = chain_score_sel(rpi, qi, sidi, axj & 0x7fffffffu, (int32_t)(ayj & 0x3ffu), (int32_t)(ayj >> 10 & 1u), k, max_dist_x,
                                                    max_dist_y, bw, gp, sp, 2);
                const bool valid = sc0 != INT32_MIN;
                const int32_t sc = sc0 + (int32_t)(fj & 0x3ffu);
                const bool better = valid && sc > max_f;
                const bool marked = valid && !better && (mark >> ((j - st) & 63) & 1ULL) != 0;
                max_f = better ? sc : max_f;
                mj = better ? j : mj;
                n_skip += (better && n_skip > 0) ? -1 : 0;
                n_skip += marked ? 1 : 0;
                const bool brk = marked && n_skip > max_skip;   // the reference breaks before marking p[j]
                ej = brk ? j : ej;
                stop = brk;
                const int32_t pj = (int32_t)(fj >> 10) - 1 - st;   // p[j] relative to st
                mark |= (valid && !brk && (fj >> 10) != 0u && pj >= 0) ? 1ULL << (pj & 63) : 0ULL;
            }
            int32_t max_j = mj;
            const int32_t end_j = ej;
            if (max_ii < 0 || (int64_t)(xi - x_mi) > (int64_t)max_dist_x) {
                int32_t mx = INT32_MIN;
                max_ii = -1;
                for (int32_t j = i - 1; j >= st; --j) {
                    const int32_t fj = (int32_t)(m.F(j) & 0x3ffu);
                    if (mx < fj) { mx = fj; max_ii = j; }
                }
                if (max_ii >= 0) { x_mi = c_x64(m.AX(max_ii)); f_mi = mx; }
            }
            if (max_ii >= 0 && max_ii < end_j) {
                const uint32_t axm = m.AX(max_ii);
                const uint32_t aym = m.AY(max_ii);
                const int32_t tmp = chain_score_sel(rpi, qi, sidi, axm & 0x7fffffffu, (int32_t)(aym & 0x3ffu), (int32_t)(aym >> 10 & 1u), k, max_dist_x,
                                                    max_dist_y, bw, gp, sp, 2);
                const int32_t fm = (int32_t)(m.F(max_ii) & 0x3ffu);
                if (tmp != INT32_MIN && max_f < tmp + fm) { max_f = tmp + fm; max_j = max_ii; }
            }
            if (max_f < 0 || max_f > 1023) return PMX_C_BAIL;
            m.F(i) = (c_u16)((uint32_t)max_f | (uint32_t)(max_j + 1) << 10);
            if (max_ii < 0 || ((int64_t)(xi - x_mi) <= (int64_t)max_dist_x && f_mi < max_f)) { max_ii = i; x_mi = xi; f_mi = max_f; }
        }
    }

    PMX_C_STAMP(3);
    // ---------------------------------------------------------------- backtrack (lchain.c:27-76) -> V, up to two chains
    int n_u = 0;
    int32_t u_sc[2] = {0, 0}, u_cnt[2] = {0, 0};
    {
        int n_z = 0;
        for (int i = 0; i < n; ++i) {
            const uint32_t f = m.F(i) & 0x3ffu;
            if ((int32_t)f >= min_sc) m.Z(n_z++) = (c_u16)(f << 6 | (uint32_t)i);
        }
        if (n_z == 0) return PMX_C_DONE;   // no chain: unmapped
        for (int i = 1; i < n_z; ++i) {    // (score, index) ascending = the stable insertion sort of radix_sort_128x for n <= 64
            const uint32_t t = m.Z(i);
            if (t < m.Z(i - 1)) {
                int j = i;
                for (; j > 0 && t < m.Z(j - 1); --j) m.Z(j) = m.Z(j - 1);
                m.Z(j) = (c_u16)t;
            }
        }
        uint64_t used = 0;
        int n_v = 0;
        for (int kz = n_z - 1; kz >= 0; --kz) {
            const uint32_t zk = m.Z(kz);
            const int i0 = (int)(zk & 63u);
            if (used >> i0 & 1ULL) continue;
            const int32_t zx = (int32_t)(zk >> 6);
            const int n_v0 = n_v;
            int i = i0;
            int32_t max_s = 0;
            uint64_t walk = 0, keep = 0;
            do {
                walk |= 1ULL << i;
                m.V(n_v0 + __builtin_popcountll(walk) - 1) = (c_u8)i;
                i = (int)(m.F(i) >> 10) - 1;
                const int32_t s = i < 0 ? zx : zx - (int32_t)(m.F(i) & 0x3ffu);
                if (s > max_s) { max_s = s; keep = walk; }
                else if (max_s - s > max_drop) break;
            } while (i >= 0 && !(used >> i & 1ULL));
            const int cnt = __builtin_popcountll(keep);
            used |= keep;
            n_v = n_v0 + cnt;
            if (max_s >= min_sc && cnt > 0 && cnt >= min_cnt) {
                if (n_u >= 2) return PMX_C_BAIL;   // a third chain: general tier
                u_sc[n_u] = max_s; u_cnt[n_u] = cnt;
                ++n_u;
            } else n_v = n_v0;
        }
    }
    if (n_u == 0) return PMX_C_DONE;   // unmapped
    PMX_C_STAMP(4);

    // ---------------------------------------------------------------- chains -> one region per mate (hit.c:54-94, 345-400)
    // anchors of segment s in chain c; a mate followed here has exactly one chain (then regs0's parent / secondary logic
    // has nothing to decide: the chains of different mates do not overlap on the fragment), a mate without anchors makes
    // the pair unmapped whatever the other one does
    int cs[2][2] = {{0, 0}, {0, 0}};
    {
        int off = 0;
        for (int c = 0; c < n_u; ++c) {
            for (int j = 0; j < u_cnt[c]; ++j) ++cs[c][(m.AY((int)m.V(off + j)) >> 10) & 1u];
            off += u_cnt[c];
        }
    }
    int chain_of[2];
    for (int s = 0; s < 2; ++s) {
        const int nreg = (cs[0][s] > 0) + (cs[1][s] > 0);
        if (nreg == 0) return PMX_C_DONE;   // unmapped
        chain_of[s] = cs[0][s] > 0 ? 0 : 1;
    }
    for (int s = 0; s < 2; ++s)
        if ((cs[0][s] > 0) + (cs[1][s] > 0) > 1) return PMX_C_BAIL;
    CReg R[2];
    int base[2];
    {
        // per-mate lists (ascending = the chain walked backwards), y rebased to the mate (hit.c:381)
        int wr = 0;
        for (int s = 0; s < 2; ++s) {
            const int c = chain_of[s];
            const int off = c ? u_cnt[0] : 0;
            base[s] = wr;
            int rev = 0;
            for (int j = u_cnt[c] - 1; j >= 0; --j) {
                const int ai = (int)m.V(off + j);
                const uint32_t ay = m.AY(ai);
                if ((int)((ay >> 10) & 1u) != s) continue;
                const uint32_t ax = m.AX(ai);
                rev = (int)(ax >> 31);
                const int ql = s ? qlen1 : qlen0, acc = s ? qlen0 : 0;
                const int shift = rev ? qlen_sum - (ql + acc) : acc;
                m.SX(wr) = ax;
                m.SY(wr) = (c_u16)(((ay & 0x3ffu) - (uint32_t)shift) | (ay & ~0x7ffu));   // position rebased, flags kept, segment dropped
                ++wr;
            }
            CReg& r = R[s];
            r.cnt = cs[c][s];
            r.score = u_sc[c];
            r.rev = rev;
            r.has_p = 0; r.dp_score = r.dp_max = 0; r.mapq = 0; r.proper_frag = 0; r.m_len = 0;
            c_reg_set_coor(m, base[s], r, s ? qlen1 : qlen0, k);
        }
    }

    PMX_C_STAMP(5);
    // ---------------------------------------------------------------- align each mate, filter, mapq (hit.c:301-322, 421-466)
    for (int s = 0; s < 2; ++s) {
        CReg& r = R[s];
        const int qlen = s ? qlen1 : qlen0;
        if (c_align1(m, base[s], o, ri, rd[s], qlen, r) != PMX_C_DONE) return PMX_C_BAIL;
        // mm_filter_regs (the region is a segment split: the min_cnt test does not apply)
        bool flt = false;
        if (r.mlen < o.min_chain_score) flt = true;
        else if (r.dp_max < o.min_dp_max) flt = true;
        else if (r.qs > qlen * o.max_clip_ratio && qlen - r.qe > qlen * o.max_clip_ratio) flt = true;
        if (flt) return PMX_C_DONE;   // the mate loses its only region: unmapped pair
        if (qlen >= o.rank_min_len) return PMX_C_BAIL;
        // mm_set_mapq for a lone primary without secondaries (subsc = n_sub = dp_max2 = 0, rep_len = 0)
        if (r.dp_max < 0 || r.dp_max >= ri.n_logf || r.score < 0 || r.score >= ri.n_logf) return PMX_C_BAIL;
        const float uniq_ratio = (float)(int64_t)r.score / (float)((int64_t)r.score + 0);
        const float pen_s1 = (r.score > 100 ? 1.0f : 0.01f * r.score) * uniq_ratio;
        float pen_cm = r.cnt > 10 ? 1.0f : 0.1f * r.cnt;
        pen_cm = pen_s1 < pen_cm ? pen_s1 : pen_cm;
        const int subsc = o.min_chain_score;   // max(r.subsc = 0, min_chain_sc)
        const float x = (float)subsc / r.score;   // score0 == score
        const float identity = (float)r.mlen / r.blen;
        int mapq = (int)(identity * pen_cm * 40.0f * (1.0f - x) * ri.logf_ratio[r.dp_max]);
        mapq -= (int)(4.343f * ri.logf_int[1] + .499f);
        mapq = mapq > 0 ? mapq : 0;
        r.mapq = mapq < 60 ? mapq : 60;
        if (r.dp_max > 0 && r.mapq == 0) r.mapq = 1;
    }

    PMX_C_STAMP(6);
    // ---------------------------------------------------------------- pairing (pe.c:76-177) with one region per mate
    if (o.pe_ori >= 0) {
        const int sub_diff = o.a * 2 + o.b;
        (void)sub_diff;
        uint64_t key[2];
        int ps[2] = {0, 1};
        for (int s = 0; s < 2; ++s) key[s] = (uint64_t)(uint32_t)(R[s].rs << 1) | (uint32_t)(s ^ R[s].rev);
        int dp_thres = R[0].dp_max + R[1].dp_max - o.pe_bonus;
        if (dp_thres < 0) dp_thres = 0;
        if (key[1] < key[0]) { const uint64_t t = key[0]; key[0] = key[1]; key[1] = t; ps[0] = 1; ps[1] = 0; }
        int64_t mx = -1;
        int n_sc = 0;
        int last[2] = {-1, -1};
        for (int i = 0; i < 2; ++i) {
            const CReg& ri_ = R[ps[i]];
            const int rev_i = ri_.rev;
            if (key[i] & 1) {   // reverse first read or forward second read
                if (last[rev_i] < 0) continue;
                const CReg* q = &R[ps[last[rev_i]]];
                if (ri_.rs - q->re > max_chain_gap_ref) continue;
                for (int j = last[rev_i]; j >= 0; --j) {
                    q = &R[ps[j]];
                    if (q->rev != rev_i || ps[j] == ps[i]) continue;
                    if (ri_.rs - q->re > max_chain_gap_ref) break;
                    if (ri_.dp_max + q->dp_max < dp_thres) continue;
                    const int64_t score = (int64_t)(ri_.dp_max + q->dp_max) << 32;   // (+ hash sum: only ranks several candidates)
                    if (score > mx) mx = score;
                    ++n_sc;
                }
            } else last[rev_i] = i;
        }
        if (n_sc > 0 && mx > 0) {   // one candidate pair: n_sc == 1
            R[0].proper_frag = R[1].proper_frag = 1;
            const int mapq_pe = R[0].mapq > R[1].mapq ? R[0].mapq : R[1].mapq;
            if (R[0].mapq < mapq_pe) R[0].mapq = (int)(.2f * R[0].mapq + .8f * mapq_pe + .499f);
            if (R[1].mapq < mapq_pe) R[1].mapq = (int)(.2f * R[1].mapq + .8f * mapq_pe + .499f);
            if (R[0].mapq < 2) R[0].mapq = 2;
            if (R[1].mapq < 2) R[1].mapq = 2;
        }
    }

    PMX_C_STAMP(7);
    // ---------------------------------------------------------------- the record (src/mm_align.c:271-354)
    if (!(R[0].score > 0 && R[1].score > 0)) return PMX_C_DONE;
    out.mapped = 1;
    for (int s = 0; s < 2; ++s) {
        CMate& t = out.m[s];
        t.rs = R[s].rs; t.re = R[s].re; t.qs = R[s].qs; t.qe = R[s].qe;
        t.dp_max = R[s].dp_max;
        t.cigar = (uint32_t)R[s].m_len << 4;
        t.mapq = (uint8_t)R[s].mapq; t.rev = (uint8_t)R[s].rev; t.proper_frag = (uint8_t)R[s].proper_frag; t.has_aln = 1;
    }
    return PMX_C_DONE;
}

}  // namespace aln
}  // namespace pmx
