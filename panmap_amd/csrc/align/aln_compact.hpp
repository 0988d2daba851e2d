// ALIGN stage, COMPACT tier: the short-read pair pipeline with its whole per-pair work state in LDS and registers.
//
// The general pipeline (aln_map.hpp) keeps minimap2's data shapes -- 16-byte anchors, 16-byte chain cells, 108-byte
// region records, byte-per-base sequences -- about 10 KB of work state per pair, which in the thread-per-pair kernel
// lives in an HBM arena and is what that kernel moves (23-35 GB per launch of 500k pairs against 93 MB of input and
// output, profiles/r01).  This tier restates the SAME decisions for the pairs that make up a short-read batch on a small
// genome, with the state packed to 7 bytes per anchor (9 when the reference is longer than 32,767 bases):
//   * reads stay 2 bit/base where the host left them (no decode pass); mismatches against the reference come from
//     XOR + popcount on 32-base words, once per mate and diagonal;
//   * a minimizer is 8 bytes while it waits for its index probe; a seed is a reference position word (2 or 4 bytes) and
//     a query word (2 bytes) and BECOMES the anchor in place (the heap merge of map.c:102-166 runs on a heap of one-byte
//     seed indices, its pop order is turned into destinations and the seeds are permuted along the cycles); a chain
//     cell is 2 bytes (score | predecessor); the members of a chain are a 64-bit set in a register (a chain walks to ever
//     smaller indices), per-mate anchor lists are two-byte indices in the dead chain cells;
//   * 56 anchors per pair (99.9 % of 150 bp pairs; mean 40): 336 bytes per pair (448 with 32-bit positions),
//     interleaved word-wise across the 64 lanes of a wave in LDS (conflict-free): 21 KB per wave, seven waves per CU;
//     the first form of the two-kernel chain kernel keeps 48 (99.1 % of the pairs; the others are the second form's):
//     288 bytes per pair, eight waves per CU (CMemT's CAP parameter, PMX_C_CAP1);
//   * regions are a handful of scalars in registers: at most one region per mate is followed.
// Everything outside that envelope -- an ambiguous base, a seed that occurs twice in the reference, more than 56 seeds,
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

// One lane's work memory; word i of lane l sits at base + i * 64 + l (device), so a wave touching word i of its 64
// pairs touches 256 contiguous bytes of LDS.  PT = uint16_t when every reference position word fits 16 bits
// (position << 1 | strand: references up to 32,767 bases), uint32_t otherwise.
//   X  PT  [56]  seed: reference position word (position << 1 | strand)  ->  anchor x: strand << (bits-1) | position
//   Y  u16 [56]  seed / anchor query word (PMX_CQ_*); bits 12..15: during the chain fill, the length of the colinear run that ends here
//   G  u16 [56]  merge: heap of seed indices / its pop order (high bytes) + destination of every seed (low bytes) | chain
//                cells F | after the backtrack: per-mate anchor index lists
//   M  u64 [..]  minimizers of one read waiting for their probes: overlays everything from word 0
// Chain gap penalties by diagonal difference (lchain.c:113-141 with chn_pen_skip == 0): the two float expressions of
// comput_sc depend on dd alone, so a wave tabulates them once per kernel (exactly as chain_score_sel evaluates them) and the
// fill reads a byte instead of doing ~25 VALU operations per anchor pair.  same[dd] = (int)(gp*dd + .5f*log2(dd+1)),
// diff[dd] = (int)min(gp*dd, log2(dd+1)); dd == 0 -> 0.
struct CPenTab {
    const c_u8* same;   // NULL: no tables (evaluate arithmetically)
    const c_u8* diff;
};
PMX_HD void c_pen_values(float gp, int dd, int* same, int* diff) {
    const float lin_pen = gp * (float)dd;
    const float log_pen = dd >= 1 ? mg_log2f((float)((uint32_t)dd + 1u)) : 0.0f;
    const float ps = lin_pen + .5f * log_pen, pd = lin_pen < log_pen ? lin_pen : log_pen;
    *same = (int)ps;
    *diff = (int)pd;
}

// comput_sc (lchain.c:113-141) for two segments with the float penalties read from the tables; same integer
// expressions as chain_score_sel, values of rejected pairs are discarded
PMX_HD int32_t c_chain_score_tab(const CPenTab& T, uint32_t xi, int32_t yi, int32_t sidi, uint32_t xj, int32_t yj, int32_t sidj, int32_t q_span,
                                 int32_t max_dist_x, int32_t max_dist_y, int32_t bw) {
    const int32_t dq = yi - yj;
    const int32_t dr = (int32_t)(xi - xj);
    const bool same = sidi == sidj;
    const int32_t dd = dr > dq ? dr - dq : dq - dr;
    bool bad = dq <= 0 || dq > max_dist_x;
    bad = bad || (same && (dr == 0 || dq > max_dist_y || dd > bw || dr > max_dist_y));
    const int32_t dg = dr < dq ? dr : dq;
    const int32_t sc = q_span < dg ? q_span : dg;
    const int32_t lim = same ? PMX_C_PEN_SAME - 1 : PMX_C_PEN_DIFF - 1;
    const int32_t ddc = dd < lim ? dd : lim;                 // (only rejected pairs can exceed the tables)
    const int32_t pen = same ? (int32_t)T.same[ddc] : (int32_t)T.diff[ddc];
    const int32_t r = (!same && dr == 0) ? sc + 1 : sc - pen;   // overlapping paired ends (lchain.c:135)
    return bad ? INT32_MIN : r;
}

template <class PT, int CAP = PMX_C_CAP>
struct CMemT {
    c_u32* base;
    static constexpr int kCap = CAP;
    static constexpr int kXH = 0;                                        // X in units of PT
    static constexpr int kYH = (int)(sizeof(PT) / 2) * kCap;             // Y, G in halves
    static constexpr int kGH = kYH + kCap;
    static constexpr int kWords = (kGH + kCap + 1) / 2;
    static_assert(kCap <= 63, "predecessor + 1 in six bits, marks and the used set in 64");
    static constexpr int kMCap = kWords / 2 < PMX_C_MCAP ? kWords / 2 : PMX_C_MCAP;
    static constexpr uint32_t kRevBit = 1u << (8 * sizeof(PT) - 1);
    PMX_HD c_u32& w(int i) const { return base[i * PMX_C_STRIDE]; }
    PMX_HD c_u16& h(int i) const { return ((c_u16*)(base + (i >> 1) * PMX_C_STRIDE))[i & 1]; }
    PMX_HD c_u8& b(int i) const { return ((c_u8*)(base + (i >> 2) * PMX_C_STRIDE))[i & 3]; }
    PMX_HD uint32_t X(int i) const { return sizeof(PT) == 2 ? (uint32_t)h(i) : (uint32_t)w(i); }
    PMX_HD void setX(int i, uint32_t v) const { if (sizeof(PT) == 2) h(i) = (c_u16)v; else w(i) = v; }
    PMX_HD c_u16& Y(int i) const { return h(kYH + i); }
    PMX_HD c_u16& G(int i) const { return h(kGH + i); }
    PMX_HD c_u8& GL(int i) const { return b(2 * (kGH + i)); }       // the two bytes of G(i), for the phases that keep two
    PMX_HD c_u8& GH(int i) const { return b(2 * (kGH + i) + 1); }   // one-byte lists in it
    PMX_HD uint64_t M(int i) const { return (uint64_t)w(2 * i) | (uint64_t)w(2 * i + 1) << 32; }
    PMX_HD void setM(int i, uint64_t v) const { w(2 * i) = (uint32_t)v; w(2 * i + 1) = (uint32_t)(v >> 32); }
    // what a CSeeder asks of the memory it works on: a queue of waiting minimizers (here: the M entries that lie above X
    // and Y, 14 of them) and a place for the seeds (here: X and Y themselves)
    typedef PT PosT;
    static constexpr int kQBase = (kGH / 2 + 1) / 2;
    static constexpr int kQCap = kWords / 2 - kQBase;
    PMX_HD uint64_t QM(int i) const { return M(kQBase + i); }
    PMX_HD void setQM(int i, uint64_t v) const { setM(kQBase + i, v); }
    PMX_HD void setSeed(int i, uint32_t x, uint32_t y) const { setX(i, x); h(kYH + i) = (c_u16)y; }
    PMX_HD void orY(int i, uint32_t f, int /*seeds so far*/ = 0) const { h(kYH + i) |= (c_u16)f; }
    PMX_HD void flush_tail(int /*seeds*/) const {}
    // anchor x split: strand, reference position
    PMX_HD static uint32_t rev_of(uint32_t x) { return x >> (8 * sizeof(PT) - 1); }
    PMX_HD static uint32_t pos_of(uint32_t x) { return x & (kRevBit - 1u); }
    // 64-bit anchor x as the reference holds it (strand in bit 63, one reference sequence: rid = 0)
    PMX_HD static uint64_t x64(uint32_t x) { return (uint64_t)rev_of(x) << 63 | (uint64_t)pos_of(x); }
};

// query word of a seed / anchor: bits 0..9 position (seed: position << 1 | strand), 10 segment, 11 tandem, 12 ignore
#define PMX_CQ_SEG 0x400u
#define PMX_CQ_TANDEM 0x800u
#define PMX_CQ_IGNORE 0x1000u
#define PMX_CQ_LONG_JOIN 0x2000u

// The seed hand-over of the two-kernel form (k_compact_seeds -> k_align_compact): the sketch and the index probes need
// none of the pair's LDS state, only the short minimizer queue, so they run in a kernel of their own at twice the waves
// per CU and leave the seeds -- the very X and Y words the fused form writes to LDS -- in HBM, launch position `it` owning
// the kPairWords words from it * kPairWords on (PT = u16: one word per seed, X | Y << 16; PT = u32: two, X then Y).
// (Until round 4 the words of the 64 pairs of a wave were interleaved, seed i of lane l at ((b * kCap + i) * kWS + j) * 64 + l:
//  one contiguous store per seed index IF the lanes had reached the same index -- they have not, every lane is at its own
//  count when the wave drains its queues, so the stores were single words scattered over as many lines, each a partial-line
//  write the memory side completes by reading the line first: 5.6 GB of HBM traffic per 5M pairs for 0.8 GB of seeds.
//  A pair's own stretch fills its lines one word after the other.)
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(1))) uint32_t c_g32;
#else
typedef uint32_t c_g32;
#endif
template <class PT>
struct CSeedOutT {
    typedef PT PosT;
    static constexpr int kCap = PMX_C_CAP;
    static constexpr int kWS = sizeof(PT) == 2 ? 1 : 2;
    static constexpr int kQCap = PMX_C_SEEDQ;
    static constexpr int kPairWords = kCap * kWS;            // per launch position (a multiple of four: 16-byte loads)
    static constexpr int kBlockWords = kPairWords * 64;      // per 64 launch positions
    static_assert(kPairWords % 4 == 0, "a pair's hand-over words are read as 16-byte pieces");
    c_u32* q;         // the lane's queue words (LDS, word i at q[i * stride])
    c_g32* out;       // the pair's hand-over words
    // Seeds wait in eight staging words of the lane (LDS, word j at st[j * stride]) until a 32-byte piece of the pair's
    // stretch is complete and leave as that piece: whole sectors, written once.  (Stored word by word they were partial
    // writes: the lines of the 1,024 pairs a CU works on at a time are as large as its share of the L2, a line was gone
    // before the pair's next seed arrived, and the memory side read it back to merge four bytes -- 7.6 GB per 5M pairs.)
    c_u32* st;
    static constexpr int kSPC = 8 / kWS;   // seeds per piece
    PMX_HD uint64_t QM(int i) const { return (uint64_t)q[2 * i * PMX_C_STRIDE] | (uint64_t)q[(2 * i + 1) * PMX_C_STRIDE] << 32; }
    PMX_HD void setQM(int i, uint64_t v) const { q[2 * i * PMX_C_STRIDE] = (uint32_t)v; q[(2 * i + 1) * PMX_C_STRIDE] = (uint32_t)(v >> 32); }
    PMX_HD void flush(int piece) const {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef __attribute__((address_space(1))) uint4 g_u4;
        g_u4* dst = reinterpret_cast<g_u4*>(out + 8 * piece);
        dst[0] = make_uint4(st[0], st[1 * PMX_C_STRIDE], st[2 * PMX_C_STRIDE], st[3 * PMX_C_STRIDE]);
        dst[1] = make_uint4(st[4 * PMX_C_STRIDE], st[5 * PMX_C_STRIDE], st[6 * PMX_C_STRIDE], st[7 * PMX_C_STRIDE]);
#else
        for (int j = 0; j < 8; ++j) out[8 * piece + j] = st[j * PMX_C_STRIDE];
#endif
    }
    PMX_HD void setSeed(int i, uint32_t x, uint32_t y) const {
        const int j = i % kSPC;
        if (kWS == 1) st[j * PMX_C_STRIDE] = x | y << 16;
        else { st[2 * j * PMX_C_STRIDE] = x; st[(2 * j + 1) * PMX_C_STRIDE] = y; }
        if (j == kSPC - 1) flush(i / kSPC);
    }
    PMX_HD void flush_tail(int n_seeds) const {   // the pair is through: its last, incomplete piece
        if (n_seeds % kSPC) flush(n_seeds / kSPC);
    }
    // (seed i is one of the lane's own earlier ones; n_now seeds exist: its piece has left iff it is complete)
    PMX_HD void orY(int i, uint32_t f, int n_now) const {
        const int w = kWS == 1 ? i : 2 * i + 1;
        const uint32_t v = kWS == 1 ? f << 16 : f;
        if (n_now / kSPC > i / kSPC) out[w] |= v;
        else st[(w % 8) * PMX_C_STRIDE] |= v;
    }
    PMX_HD void get(int i, uint32_t* x, uint32_t* y) const {
        if (kWS == 1) { const uint32_t v = out[i]; *x = v & 0xffffu; *y = v >> 16; }
        else { *x = out[2 * i]; *y = out[2 * i + 1]; }
    }
    // seeds i0 .. i0 + 3 (i0 a multiple of four; entries past the pair's count are whatever the stretch holds)
    PMX_HD void get4(int i0, uint32_t* x, uint32_t* y) const {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef __attribute__((address_space(1))) const uint4 g_u4;
        if (kWS == 1) {
            const uint4 v = *reinterpret_cast<g_u4*>(out + i0);
            x[0] = v.x & 0xffffu; y[0] = v.x >> 16; x[1] = v.y & 0xffffu; y[1] = v.y >> 16;
            x[2] = v.z & 0xffffu; y[2] = v.z >> 16; x[3] = v.w & 0xffffu; y[3] = v.w >> 16;
        } else {
            const uint4 a = *reinterpret_cast<g_u4*>(out + 2 * i0), b = *reinterpret_cast<g_u4*>(out + 2 * i0 + 4);
            x[0] = a.x; y[0] = a.y; x[1] = a.z; y[1] = a.w; x[2] = b.x; y[2] = b.y; x[3] = b.z; y[3] = b.w;
        }
#else
        for (int b = 0; b < 4; ++b) get(i0 + b, &x[b], &y[b]);
#endif
    }
};
#define PMX_C_NSEED_BAIL 0xffffu   // the hand-over count (seeds | seeds of mate 1 << 8) of a pair the seeding already gave up on

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
    int32_t edit[2];      // want_edits: count_read_errors of each mate (valid whenever PMX_C_DONE is returned)
};
#define PMX_C_DONE 0
#define PMX_C_BAIL 1

// one region of one mate while it is aligned (the fields of mm_reg1_t this tier can reach)
struct CReg {
    int32_t cnt, score, rev, qs, qe, rs, re, mlen, blen, dp_score, dp_max, has_p, mapq, proper_frag, m_len;
};

#define PMX_LAMBDA_INLINE __attribute__((always_inline))
// does any lane of the wave that is still with us see `p`? (host: the one "lane")
PMX_HD bool c_wave_any(bool p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ballot(p) != 0ULL;
#else
    return p;
#endif
}

// The anchors of one mate's only region, in chain order: anchor i of the list = anchor G(base + i) of the pair, its query
// position rebased to the mate (hit.c:381).  (x: reference position, y: query position; flags never survive into this
// tier's regions -- a LONG_JOIN / IGNORE mark needs the bad-seed filters, which make the pair bail.)
template <class MT>
struct CList {
    const MT& m;
    int base, shift;
    PMX_HD int32_t x(int i) const { return (int32_t)MT::pos_of(m.X((int)m.G(base + i))); }
    PMX_HD int32_t y(int i) const { return (int32_t)(m.Y((int)m.G(base + i)) & 0x3ffu) - shift; }
    PMX_HD uint32_t rev(int i) const { return MT::rev_of(m.X((int)m.G(base + i))); }
};

// mm_cal_fuzzy_len + mm_reg_set_coor (hit.c:8-40) on a per-mate anchor list
template <class MT>
PMX_HD void c_reg_set_coor(const CList<MT>& a, CReg& r, int32_t qlen, int span) {
    const int32_t x0 = a.x(0), y0 = a.y(0);
    const int32_t xl = a.x(r.cnt - 1), yl = a.y(r.cnt - 1);
    r.rs = x0 + 1 > span ? x0 + 1 - span : 0;
    r.re = xl + 1;
    if (!r.rev) { r.qs = y0 + 1 - span; r.qe = yl + 1; }
    else { r.qs = qlen - (yl + 1); r.qe = qlen - (y0 + 1 - span); }
    r.mlen = r.blen = span;
    int32_t px = x0, py = y0;
    for (int i = 1; i < r.cnt; ++i) {
        const int32_t x = a.x(i), y = a.y(i);
        const int tl = x - px, ql = y - py;
        r.blen += tl > ql ? tl : ql;
        r.mlen += tl > span && ql > span ? span : tl < ql ? tl : ql;
        px = x; py = y;
    }
}

// mm_fix_bad_ends (align.c:464-502) on a per-mate list (r.as == 0, no LONG_JOIN marks)
template <class MT>
PMX_HD void c_fix_bad_ends(const CList<MT>& a, const CReg& r, int span, int bw, int min_match, int32_t* as, int32_t* cnt) {
    *as = 0;
    *cnt = r.cnt;
    if (r.cnt < 3) return;
    int32_t mm_ = span, l = span;
    for (int32_t i = 1; i < r.cnt - 1; ++i) {
        const int32_t lr = a.x(i) - a.x(i - 1);
        const int32_t lq = a.y(i) - a.y(i - 1);
        const int32_t mn = lr < lq ? lr : lq, mx = lr > lq ? lr : lq;
        if (mx - mn > l >> 1) *as = i;
        l += mn;
        mm_ += mn < span ? mn : span;
        if (l >= bw << 1 || (mm_ >= min_match && mm_ >= bw) || mm_ >= r.mlen >> 1) break;
    }
    *cnt = r.cnt - *as;
    mm_ = l = span;
    for (int32_t i = r.cnt - 2; i > *as; --i) {
        const int32_t lr = a.x(i + 1) - a.x(i);
        const int32_t lq = a.y(i + 1) - a.y(i);
        const int32_t mn = lr < lq ? lr : lq, mx = lr > lq ? lr : lq;
        if (mx - mn > l >> 1) *cnt = i + 1 - *as;
        l += mn;
        mm_ += mn < span ? mn : span;
        if (l >= bw << 1 || (mm_ >= min_match && mm_ >= bw) || mm_ >= r.mlen >> 1) break;
    }
}

// base readers for the (1c) probe of ksw_shortcut_ext_decide: position i of the extension = query base at + step * i
// of the mate (orientation `orient`) / reference base at + step * i
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
// MULTI (aln_compact_multi.hpp): the region is one of several of its mate; `al` is still the region's own list, `mate`
// the mate's whole anchor list (n_a entries, the region's at [as, as + cnt)): neighbouring chains of the same strand
// fence the end extensions off (align.c:636-691; extension_reach, aln_align.hpp).
template <bool MULTI = false, class MT>
PMX_HD int c_align1(const CList<MT>& al, const Opt& o, const RefIndex& ri, const CRead& rd, int qlen, CReg& r, const CList<MT>* mate = nullptr, int as = 0,
                    int n_a = 0) {
    const int span = o.k;
    const int32_t rev = r.rev;
    const int32_t ref_len = ri.len;
    const int bw = (int)(o.bw * 1.5 + 1.);
    int bw_long = (int)(o.bw_long * 1.5 + 1.);
    if (bw_long < bw) bw_long = bw;
    int32_t as1, cnt1;
    c_fix_bad_ends(al, r, span, o.bw, o.min_chain_score * 2, &as1, &cnt1);
    {   // mm_filter_bad_seeds / _alt (align.c:391-462) act only when two or more anchor steps change the diagonal by more
        // than 10 (their min_gap; the second filter's 30 is implied): not followed here
        int n_long = 0;
        for (int i = 1; i < cnt1; ++i) {
            const int gap = (al.y(as1 + i) - al.y(as1 + i - 1)) - (al.x(as1 + i) - al.x(as1 + i - 1));
            if (gap < -10 || gap > 10) ++n_long;
        }
        if (n_long > 1) return PMX_C_BAIL;
    }
    int32_t rs = al.x(as1) - (o.k >> 1), qs = al.y(as1) - (o.k >> 1);                           // mm_adjust_minier, non-HPC
    int32_t re = al.x(as1 + cnt1 - 1) - (o.k >> 1), qe = al.y(as1 + cnt1 - 1) - (o.k >> 1);
    int32_t l, rs0, re0, qs0, qe0, rs1, qs1, re1, qe1;
    if (MULTI) {
        const int32_t own_t0 = al.x(0) + 1 - span, own_q0 = al.y(0) + 1 - span;
        const int32_t own_t = own_t0 < 0 ? 0 : own_t0;
        if (qs > 0 && rs > 0) {
            Reach fence{qs, rs};
            int seen = 0;
            for (int32_t i = as - 1; i >= 0 && (int32_t)mate->rev(i) == rev; --i) {
                const int32_t t_i = mate->x(i) + 1 - span, q_i = mate->y(i) + 1 - span;
                if (t_i >= own_t || q_i >= own_q0) continue;
                if (++seen > o.min_cnt) {
                    const int32_t back = own_t - t_i > own_q0 - q_i ? own_t - t_i : own_q0 - q_i;
                    fence.q = qs - (own_q0 - back);
                    fence.t = rs - (own_t - back);
                    break;
                }
            }
            const Reach rch = extension_reach(o, qs, rs, fence, Reach{qs - own_q0, rs - own_t});
            qs0 = qs - rch.q;
            rs0 = rs - (rch.t < 0 ? 0 : rch.t);
        } else { rs0 = rs; qs0 = qs; }
        const int32_t own_t1 = al.x(r.cnt - 1) + 1, own_q1 = al.y(r.cnt - 1) + 1;
        if (qe < qlen && re < ref_len) {
            Reach fence{qlen - qe, ref_len - re};
            int seen = 0;
            for (int32_t i = as + r.cnt; i < n_a && (int32_t)mate->rev(i) == rev; ++i) {
                const int32_t t_i = mate->x(i) + 1, q_i = mate->y(i) + 1;
                if (t_i <= own_t1 || q_i <= own_q1) continue;
                if (++seen > o.min_cnt) {
                    const int32_t ahead = t_i - own_t1 > q_i - own_q1 ? t_i - own_t1 : q_i - own_q1;
                    fence.q = own_q1 + ahead - qe;
                    fence.t = own_t1 + ahead - re;
                    break;
                }
            }
            const Reach rch = extension_reach(o, qlen - qe, ref_len - re, fence, Reach{own_q1 - qe, own_t1 - re});
            qe0 = qe + rch.q;
            re0 = re + rch.t;
        } else { re0 = re; qe0 = qe; }
        rs1 = qs1 = re1 = qe1 = 0; l = 0; (void)l;
    } else {
    // region to align (align.c:636-691); the region is the only one of its list: no neighbouring anchors to stop at
    rs0 = al.x(0) + 1 - span;
    qs0 = al.y(0) + 1 - span;
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
    re0 = al.x(r.cnt - 1) + 1;
    qe0 = al.y(r.cnt - 1) + 1;
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
    }
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

// mm_filter_regs + mm_set_mapq (hit.c:301-322, 421-466) for a mate's lone primary without secondaries (the region is a
// segment split, so the min_cnt test does not apply; subsc = n_sub = dp_max2 = 0, rep_len = 0).  PMX_C_DONE with
// *kept = false: the region is filtered, i.e. the pair is unmapped.
PMX_HD int c_filter_mapq(const Opt& o, const RefIndex& ri, int qlen, CReg& r, bool* kept) {
    *kept = false;
    bool flt = false;
    if (r.mlen < o.min_chain_score) flt = true;
    else if (r.dp_max < o.min_dp_max) flt = true;
    else if (r.qs > qlen * o.max_clip_ratio && qlen - r.qe > qlen * o.max_clip_ratio) flt = true;
    if (flt) return PMX_C_DONE;
    if (qlen >= o.rank_min_len) return PMX_C_BAIL;
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
    *kept = true;
    return PMX_C_DONE;
}

// Minimizers -> index probes -> seeds, one read at a time.  The object is BOTH functors of sketch_core: operator()(i)
// hands out base i (and is where the wave drains its queues: it runs once per base in every lane), operator()(x, y)
// queues a minimizer.  Minimizers wait for their probes in a short queue that overlays G (14 entries); the whole
// wave drains its queues together -- four probes in flight per lane -- whenever some lane holds six (one base can add
// up to w), so the drain is a uniform branch and the seeds never need a staging copy of the minimizer list.  The newest
// entry stays queued until its right neighbour is known (the tandem mark compares adjacent minimizers, seed.c:40-46).
template <class MS>
struct CSeeder {
    typedef typename MS::PosT PT;
    static constexpr int kQCap = MS::kQCap;
    static constexpr int kQDrain = kQCap - 2;                         // drain threshold (sketch_distinct pushes at most one minimizer per base)
    static_assert(kQCap >= 8, "minimizer queue too short");
    const MS& m;
    const RefIndex& ri;
    CRead r;
    int seg, sum;          // segment, summed length of the segments before it
    int n_q, n_s;          // queue fill, seeds so far (both reads)
    bool first_of_read, ovf, bail, have_prev;
    uint64_t prev_key;
    int pending;           // seed of the previous read's last minimizer (its right neighbour is not known yet)
    uint64_t cw;
    int ck;
    // the read's packed words, fetched together when the read starts (its record is one line: word by word, 32 bases apart in
    // time, the line was gone from the L2 in between -- the records of the pairs a CU works on fill its share of the L2 --
    // and came from HBM again: 3.8 line fetches per pair instead of one)
    uint64_t rw[PMX_C_NW];
    PMX_HD void load_words() {
#pragma unroll
        for (int c = 0; c < PMX_C_NW; ++c) rw[c] = c < (r.len + 31) >> 5 ? r.w[c] : 0ULL;
    }

    // entries [0, lim) of the queue -> seeds; final: the read is over, the last entry has no right neighbour here
    PMX_HD void drain(bool final) {
        const int lim = final ? n_q : n_q - 1;
        for (int e0 = 0; c_wave_any(e0 < lim && !bail); e0 += 4) {
            uint64_t key[5];
            uint32_t yl[4], slot[4], pv[4];
            HtEnt e[4];
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                const uint64_t mi = e0 + b < n_q ? m.QM(e0 + b) : 0ULL;
                key[b] = mi >> 11;
                if (b < 4) yl[b] = (uint32_t)mi & 0x7ffu;
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {   // mm_idx_get (index.c:81-99): first probes of four minimizers together
                slot[b] = (uint32_t)mix64(key[b]) & ri.ht_mask;
                e[b] = HtEnt{UINT64_MAX, 0u, 0u};
                pv[b] = 0;
                if (e0 + b < lim && !bail) { e[b] = ri.ht[slot[b]]; pv[b] = ri.ht_pv[slot[b]]; }
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const bool live = e0 + b < lim && !bail;
                if (live) {
                    while (e[b].key != key[b] && e[b].key != UINT64_MAX) {   // collision: keep probing
                        slot[b] = (slot[b] + 1) & ri.ht_mask;
                        e[b] = ri.ht[slot[b]];
                        pv[b] = ri.ht_pv[slot[b]];
                    }
                    const uint32_t cnt = e[b].key == key[b] ? e[b].cnt : 0u;
                    bool tandem = have_prev && key[b] == prev_key;
                    if (first_of_read && tandem && pending >= 0) m.orY(pending, PMX_CQ_TANDEM, n_s);   // ... of the previous read's last one
                    const bool has_next = e0 + b + 1 < n_q;
                    if (has_next && key[b] == key[b + 1]) tandem = true;
                    if (first_of_read) pending = -1;
                    first_of_read = false;
                    if (cnt > 1) bail = true;                      // a repeated minimizer: general tier
                    else if (cnt == 1) {
                        if (n_s >= MS::kCap || (pv[b] >> (8 * sizeof(PT) - 1) >> 1) != 0u) bail = true;
                        else {
                            m.setSeed(n_s, pv[b], (yl[b] + ((uint32_t)sum << 1)) | (seg ? PMX_CQ_SEG : 0u) | (tandem ? PMX_CQ_TANDEM : 0u));
                            pending = has_next ? -1 : n_s;   // (only the read's last minimizer has no right neighbour yet)
                            ++n_s;
                        }
                    } else if (!has_next) pending = -1;
                    prev_key = key[b];
                    have_prev = true;
                }
            }
        }
        if (!final && n_q > 0) {   // the newest entry moves to the front
            const uint64_t last = m.QM(n_q - 1);
            m.setQM(0, last);
            n_q = 1;
        } else if (final) n_q = 0;
    }
    PMX_HD int operator()(int i) {   // base i of the read in the orientation the aligner sees
        if (c_wave_any(n_q >= kQDrain)) drain(false);
        const int j = r.flip ? r.len - 1 - i : i;
        if ((j >> 5) != ck) {
            ck = j >> 5;
            cw = rw[0];
#pragma unroll
            for (int c = 1; c < PMX_C_NW; ++c) cw = ck == c ? rw[c] : cw;
        }
        const int c = (int)(cw >> (2 * (j & 31))) & 3;
        return r.flip ? 3 - c : c;
    }
    PMX_HD void operator()(uint64_t x, uint64_t y) {   // a minimizer
        if (n_q < kQCap) { m.setQM(n_q, (x >> 8) << 11 | ((uint32_t)y & 0x7ffu)); ++n_q; }
        else ovf = true;
    }
};

// (w,k)-minimizers of a read without ambiguous bases, k odd, when no window holds its minimum twice: then sketch.c:77-143
// emits exactly the distinct minima of the full windows (w consecutive k-mers) from left to right -- or, for a read with
// fewer than w k-mers, the minimum of all of them.  k odd: no k-mer equals its reverse complement, so the window
// advances at every base.  The minima come from block prefix / suffix minima (blocks of W k-mers; the window that ends
// at offset r of a block = the previous block's suffix from r+1 and this block's prefix up to r), three comparisons per
// k-mer with every slot index a compile-time constant, instead of the branch-free W-slot ring scan of sketch_core.
// Returns false when two k-mers that share a window tie for a minimum (the duplicate-emission rules of sketch.c:107-139
// would apply): the caller hands the pair to the general tier.  Pushes arrive in the same order as sketch_core's.
template <int W, class BaseFn, class PushFn>
PMX_HD bool sketch_distinct(int len, int k, BaseFn& base_at, PushFn& push) {
    const uint64_t shift1 = 2 * (uint64_t)(k - 1), mask = (1ULL << 2 * k) - 1;
    const int n_k = len - k + 1;
    if (n_k <= 0) return true;
    uint64_t kmer0 = 0, kmer1 = 0;
    for (int i = 0; i < k - 1; ++i) {
        const int c = base_at(i);
        kmer0 = (kmer0 << 2 | (uint64_t)c) & mask;
        kmer1 = (kmer1 >> 2) | (uint64_t)(3 ^ c) << shift1;
    }
    // slot r: until step r of the current block the previous block's suffix minimum from offset r, afterwards the
    // block's own k-mer r
    uint64_t sx[W];
    uint32_t sy[W];
#pragma unroll
    for (int r = 0; r < W; ++r) { sx[r] = UINT64_MAX; sy[r] = 0xffffffffu; }
    uint64_t px = UINT64_MAX;
    uint32_t py = 0xffffffffu, last_y = 0xffffffffu;
    bool tie = false;
    for (int e0 = 0; e0 < n_k; e0 += W) {
        px = UINT64_MAX; py = 0xffffffffu;
#pragma unroll
        for (int r = 0; r < W; ++r) {
            const int e = e0 + r;
            uint64_t x = UINT64_MAX;
            uint32_t y = 0xffffffffu;
            if (e < n_k) {
                const int i = e + k - 1;
                const int c = base_at(i);
                kmer0 = (kmer0 << 2 | (uint64_t)c) & mask;
                kmer1 = (kmer1 >> 2) | (uint64_t)(3 ^ c) << shift1;
                const uint32_t z = kmer0 < kmer1 ? 0u : 1u;
                x = mz_hash64(z ? kmer1 : kmer0, mask) << 8 | (uint64_t)k;
                y = (uint32_t)i << 1 | z;
                tie |= x == px;
                if (x < px) { px = x; py = y; }
                uint64_t wx = px;
                uint32_t wy = py;
                if (r + 1 < W) {
                    tie |= sx[r + 1 < W ? r + 1 : 0] == px;
                    if (sx[r + 1 < W ? r + 1 : 0] < px) { wx = sx[r + 1 < W ? r + 1 : 0]; wy = sy[r + 1 < W ? r + 1 : 0]; }
                }
                if (e >= W - 1 && wy != last_y) { push(wx, (uint64_t)wy); last_y = wy; }
            }
            sx[r] = x; sy[r] = y;
        }
#pragma unroll
        for (int r = W - 2; r >= 1; --r) {
            tie |= sx[r] == sx[r + 1] && sx[r] != UINT64_MAX;
            if (sx[r + 1] < sx[r]) { sx[r] = sx[r + 1]; sy[r] = sy[r + 1]; }
        }
    }
    if (n_k < W) push(px, (uint64_t)py);
    return !tie;
}

#ifndef PMX_C_COUNT
#define PMX_C_COUNT(k, v) ((void)0)   // tests/hostsim counts work here
#endif
#ifndef PMX_C_TRACE
#define PMX_C_TRACE(i, trips, rescan) ((void)0)   // tests/hostsim: per-anchor trip counts of the chain fill's inner loops
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define PMX_C_STAMP(k) do { if (prof_on) { const unsigned long long t_ = (unsigned long long)clock64(); prof[k] += t_ - prof_t; prof_t = t_; } } while (0)
#else
#define PMX_C_STAMP(k) ((void)0)
#endif
// The pair, in two parts.  rd / amb = the two mates as packed by the host.
// (1) compact_seed_pair: envelope checks, minimizers -> index probes -> seeds, into whatever `ms` is (CMemT: the pair's
//     LDS block; CSeedOutT: the HBM hand-over).  PMX_C_DONE with the seed count in *n_seeds, or PMX_C_BAIL.
// (2) compact_chain_pair: from the seeds in X / Y of the pair's LDS block to the records; `out` is only meaningful when
//     PMX_C_DONE is returned.
// compact_map_pair runs both on one memory block (the fused form: hostsim, PMX_ALIGN_COMPACT_FUSED).
// prof (read only when prof_on): 8 per-lane cycle accumulators (sketch, probes, merge, chain fill, backtrack, regions, align + mapq, pairing)
template <class MS>
PMX_HD int compact_seed_pair(const MS& ms, const Opt& o, const RefIndex& ri, const CRead* rd, const uint32_t* const* amb, int* n_seeds, int* n_first,
                              unsigned long long* prof = nullptr, bool prof_on = false) {
    typedef typename MS::PosT PT;
    *n_seeds = 0;
    *n_first = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long prof_t = prof_on ? (unsigned long long)clock64() : 0ULL;
#endif
    const int k = o.k, w = o.w;
    const int qlen0 = rd[0].len, qlen1 = rd[1].len;
    if (qlen0 > PMX_C_MAXLEN || qlen1 > PMX_C_MAXLEN || qlen0 <= 0 || qlen1 <= 0 || w != PMX_C_W || !(k & 1) || 2 * k + 11 > 64 || k > 255 || !o.is_sr_like)
        return PMX_C_BAIL;
    if (sizeof(PT) == 2 ? ri.len > 32767 : ri.len > 0x3fffffff) return PMX_C_BAIL;   // position << 1 | strand must fit PT
#pragma unroll
    for (int s = 0; s < 2; ++s)   // an ambiguous base anywhere: general tier
        for (int c = 0; c < (rd[s].len + 31) >> 5; ++c)
            if (amb[s][c]) return PMX_C_BAIL;

    // ---------------------------------------------------------------- minimizers -> index probes -> seeds (X, Y)
    CSeeder<MS> sd{ms, ri};
    sd.n_s = 0; sd.bail = false; sd.have_prev = false; sd.prev_key = 0; sd.pending = -1;
    {
        const CRead r0 = rd[0], r1 = rd[1];
        for (int s = 0; s < 2; ++s) {
            sd.r.w = s ? r1.w : r0.w; sd.r.len = s ? r1.len : r0.len; sd.r.flip = s ? r1.flip : r0.flip;
            sd.seg = s; sd.sum = s ? qlen0 : 0;
            sd.n_q = 0; sd.first_of_read = true; sd.ovf = false; sd.cw = 0; sd.ck = -1;
            sd.load_words();
            if (!sketch_distinct<PMX_C_W>(sd.r.len, k, sd, sd)) sd.bail = true;
            PMX_C_STAMP(0);
            if (sd.ovf) sd.bail = true;
            sd.drain(true);
            if (s == 0) *n_first = sd.n_s;
            PMX_C_STAMP(1);
        }
    }
    if (sd.bail) return PMX_C_BAIL;
    ms.flush_tail(sd.n_s);
    *n_seeds = sd.n_s;
    return PMX_C_DONE;
}

}  // namespace aln
}  // namespace pmx
#include "aln_compact_multi.hpp"
namespace pmx {
namespace aln {

// MULTI: follow up to PMX_CM_MAXC fragment chains and several regions per mate (compact_regions_multi) instead of bailing
template <bool MULTI = false, class PT, int CAP>
PMX_HD int compact_chain_pair(const CMemT<PT, CAP>& m, const Opt& o, const RefIndex& ri, const CRead* rd, int n_s, int n_s0, CResult& out, const CPenTab& pen_tab,
                               unsigned long long* prof = nullptr, bool want_edits = false, bool prof_on = false, const SWork* mw = nullptr) {
    typedef CMemT<PT, CAP> MT;
    out.mapped = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long prof_t = prof_on ? (unsigned long long)clock64() : 0ULL;
#endif
    const int k = o.k;
    const int qlen0 = rd[0].len, qlen1 = rd[1].len, qlen_sum = qlen0 + qlen1;
    out.edit[0] = qlen0; out.edit[1] = qlen1;   // a mate without a region counts its whole length
    if (n_s == 0) return PMX_C_DONE;   // no anchors: unmapped
    if (n_s > MT::kCap) return PMX_C_BAIL;   // (the first form's memory holds fewer anchors than the hand-over: the second form takes the pair)

    // ---------------------------------------------------------------- heap merge (map.c:102-166) -> anchors in place
    const int n = n_s;
    {
        // (heap of seed indices and, later, its pop order in the HIGH bytes of G; the destinations go to the LOW bytes)
        // The usual pair first: the seeds of a mate arrive in query order, i.e. with strictly rising (or, on the other
        // strand, strictly falling) reference position words, so the pair is two sorted runs.  While no two words are equal
        // the heap can only pop them in ascending order, which a two-way merge gives in n steps -- GH(n-1-t) = t-th seed
        // out, n_for as below.  Equal words (the same minimizer in both mates where they overlap: the pop order of the
        // tie is the heap's) or a run that is not monotone (a seed from elsewhere) leave the pair to the heap.
        // Mates that overlap on the reference share minimizers -- equal words, one from each run -- in nearly half of the
        // pairs of a 300 +- 30 bp library: for them the merge still gives the ORDER OF THE KEYS, only the order inside a
        // tie is the heap's.  The heap's decisions depend on nothing but comparisons of keys, so it is run on the seeds'
        // RANKS in that order (tie partners share a rank): heap entries rank << 8 | seed in the 16-bit G words, one LDS
        // round trip per level instead of two (entry, then its key through X).
        int n_for = 0;
        bool merged = false, by_rank = false;
        if (n_s0 >= 0 && n_s0 <= n) {
            int ci = n_s0, cj = n - n_s0;
            int i = 0, di = 1, j = n_s0, dj = 1;
            if (ci > 1 && m.X(0) > m.X(ci - 1)) { i = ci - 1; di = -1; }
            if (cj > 1 && m.X(n_s0) > m.X(n - 1)) { j = n - 1; dj = -1; }
            uint32_t xi = ci ? m.X(i) : 0xffffffffu, xj = cj ? m.X(j) : 0xffffffffu;
            bool ok = true, ties = false;
            int t = 0;
            for (; t < n && ok; ++t) {
                ties = ties || xi == xj;
                const bool first = xi <= xj;
                const int idx = first ? i : j;
                m.GH(n - 1 - t) = (c_u8)idx;
                n_for += ((first ? xi : xj) ^ m.Y(idx)) & 1u ? 0 : 1;
                if (first) {
                    --ci; i += di;
                    const uint32_t nx = ci ? m.X(i) : 0xffffffffu;
                    ok = ok && nx > xi;
                    xi = nx;
                } else {
                    --cj; j += dj;
                    const uint32_t nx = cj ? m.X(j) : 0xffffffffu;
                    ok = ok && nx > xj;
                    xj = nx;
                }
            }
            merged = ok && !ties;
            by_rank = ok && ties;
        }
        PMX_C_COUNT(3, merged ? 1 : 0);
        if (by_rank) {
            {   // ranks (the place of a key's first seed in the merge order) into the low bytes
                uint32_t prev = 0xffffffffu;
                int rk = 0;
                for (int t = 0; t < n; ++t) {
                    const int idx = (int)m.GH(n - 1 - t);
                    const uint32_t key = m.X(idx);
                    if (key != prev) rk = t;
                    prev = key;
                    m.GL(idx) = (c_u8)rk;
                }
            }
            for (int i = 0; i < n; ++i) m.G(i) = (c_u16)((uint32_t)m.GL(i) << 8 | (uint32_t)i);   // the heap array: seed i at place i
            auto heapdown = [&](int i, int sz) PMX_LAMBDA_INLINE {   // ks_heapdown, keys = the high bytes
                const uint32_t tmp = m.G(i);
                const uint32_t tk = tmp >> 8;
                int kk;
                while ((kk = (i << 1) + 1) < sz) {
                    uint32_t ce = m.G(kk);
                    if (kk != sz - 1) {
                        const uint32_t c1 = m.G(kk + 1);
                        if ((ce >> 8) > (c1 >> 8)) { ++kk; ce = c1; }
                    }
                    if ((ce >> 8) > tk) break;
                    m.G(i) = (c_u16)ce;
                    i = kk;
                }
                m.G(i) = (c_u16)tmp;
            };
            for (int q = (n >> 1) - 1; q >= 0; --q) heapdown(q, n);
            for (int sz = n; sz > 0;) {
                const uint32_t si = m.G(0);
                const uint32_t last = m.G(sz - 1);
                --sz;
                if (sz > 0) { m.G(0) = (c_u16)last; heapdown(0, sz); }
                m.G(sz) = (c_u16)si;
            }
            for (int p = 0; p < n; ++p) m.G(p) = (c_u16)((uint32_t)m.G(p) << 8);   // GH(n-1-t) = t-th seed out, as below
        } else if (!merged) {
        n_for = 0;
        for (int i = 0; i < n; ++i) m.GH(i) = (c_u8)i;
        auto heapdown = [&](int i, int sz) PMX_LAMBDA_INLINE {   // ks_heapdown with "less" = larger reference position word (min-heap)
            const uint32_t tmp = m.GH(i);
            const uint32_t tk = m.X((int)tmp);
            int kk;
            while ((kk = (i << 1) + 1) < sz) {
                uint32_t ce = m.GH(kk);
                uint32_t ckey = m.X((int)ce);
                if (kk != sz - 1) {
                    const uint32_t c1 = m.GH(kk + 1);
                    const uint32_t k1 = m.X((int)c1);
                    if (ckey > k1) { ++kk; ce = c1; ckey = k1; }
                }
                if (ckey > tk) break;
                m.GH(i) = (c_u8)ce;
                i = kk;
            }
            m.GH(i) = (c_u8)tmp;
        };
        for (int q = (n >> 1) - 1; q >= 0; --q) heapdown(q, n);
        // pops: the root leaves, the last element takes its place (every occurrence list has one entry), the slot the
        // heap gave up keeps the popped seed -> GH(n-1-t) = t-th seed out
        for (int sz = n; sz > 0;) {
            const uint32_t si = m.GH(0);
            const uint32_t last = m.GH(sz - 1);
            --sz;
            if (sz > 0) { m.GH(0) = (c_u8)last; heapdown(0, sz); }
            m.GH(sz) = (c_u8)si;
            n_for += ((m.X((int)si) ^ m.Y((int)si)) & 1u) ? 0 : 1;   // strand of the reference copy == strand of the query copy
        }
        }
        // destinations: forward-strand anchors first, in pop order, then the reverse-strand ones, in pop order
        {
            int df = 0, dr = n_for;
            for (int t = 0; t < n; ++t) {
                const int si = (int)m.GH(n - 1 - t);
                const bool fwd = ((m.X(si) ^ m.Y(si)) & 1u) == 0u;
                m.GL(si) = (c_u8)(fwd ? df : dr);
                df += fwd ? 1 : 0;
                dr += fwd ? 0 : 1;
            }
        }
        // seeds -> anchors, moved along the cycles of the destination map (bit 7 of the destination byte: already placed)
        auto to_anchor = [&](uint32_t pv, uint32_t qy, uint32_t* ax, uint32_t* ay) PMX_LAMBDA_INLINE {
            const uint32_t rpos = pv >> 1, qp = qy & 0x3ffu, fl = qy & (PMX_CQ_SEG | PMX_CQ_TANDEM);
            if ((pv & 1u) == (qp & 1u)) { *ax = rpos; *ay = (qp >> 1) | fl; }
            else { *ax = MT::kRevBit | rpos; *ay = (uint32_t)(qlen_sum - ((int)(qp >> 1) + 1 - k) - 1) | fl; }
        };
        for (int s0 = 0; s0 < n; ++s0) {
            if (m.GL(s0) & 0x80u) continue;
            uint32_t cx, cy;
            to_anchor(m.X(s0), m.Y(s0), &cx, &cy);
            int d = (int)m.GL(s0);
            m.GL(s0) = (c_u8)(d | 0x80);
            while (d != s0) {
                uint32_t nx, ny;
                to_anchor(m.X(d), m.Y(d), &nx, &ny);
                m.setX(d, cx); m.Y(d) = (c_u16)cy;
                cx = nx; cy = ny;
                const int d2 = (int)(m.GL(d) & 0x7fu);
                m.GL(d) = (c_u8)(d2 | 0x80);
                d = d2;
            }
            m.setX(s0, cx); m.Y(s0) = (c_u16)cy;
        }
    }

    PMX_C_STAMP(2);
    // ---------------------------------------------------------------- chain fill (lchain.c:148-230) -> F = G
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
        const bool use_tab = pen_tab.same != nullptr && sp == 0.0f && bw < PMX_C_PEN_SAME && max_dist_x < PMX_C_PEN_DIFF;   // uniform
        int st = 0, max_ii = -1;
        uint32_t ax_st = m.X(0);
        uint64_t x_mi = 0;
        int32_t f_mi = 0;
        // Colinear runs.  Anchors r0 .. i-1 form a run when consecutive ones lie on one diagonal of one mate and strand,
        // are a valid pair for comput_sc, and each has its left neighbour as predecessor.  If anchor i extends the run,
        // the visits of j = i-1 .. r0 have a known outcome: j = i-1 gives sc = f[i-1] + min(dq, k) > k; for every
        // other valid j of the run sc_j = f[j] + min(d(j,i), k) <= f[i-1] + min(d(i-1,i), k), because f[t] >= f[t-1] +
        // min(d(t-1,t), k) along the run and min(a,k) + min(b,k) >= min(a+b,k) -- not better, and marked (its right
        // neighbour was visited and points at it), so it only counts towards max_skip; run members farther than the
        // distance limits are invalid and have no effect at all (d grows with every step down).  The loop below then
        // starts under the run with that state instead of visiting up to max_skip + 2 anchors for nothing.
        const int32_t run_lim = max_dist_x < max_dist_y ? max_dist_x : max_dist_y;
        int r0 = 0, jv = 0;
        uint32_t ax_p = 0, ay_p = 0, f_p = 0;   // anchor i - 1
        for (int i = 0; i < n; ++i) {
            const uint32_t axi = m.X(i);
            const uint32_t ayi = m.Y(i);
            const uint64_t xi = MT::x64(axi);
            const uint32_t rpi = MT::pos_of(axi);
            const int32_t qi = (int32_t)(ayi & 0x3ffu), sidi = (int32_t)(ayi >> 10 & 1u);
            int32_t max_f = k, n_skip = 0, mj = -1;
            while (st < i && (MT::rev_of(axi ^ ax_st) != 0u || xi > MT::x64(ax_st) + (uint64_t)max_dist_x)) {
                ++st;
                ax_st = st < i ? m.X(st) : axi;
            }
            uint64_t mark = 0;   // bit (j - st): anchor j is the predecessor of an anchor already visited for this i
            int32_t ej = st - 1;
            bool stop = false;
            int32_t j_from = i - 1;
            const int32_t dq_p = qi - (int32_t)(ay_p & 0x3ffu);
            const bool extends = use_tab && i > 0 && st <= i - 1 && MT::rev_of(axi ^ ax_p) == 0u && ((ayi ^ ay_p) >> 10 & 1u) == 0u && dq_p > 0 &&
                                 (int32_t)(rpi - MT::pos_of(ax_p)) == dq_p && dq_p <= run_lim;
            if (extends) {
                const int lo = r0 > st ? r0 : st;
                if (jv < lo) jv = lo;
                while (qi - (int32_t)(m.Y(jv) & 0x3ffu) > run_lim) ++jv;   // ends at i - 1 at the latest
                const int32_t n_marked = i - 1 - jv;
                max_f = (int32_t)(f_p & 0x3ffu) + (dq_p < k ? dq_p : k);
                mj = i - 1;
                if (n_marked > max_skip) { ej = i - 2 - max_skip; stop = true; }
                else {
                    n_skip = n_marked;
                    if (jv == r0) {
                        const int32_t pj = (int32_t)(m.G(r0) >> 10) - 1 - st;
                        if (pj >= 0) mark = 1ULL << (pj & 63);
                    }
                    j_from = lo - 1;
                }
            }
            PMX_C_COUNT(0, 1);
            int trace_trips = 0, trace_rescan = 0;
            (void)trace_trips; (void)trace_rescan;
            for (int32_t j = j_from; j >= st && !stop; --j) {
                PMX_C_COUNT(1, 1);
                ++trace_trips;
                const uint32_t axj = m.X(j);
                const uint32_t ayj = m.Y(j);
                const uint32_t fj = m.G(j);
                const int32_t sc0 = use_tab ? c_chain_score_tab(pen_tab, rpi, qi, sidi, MT::pos_of(axj), (int32_t)(ayj & 0x3ffu), (int32_t)(ayj >> 10 & 1u), k,
                                                                max_dist_x, max_dist_y, bw)
                                            : chain_score_sel(rpi, qi, sidi, MT::pos_of(axj), (int32_t)(ayj & 0x3ffu), (int32_t)(ayj >> 10 & 1u), k, max_dist_x,
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
                // j is the top of a run a .. j of OTHER anchors (B[j]): seen from i all of them have one diagonal
                // difference, so one penalty; with dr > 0 at the top (then dg >= 0 and no dr == 0 below) and the
                // bottom within the distance limits every member is valid and sc falls (weakly) from the top down, by
                // the inequality above -- the rest of the run is not better and marked: it only counts skips
                if (use_tab && valid && !stop) {
                    const int32_t a0 = j - (int32_t)(ayj >> 12);   // (run length, at most 15: a longer run is skipped in several steps)
                    const int32_t a = a0 > st ? a0 : st;
                    if (j - a >= 2) {
                        const uint32_t axa = m.X(a);
                        const uint32_t aya = m.Y(a);
                        const int32_t dq_a = qi - (int32_t)(aya & 0x3ffu), dr_a = (int32_t)(rpi - MT::pos_of(axa));
                        const bool same = ((ayi ^ ayj) >> 10 & 1u) == 0u;
                        const bool whole = (int32_t)(rpi - MT::pos_of(axj)) > 0 && dq_a <= max_dist_x && (!same || (dq_a <= max_dist_y && dr_a <= max_dist_y));
                        if (whole) {
                            PMX_C_COUNT(2, 1);
                            const int32_t c = j - a;
                            if (n_skip + c > max_skip) { ej = j - (max_skip + 1 - n_skip); stop = true; }
                            else {
                                n_skip += c;
                                const int32_t pa = (int32_t)(m.G(a) >> 10) - 1 - st;
                                if (pa >= 0) mark |= 1ULL << (pa & 63);
                                j = a;
                            }
                        }
                    }
                }
            }
            int32_t max_j = mj;
            const int32_t end_j = ej;
            // (unsigned, as lchain.c:197 compares: across a strand change the difference is huge and max_ii starts over)
            if (max_ii < 0 || (xi - x_mi) > (uint64_t)(int64_t)max_dist_x) {
                int32_t mx = INT32_MIN;
                max_ii = -1;
                for (int32_t j = i - 1; j >= st; --j) {
                    const int32_t fj = (int32_t)(m.G(j) & 0x3ffu);
                    if (mx < fj) { mx = fj; max_ii = j; }
                    ++trace_rescan;
                }
                if (max_ii >= 0) { x_mi = MT::x64(m.X(max_ii)); f_mi = mx; }
            }
            if (max_ii >= 0 && max_ii < end_j) {
                const uint32_t axm = m.X(max_ii);
                const uint32_t aym = m.Y(max_ii);
                const int32_t tmp = chain_score_sel(rpi, qi, sidi, MT::pos_of(axm), (int32_t)(aym & 0x3ffu), (int32_t)(aym >> 10 & 1u), k, max_dist_x,
                                                    max_dist_y, bw, gp, sp, 2);
                const int32_t fm = (int32_t)(m.G(max_ii) & 0x3ffu);
                if (tmp != INT32_MIN && max_f < tmp + fm) { max_f = tmp + fm; max_j = max_ii; }
            }
            if (max_f < 0 || max_f > 1023) return PMX_C_BAIL;
            PMX_C_TRACE(i, trace_trips, trace_rescan);
            m.G(i) = (c_u16)((uint32_t)max_f | (uint32_t)(max_j + 1) << 10);
            if (max_ii < 0 || ((xi - x_mi) <= (uint64_t)(int64_t)max_dist_x && f_mi < max_f)) { max_ii = i; x_mi = xi; f_mi = max_f; }
            if (!(extends && max_j == i - 1)) { r0 = i; jv = i; }
            { const int rl = i - r0; m.Y(i) = (c_u16)(ayi | (uint32_t)(rl < 15 ? rl : 15) << 12); }   // bits 12..15 of Y: free in this tier
#ifdef PMX_C_DUMP
            if (getenv("PMX_C_DUMP")) fprintf(stderr, "i=%d pos=%u q=%d seg=%d f=%d p=%d r0=%d ext=%d st=%d\n", i, rpi, qi, sidi, max_f, max_j, r0, (int)extends, st);
#endif
            ax_p = axi; ay_p = ayi; f_p = (uint32_t)max_f;
        }
    }

    PMX_C_STAMP(3);
    // ---------------------------------------------------------------- backtrack (lchain.c:27-76) -> B, up to two chains
    // The reference sorts the chain ends by (score, index) and walks them from the top, skipping ends that a kept chain
    // already uses: picking, each time, the largest (score, index) below the previous pick among the unused anchors is
    // the same sequence without the sorted copy.
    int n_u = 0;
    int32_t u_sc0 = 0, u_sc1 = 0;
    uint64_t keep0 = 0, keep1 = 0;   // the members of the two chains (a chain walks to ever smaller indices: the set is the list)

    {
        uint64_t used = 0;
        int n_v = 0;
        uint32_t below = 0xffffffffu;   // (score << 6 | index) of the previous pick
        for (;;) {
            uint32_t best = 0;
            bool found = false;
            for (int i = 0; i < n; ++i) {
                const uint32_t f = m.G(i) & 0x3ffu;
                const uint32_t key = f << 6 | (uint32_t)i;
                if ((int32_t)f >= min_sc && key < below && !(used >> i & 1ULL) && (!found || key > best)) { best = key; found = true; }
            }
            if (!found) break;
            below = best;
            const int i0 = (int)(best & 63u);
            const int32_t zx = (int32_t)(best >> 6);
            const int n_v0 = n_v;
            int i = i0;
            int32_t max_s = 0;
            uint64_t walk = 0, keep = 0;
            do {
                walk |= 1ULL << i;
                i = (int)(m.G(i) >> 10) - 1;
                const int32_t s = i < 0 ? zx : zx - (int32_t)(m.G(i) & 0x3ffu);
                if (s > max_s) { max_s = s; keep = walk; }
                else if (max_s - s > max_drop) break;
            } while (i >= 0 && !(used >> i & 1ULL));
            const int cnt = __builtin_popcountll(keep);
            used |= keep;
            n_v = n_v0 + cnt;
            if (max_s >= min_sc && cnt > 0 && cnt >= min_cnt) {
                if (n_u >= (MULTI ? PMX_CM_MAXC : 2)) return PMX_C_BAIL;   // a third (fifth) chain: general tier
                if (MULTI) { mw->I(PMX_CMI_USC + n_u) = (uint32_t)max_s; s_set64(*mw, PMX_CMI_UKEEP + 2 * n_u, keep); }
                else if (n_u == 0) { u_sc0 = max_s; keep0 = keep; }
                else { u_sc1 = max_s; keep1 = keep; }
                ++n_u;
            } else n_v = n_v0;
        }
    }
    if (n_u == 0) return PMX_C_DONE;   // unmapped
    PMX_C_STAMP(4);
    if (MULTI) return compact_regions_multi(*mw, m, o, ri, rd, n_u, max_chain_gap_ref, out, want_edits);

    // ---------------------------------------------------------------- chains -> one region per mate (hit.c:54-94, 345-400)
    // anchors of segment s in chain c; a mate followed here has exactly one chain (then regs0's parent / secondary logic
    // has nothing to decide: the chains of different mates do not overlap on the fragment), a mate without anchors makes
    // the pair unmapped whatever the other one does
    int c00 = 0, c01 = 0, c10 = 0, c11 = 0;   // c<chain><segment>
    for (uint64_t wk = keep0; wk; wk &= wk - 1) { const uint32_t sg = (m.Y(__builtin_ctzll(wk)) >> 10) & 1u; c00 += sg ? 0 : 1; c01 += sg ? 1 : 0; }
    for (uint64_t wk = keep1; wk; wk &= wk - 1) { const uint32_t sg = (m.Y(__builtin_ctzll(wk)) >> 10) & 1u; c10 += sg ? 0 : 1; c11 += sg ? 1 : 0; }
    if ((c00 == 0 && c10 == 0) || (c01 == 0 && c11 == 0)) return want_edits ? PMX_C_BAIL : PMX_C_DONE;   // a mate without a region: unmapped (the other mate's edit count: general tier)
    if ((c00 > 0 && c10 > 0) || (c01 > 0 && c11 > 0)) return PMX_C_BAIL;        // a mate with two regions
    // per-mate anchor index lists in G (the chain cells are dead): mate 0 from 0, mate 1 behind it, ascending = the
    // chain walked backwards
    CReg R0, R1;
    int base1 = 0;
    {
        int wr = 0;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bool from1 = s == 0 ? c00 == 0 : c01 == 0;   // the mate's chain
            if (s == 1) base1 = wr;
            uint32_t rev = 0;
            for (uint64_t wk = from1 ? keep1 : keep0; wk; wk &= wk - 1) {   // ascending = the chain walked backwards
                const int ai = __builtin_ctzll(wk);
                if ((int)((m.Y(ai) >> 10) & 1u) != s) continue;
                rev = MT::rev_of(m.X(ai));
                m.G(wr++) = (c_u16)ai;
            }
            CReg& r = s == 0 ? R0 : R1;
            r.cnt = s == 0 ? (from1 ? c10 : c00) : (from1 ? c11 : c01);
            r.score = from1 ? u_sc1 : u_sc0;
            r.rev = (int32_t)rev;
            r.has_p = 0; r.dp_score = r.dp_max = 0; r.mapq = 0; r.proper_frag = 0; r.m_len = 0;
        }
    }
    const CList<MT> L0{m, 0, R0.rev ? qlen_sum - qlen0 : 0};                     // hit.c:381: rev ? qlen_sum - (ql + acc) : acc
    const CList<MT> L1{m, base1, R1.rev ? qlen_sum - (qlen1 + qlen0) : qlen0};
    c_reg_set_coor(L0, R0, qlen0, k);
    c_reg_set_coor(L1, R1, qlen1, k);

    PMX_C_STAMP(5);
    // ---------------------------------------------------------------- align each mate, filter, mapq
    {
        bool kept;
        if (c_align1(L0, o, ri, rd[0], qlen0, R0) != PMX_C_DONE) return PMX_C_BAIL;
        if (c_filter_mapq(o, ri, qlen0, R0, &kept) != PMX_C_DONE) return PMX_C_BAIL;
        if (!kept) return want_edits ? PMX_C_BAIL : PMX_C_DONE;   // the mate loses its only region: unmapped pair
        if (c_align1(L1, o, ri, rd[1], qlen1, R1) != PMX_C_DONE) return PMX_C_BAIL;
        if (c_filter_mapq(o, ri, qlen1, R1, &kept) != PMX_C_DONE) return PMX_C_BAIL;
        if (!kept) return want_edits ? PMX_C_BAIL : PMX_C_DONE;
    }

    PMX_C_STAMP(6);
    // ---------------------------------------------------------------- pairing (pe.c:76-177) with one region per mate
    // Two entries sorted by key = rs << 1 | (segment ^ rev) (ties keep mate 0 first).  The scan pairs the SECOND entry with
    // the first iff the first is a "left" end (key bit 0 clear), the second a "right" end on the same strand, the gap
    // between them at most max_gap_ref and their DP scores reach the threshold; with one candidate the pair's mapq is the
    // larger of the two, floored at 2.
    if (o.pe_ori >= 0) {
        const uint64_t key0 = (uint64_t)(uint32_t)(R0.rs << 1) | (uint32_t)(0 ^ R0.rev), key1 = (uint64_t)(uint32_t)(R1.rs << 1) | (uint32_t)(1 ^ R1.rev);
        int dp_thres = R0.dp_max + R1.dp_max - o.pe_bonus;
        if (dp_thres < 0) dp_thres = 0;
        const bool swap = key1 < key0;
        const CReg& A_ = swap ? R1 : R0;
        const CReg& B_ = swap ? R0 : R1;
        const uint64_t ka = swap ? key1 : key0, kb = swap ? key0 : key1;
        const bool proper = !(ka & 1ULL) && (kb & 1ULL) && A_.rev == B_.rev && !(B_.rs - A_.re > max_chain_gap_ref) &&
                            !(B_.dp_max + A_.dp_max < dp_thres) && ((int64_t)(B_.dp_max + A_.dp_max) << 32) > 0;
        if (proper) {
            R0.proper_frag = R1.proper_frag = 1;
            const int mapq_pe = R0.mapq > R1.mapq ? R0.mapq : R1.mapq;
            if (R0.mapq < mapq_pe) R0.mapq = (int)(.2f * R0.mapq + .8f * mapq_pe + .499f);
            if (R1.mapq < mapq_pe) R1.mapq = (int)(.2f * R1.mapq + .8f * mapq_pe + .499f);
            if (R0.mapq < 2) R0.mapq = 2;
            if (R1.mapq < 2) R1.mapq = 2;
        }
    }

    PMX_C_STAMP(7);
    // ---------------------------------------------------------------- the record (src/mm_align.c:271-354)
    if (R0.has_p && R0.blen > 0) out.edit[0] = R0.blen - R0.mlen;   // (no ambiguous base on either side in this tier)
    if (R1.has_p && R1.blen > 0) out.edit[1] = R1.blen - R1.mlen;
    if (!(R0.score > 0 && R1.score > 0)) return PMX_C_DONE;
    out.mapped = 1;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const CReg& r = s == 0 ? R0 : R1;
        CMate& t = out.m[s];
        t.rs = r.rs; t.re = r.re; t.qs = r.qs; t.qe = r.qe;
        t.dp_max = r.dp_max;
        t.cigar = (uint32_t)r.m_len << 4;
        t.mapq = (uint8_t)r.mapq; t.rev = (uint8_t)r.rev; t.proper_frag = (uint8_t)r.proper_frag; t.has_aln = 1;
    }
    return PMX_C_DONE;
}

template <bool MULTI = false, class PT, int CAP>
PMX_HD int compact_map_pair(const CMemT<PT, CAP>& m, const Opt& o, const RefIndex& ri, const CRead* rd, const uint32_t* const* amb, CResult& out,
                             const CPenTab& pen_tab, unsigned long long* prof = nullptr, bool want_edits = false, bool prof_on = false, const SWork* mw = nullptr) {
    out.mapped = 0;
    out.edit[0] = rd[0].len; out.edit[1] = rd[1].len;
    int n_s = 0, n_s0 = 0;
    if (compact_seed_pair(m, o, ri, rd, amb, &n_s, &n_s0, prof, prof_on) != PMX_C_DONE) return PMX_C_BAIL;
    return compact_chain_pair<MULTI>(m, o, ri, rd, n_s, n_s0, out, pen_tab, prof, want_edits, prof_on, mw);
}

}  // namespace aln
}  // namespace pmx
