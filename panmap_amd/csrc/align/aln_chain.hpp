// ALIGN stage, part 2: anchor chaining (mg_lchain_dp, lchain.c:113-230), chain backtracking (:9-76)
// and compaction (:78-111).  Scores mix int32 and float32 arithmetic; the float expressions keep the
// reference's operand order and are compiled without contraction.
#pragma once
#include "aln_sort.hpp"
#include "aln_types.hpp"

namespace pmx {
namespace aln {

// mg_log2 (mmpriv.h:118-126); only valid for x >= 2
PMX_HD float mg_log2f(float x) {
    uint32_t zi;
    memcpy(&zi, &x, 4);
    float log_2 = (float)(((zi >> 23) & 255) - 128);
    zi &= ~(255u << 23);
    zi += 127u << 23;
    float zf;
    memcpy(&zf, &zi, 4);
    log_2 += (-0.34484843f * zf + 2.02466578f) * zf - 0.67487759f;
    return log_2;
}

// comput_sc (lchain.c:113-141; is_cdna == 0): the score of chaining anchor i behind anchor j, evaluated on the anchors'
// fields without early exits (selects instead of branches: in the thread-per-pair kernels 64 lanes evaluate different
// anchor pairs at once, and an early return is a divergent branch).  The integer and float expressions are the
// reference's in the reference's order; what is computed for a pair it rejects is discarded (INT32_MIN).
PMX_HD int32_t chain_score_sel(uint32_t xi, int32_t yi, int32_t sidi, uint32_t xj, int32_t yj, int32_t sidj, int32_t q_span, int32_t max_dist_x,
                               int32_t max_dist_y, int32_t bw, float chn_pen_gap, float chn_pen_skip, int n_seg) {
    const int32_t dq = yi - yj;
    const int32_t dr = (int32_t)(xi - xj);
    const bool same = sidi == sidj;
    const int32_t dd = dr > dq ? dr - dq : dq - dr;
    bool bad = dq <= 0 || dq > max_dist_x;
    bad = bad || (same && (dr == 0 || dq > max_dist_y));
    bad = bad || (same && dd > bw);
    bad = bad || (n_seg > 1 && same && dr > max_dist_y);
    const int32_t dg = dr < dq ? dr : dq;
    int32_t sc = q_span < dg ? q_span : dg;
    const bool pen = dd != 0 || dg > q_span;
    const float lin_pen = chn_pen_gap * (float)dd + chn_pen_skip * (float)dg;
    const float log_pen = dd >= 1 ? mg_log2f((float)((uint32_t)dd + 1u)) : 0.0f;
    const float pen_same = lin_pen + .5f * log_pen, pen_diff = lin_pen < log_pen ? lin_pen : log_pen;
    // (int) of a float: only meaningful (and only used) for pairs that are not rejected; clamp keeps the cast defined
    const float pf = same ? pen_same : pen_diff;
    const float pc = pf < -1.0e9f ? -1.0e9f : (pf > 1.0e9f ? 1.0e9f : pf);
    const int32_t sc_pen = (!same && dr == 0) ? sc + 1 : sc - (int)pc;
    sc = pen ? sc_pen : sc;
    return bad ? INT32_MIN : sc;
}

// the same for two anchors in the reference's packing
PMX_HD int32_t chain_score(const A128 ai, const A128 aj, int32_t max_dist_x, int32_t max_dist_y, int32_t bw, float chn_pen_gap,
                           float chn_pen_skip, int n_seg) {
    return chain_score_sel((uint32_t)ai.x, (int32_t)ai.y, (int32_t)((ai.y & PMX_SEED_SEG_MASK) >> PMX_SEED_SEG_SHIFT), (uint32_t)aj.x, (int32_t)aj.y,
                           (int32_t)((aj.y & PMX_SEED_SEG_MASK) >> PMX_SEED_SEG_SHIFT), (int32_t)(aj.y >> 32 & 0xff), max_dist_x, max_dist_y, bw,
                           chn_pen_gap, chn_pen_skip, n_seg);
}

#if PMX_W > 1
__device__ __forceinline__ int32_t rl32(int32_t v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ uint64_t rl64(uint64_t v, int lane) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return (uint64_t)hi << 32 | lo;
}

// Fill phase of mg_lchain_dp for n <= 64 anchors, one anchor per lane: for anchor i every predecessor
// lane evaluates comput_sc in parallel; the order-dependent part of the reference loop (strict-max
// update, the t[]-mark / n_skip early exit, lchain.c:176-190) is replayed sequentially over the lanes'
// values with scalar reads.  t[j]==i marks only ever compare against the current i, so they live in a
// 64-bit mask that is reset per anchor.  Writes f/p/v exactly as the generic loop would.
__device__ void chain_fill_wave(Work& W, const Opt& o, int max_dist_x, int max_dist_y, int n_seg) {
    const int n = (int)W.n_a;
    const int lane = lane_id();
    const int bw = o.bw, max_skip = o.max_chain_skip;
    const float gp = o.chn_pen_gap, sp = o.chn_pen_skip;
    PMX_LDS(&W);
    A128* a_ = W.a; PMX_LDS(a_);
    A128 mine;
    mine.x = lane < n ? a_[lane].x : 0;
    mine.y = lane < n ? a_[lane].y : 0;
    int32_t fj = 0, pj = -1, vj = 0;
    int st = 0, max_ii = -1;
    for (int i = 0; i < n; ++i) {
        A128 ai;
        ai.x = rl64(mine.x, i);
        ai.y = rl64(mine.y, i);
        // advance st (lchain.c:173): first lane >= st that is on the same strand/target and within max_dist_x
        {
            const bool far = (ai.x >> 32 != mine.x >> 32) || (ai.x > mine.x + (uint64_t)max_dist_x);
            const unsigned long long ok = __ballot(!far && lane >= st && lane < i);
            st = ok ? (int)__builtin_ctzll(ok) : i;
        }
        const bool in_rng = lane >= st && lane < i;
        int32_t sc = INT32_MIN;
        if (in_rng) sc = chain_score(ai, mine, max_dist_x, max_dist_y, bw, gp, sp, n_seg);
        const unsigned long long valid = __ballot(in_rng && sc != INT32_MIN);
        const int32_t val = sc != INT32_MIN ? sc + fj : INT32_MIN;
        int32_t max_f = (int32_t)(ai.y >> 32 & 0xff), n_skip = 0;
        int max_j = -1;
        unsigned long long mark = 0;
        int j;
        for (j = i - 1; j >= st; --j) {
            if (!(valid >> j & 1)) continue;
            const int32_t s_ = rl32(val, j);
            if (s_ > max_f) {
                max_f = s_;
                max_j = j;
                if (n_skip > 0) --n_skip;
            } else if (mark >> j & 1) {
                if (++n_skip > max_skip) break;
            }
            const int32_t pp = rl32(pj, j);
            if (pp >= 0) mark |= 1ULL << pp;
        }
        const int end_j = j;
        if (max_ii < 0 || (ai.x - rl64(mine.x, max_ii < 0 ? 0 : max_ii)) > (uint64_t)(int64_t)max_dist_x) {
            // max f over [st, i-1]; equal values keep the largest j (scan runs downwards with strict <)
            int64_t key = in_rng ? ((int64_t)fj << 32 | (uint32_t)lane) : INT64_MIN;
            for (int ofs = 32; ofs > 0; ofs >>= 1) {
                const int64_t other = __shfl_xor(key, ofs);
                key = other > key ? other : key;
            }
            max_ii = key == INT64_MIN ? -1 : (int)(uint32_t)key;
        }
        if (max_ii >= 0 && max_ii < end_j) {
            // NB: max_ii may lie before st (e.g. the last forward-strand anchor when i is the first
            // reverse-strand one: the unsigned difference wraps negative and skips the refresh above),
            // so the score is evaluated here, not read from the lanes' in-range values
            A128 am;
            am.x = rl64(mine.x, max_ii);
            am.y = rl64(mine.y, max_ii);
            const int32_t tmp = chain_score(ai, am, max_dist_x, max_dist_y, bw, gp, sp, n_seg), fm = rl32(fj, max_ii);
            if (tmp != INT32_MIN && max_f < tmp + fm) { max_f = tmp + fm; max_j = max_ii; }
        }
        const int32_t v_prev = max_j >= 0 ? rl32(vj, max_j) : 0;
        const int32_t v_i = max_j >= 0 && v_prev > max_f ? v_prev : max_f;
        if (lane == i) { fj = max_f; pj = max_j; vj = v_i; }
        if (max_ii < 0) max_ii = i;
        else {
            const uint64_t xm = rl64(mine.x, max_ii);
            const int32_t fm = rl32(fj, max_ii);
            if ((ai.x - xm) <= (uint64_t)(int64_t)max_dist_x && fm < max_f) max_ii = i;
        }
    }
    {
        ChainCell* c_ = W.cc; PMX_LDS(c_);
        if (lane < n) { c_[lane].f = fj; c_[lane].p = pj; c_[lane].v = vj; }
    }
    wave_sync();
}
#endif

// The fill for more than 64 anchors (every model) and for the scalar models: everything the inner loop needs from anchor j --
// reference position, query position, span, segment, f, p, the t mark -- packed into ONE 16-byte cell, so an
// inner iteration needs one 16-byte cell instead of an anchor load plus a cell load.  Four cells form one 64-byte
// block (Cell4: its own interleave granule in the thread-per-pair arena, so a lane's block is ONE contiguous line):
// the inner loop walks the predecessors block by block -- four cells per line fetched, four loads in flight -- and
// the "t" marks of the reference (t[p[j]] = i; tested as t[j] == i, i.e. only ever against the CURRENT i) live in
// a 64-bit register mask relative to st whenever the window [st, i) has at most 64 anchors, so the fill does no
// scattered read-modify-write at all (those 2-byte stores were ~40 % of the kernel's HBM traffic).
// Valid while positions in the query fit 16 bits and there are fewer than 65,535 anchors.
struct alignas(16) PackedCell {
    uint32_t x_lo;       // low 32 bits of anchor x (comput_sc only uses the 32-bit difference)
    uint16_t y_lo;       // query position
    uint8_t span, seg;
    int32_t f;
    uint16_t p1;         // p + 1 (0 = none)
    uint16_t t;
};
static_assert(sizeof(PackedCell) == 16, "one interleave granule");
struct alignas(16) Cell4 {
    static constexpr bool kWide64 = true;   // 64-byte interleave granule (aln_types.hpp IGranule)
    PackedCell c[4];
};
static_assert(sizeof(Cell4) == 64, "four cells per block");
PMX_HD PackedCell& packed_cell(Ptr<Cell4> pk4, int64_t j) { return pk4[j >> 2].c[j & 3]; }
// one cell as four 32-bit words, fetched with ONE 16-byte load (a struct copy is split into per-field loads)
struct CellWords { uint32_t w0, w1, w2, w3; };   // x_lo | y_lo, span, seg | f | p1, t
PMX_HD CellWords packed_cell_words(Ptr<Cell4> pk4, int64_t g, int s_) {
    const PackedCell& c = pk4[g].c[s_];
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint32_t pmx_u32x4 __attribute__((ext_vector_type(4)));
    const pmx_u32x4 v = *reinterpret_cast<const pmx_u32x4*>(&c);
    return CellWords{v.x, v.y, v.z, v.w};
#else
    CellWords r;
    memcpy(&r, &c, 16);
    return r;
#endif
}
PMX_HD A128 packed_anchor(const PackedCell& c) {
    A128 r;
    r.x = c.x_lo;
    r.y = (uint64_t)c.seg << PMX_SEED_SEG_SHIFT | (uint64_t)c.span << 32 | c.y_lo;
    return r;
}

PMX_HDN void chain_finish(Work& W, const int64_t n, const int min_cnt, const int min_sc, const int32_t max_drop);

#if PMX_W > 1
// Fill phase of mg_lchain_dp for more than 64 anchors in the wave kernels (long reads: ~1,000 anchors, every earlier one in
// range): the predecessors of anchor i are taken 64 at a time, nearest first -- lane l evaluates comput_sc for anchor
// hi - l from its packed cell -- and the order-dependent part of the reference loop (strict-max update, the t[] marks with
// the n_skip early exit, lchain.c:176-190) is replayed over the lanes' values with scalar reads, as chain_fill_wave does
// for short lists.  A mark t[p[j]] = i that lands inside the chunk being replayed is a bit of a register mask, any other
// goes to the cell's t field in memory (read back by the chunk that holds it).  The sequential form above pays ~60
// instructions per visited predecessor (~26 of them per anchor before the skip limit ends the scan), this one ~10.
__device__ void chain_fill_wide_wave(Work& W, const Opt& o, int max_dist_x, int max_dist_y, int n_seg) {
    const int64_t n = W.n_a;
    const int lane = lane_id();
    const int bw = o.bw, max_skip = o.max_chain_skip, max_iter = o.max_chain_iter;
    const float chn_pen_gap = o.chn_pen_gap, chn_pen_skip = o.chn_pen_skip;
    PMX_LDS(&W);
    Ptr<A128> a = W.a; PMX_LDS(a);
    Ptr<ChainCell> c = W.cc; PMX_LDS(c);
    Ptr<Cell4> pk4 = ptr_region_cast<Cell4>(W.z); PMX_LDS(pk4);
    int64_t st = 0, max_ii = -1;
    uint64_t x_st = a[0].x, x_mi = 0;
    int32_t f_mi = 0;
    A128 a_next = a[0];   // anchor i + 1 is requested while anchor i is worked on (the anchors live in the wave's HBM slab)
    for (int64_t i = 0; i < n; ++i) {
        int64_t max_j = -1;
        const A128 ai = a_next;
        if (i + 1 < n) a_next = a[i + 1];
        // the first 64 candidates' cells too: their addresses depend on i alone (the window start only moves up, so
        // whatever lies below it is masked afterwards) -- the load travels while the window start is advanced
        CellWords q_first;
        q_first.w0 = q_first.w1 = q_first.w2 = q_first.w3 = 0;
        const int64_t j_first = i - 1 - lane;
        if (j_first >= st) q_first = packed_cell_words(pk4, j_first >> 2, (int)(j_first & 3));
        PackedCell ci;
        ci.x_lo = (uint32_t)ai.x;
        ci.y_lo = (uint16_t)(uint32_t)ai.y;
        ci.span = (uint8_t)(ai.y >> 32 & 0xff);
        ci.seg = (uint8_t)((ai.y & PMX_SEED_SEG_MASK) >> PMX_SEED_SEG_SHIFT);
        ci.f = 0; ci.p1 = 0; ci.t = 0;
        const A128 ri = packed_anchor(ci);
        int32_t max_f = (int32_t)ci.span, n_skip = 0;
        while (st < i && (ai.x >> 32 != x_st >> 32 || ai.x > x_st + (uint64_t)max_dist_x)) {
            ++st;
            x_st = st < i ? a[st].x : ai.x;
        }
        if (i - st > max_iter) { st = i - max_iter; x_st = a[st].x; }
        int64_t end_j = st - 1;
        bool stop = false;
        for (int64_t hi = i - 1; hi >= st && !stop; hi -= PMX_W) {
            const int64_t j = hi - lane;
            const bool act = j >= st;
            int32_t val = INT32_MIN, pj = -1;
            bool valid = false, marked = false;
            if (act) {
                const CellWords q = hi == i - 1 ? q_first : packed_cell_words(pk4, j >> 2, (int)(j & 3));
                const int32_t sc0 = chain_score_sel(ci.x_lo, (int32_t)ci.y_lo, (int32_t)ci.seg, q.w0, (int32_t)(q.w1 & 0xffffu), (int32_t)(q.w1 >> 24),
                                                    (int32_t)(q.w1 >> 16 & 0xffu), max_dist_x, max_dist_y, bw, chn_pen_gap, chn_pen_skip, n_seg);
                valid = sc0 != INT32_MIN;
                val = valid ? sc0 + (int32_t)q.w2 : INT32_MIN;
                pj = (int32_t)(q.w3 & 0xffffu) - 1;
                marked = (q.w3 >> 16) == ((uint32_t)i & 0xffffu);
            }
            // The order-dependent part of the reference loop (strict-max update, the t[]-mark / n_skip early exit, lchain.c:176-190)
            // over the chunk's candidates, lane 0 = the first visited.  What a candidate sees of the ones before it is a prefix
            // maximum of their scores (is it a new best?) and a prefix OR of the marks they leave on their predecessors (is it
            // marked?): two scans over the lanes, after which only the EVENTS -- a new best, a marked candidate that is not one
            // -- have to be walked in order for n_skip (a long read's anchor has ~60 valid candidates per chunk and a
            // handful of events; the walk over all of them was 11 M of a 10 kb read's 66 M cycles).  Everything before the
            // stopping candidate is processed in full, so the prefixes are exact up to it; nothing after it is used.
            int32_t pm = val;                                  // inclusive prefix maximum of the scores
            unsigned long long bit = 0ULL;
            int64_t dl_self = 0;
            if (valid && pj >= 0) { dl_self = hi - (int64_t)pj; if (dl_self < PMX_W) bit = 1ULL << dl_self; }
            unsigned long long po = bit;                       // inclusive prefix OR of the in-chunk marks
            for (int o = 1; o < PMX_W; o <<= 1) {
                const int32_t tm = __shfl_up(pm, o);
                const uint32_t tlo = (uint32_t)__shfl_up((int)(uint32_t)po, o), thi = (uint32_t)__shfl_up((int)(uint32_t)(po >> 32), o);
                if (lane >= o) { pm = tm > pm ? tm : pm; po |= (unsigned long long)thi << 32 | tlo; }
            }
            int32_t pm_ex = __shfl_up(pm, 1);
            uint32_t elo = (uint32_t)__shfl_up((int)(uint32_t)po, 1), ehi = (uint32_t)__shfl_up((int)(uint32_t)(po >> 32), 1);
            unsigned long long po_ex = (unsigned long long)ehi << 32 | elo;
            if (lane == 0) { pm_ex = INT32_MIN; po_ex = 0ULL; }
            const int32_t before = pm_ex > max_f ? pm_ex : max_f;   // the best score when this candidate is visited
            const bool better = valid && val > before;
            const unsigned long long mark_all = __ballot(marked);
            const bool is_marked = valid && !better && (((mark_all | po_ex) >> lane) & 1ULL) != 0ULL;
            const unsigned long long ev_better = __ballot(better);
            unsigned long long events = ev_better | __ballot(is_marked);
            int stop_lane = PMX_W;                             // the candidate the reference breaks at (none: PMX_W)
            while (events) {
                const int l = (int)__builtin_ctzll(events);
                events &= events - 1;
                if (ev_better >> l & 1ULL) {
                    max_f = rl32(val, l);
                    max_j = hi - l;
                    if (n_skip > 0) --n_skip;
                } else if (++n_skip > max_skip) { stop = true; end_j = hi - l; stop_lane = l; break; }
            }
            // marks on predecessors beyond this chunk: every candidate visited before the stop leaves one (the reference breaks
            // before marking p[j] of the candidate it stops at)
            if (valid && pj >= 0 && dl_self >= PMX_W && lane < stop_lane) packed_cell(pk4, (int64_t)pj).t = (uint16_t)i;
        }
        // (unsigned, as lchain.c:197 compares: across a strand or target change the difference is huge and max_ii starts over)
        if (max_ii < 0 || (ai.x - x_mi) > (uint64_t)(int64_t)max_dist_x) {
            int64_t key = INT64_MIN;   // max f over [st, i-1]; equal values keep the largest j
            for (int64_t j = i - 1 - lane; j >= st; j -= PMX_W) {
                const int64_t k = (int64_t)packed_cell(pk4, j).f << 32 | (uint32_t)j;
                key = k > key ? k : key;
            }
            for (int ofs = 32; ofs > 0; ofs >>= 1) {
                const int64_t other = __shfl_xor(key, ofs);
                key = other > key ? other : key;
            }
            max_ii = key == INT64_MIN ? -1 : (int64_t)(uint32_t)key;
            if (max_ii >= 0) { x_mi = a[max_ii].x; f_mi = (int32_t)(key >> 32); }
        }
        if (max_ii >= 0 && max_ii < end_j) {
            const PackedCell cm = packed_cell(pk4, max_ii);
            const int32_t tmp = chain_score(ri, packed_anchor(cm), max_dist_x, max_dist_y, bw, chn_pen_gap, chn_pen_skip, n_seg);
            if (tmp != INT32_MIN && max_f < tmp + cm.f) { max_f = tmp + cm.f; max_j = max_ii; }
        }
        {
            ChainCell co;
            co.f = max_f;
            co.p = (int32_t)max_j;
            co.t = 0;
            co.v = max_f;   // (the reference's v[] -- the peak score along the chain -- is never read: mg_chain_backtrack takes
                            //  the array as its output buffer, lchain.c:213; keeping it up cost one dependent HBM load per anchor)
            c[i] = co;
            ci.f = max_f;
            ci.p1 = (uint16_t)(max_j + 1);
            packed_cell(pk4, i) = ci;
        }
        if (max_ii < 0 || ((ai.x - x_mi) <= (uint64_t)(int64_t)max_dist_x && f_mi < max_f)) { max_ii = i; x_mi = ai.x; f_mi = max_f; }
    }
}
#endif

// mg_lchain_dp (lchain.c:148-230) followed by mg_chain_backtrack (:27-76) and compact_a (:78-111).
// In: W.a[0..n_a) sorted anchors.  Out: W.a holds the chained anchors grouped by chain, W.u[0..n_u)
// = score<<32 | count, chains ordered by the target position of their first anchor.
PMX_HDN void chain_dp(Work& W, const Opt& o, int max_dist_x, int max_dist_y, int n_seg) {
    const int64_t n = W.n_a;
    const int bw = o.bw, max_skip = o.max_chain_skip, max_iter = o.max_chain_iter, min_cnt = o.min_cnt, min_sc = o.min_chain_score;
    const float chn_pen_gap = o.chn_pen_gap, chn_pen_skip = o.chn_pen_skip;
    PMX_LDS(&W);
    W.n_u = 0;
    if (n == 0) return;
    Ptr<A128> a = W.a; PMX_LDS(a);
    Ptr<ChainCell> c = W.cc; PMX_LDS(c);
    const int32_t max_drop = bw;
    if (max_dist_x < bw) max_dist_x = bw;
    if (max_dist_y < bw) max_dist_y = bw;
#if PMX_W > 1
    const bool wave_fill = n <= 64;
    if (wave_fill) chain_fill_wave(W, o, max_dist_x, max_dist_y, n_seg);
#else
    const bool wave_fill = false;
#endif
    // (wave models: the scalar fill runs redundantly-uniform on the 64 lanes, like the rest of the bookkeeping)
    bool packed_fill = false;
    if (!wave_fill) {
        int qsum = 0;
        for (int sg = 0; sg < W.n_segs; ++sg) qsum += W.qlen[sg];
        packed_fill = n < 65535 && qsum < 65536;
        if (!packed_fill) {   // reads of 64 kb and more: the packed cells do not hold their positions
            W.status |= PMX_ST_UNSUPPORTED;
            W.n_a = 0;
            return;
        }
    }
#if PMX_W > 1
    if (packed_fill) {
        chain_fill_wide_wave(W, o, max_dist_x, max_dist_y, n_seg);
        packed_fill = false;
    }
#endif
    if (packed_fill) {
        Ptr<Cell4> pk4 = ptr_region_cast<Cell4>(W.z); PMX_LDS(pk4);   // z[] is idle until the backtrack
        int64_t st = 0, max_ii = -1;
        uint64_t x_st = a[0].x, x_mi = 0;   // a[st].x and a[max_ii].x, kept in registers
        int32_t f_mi = 0;                   // f[max_ii]
        for (int64_t i = 0; i < n; ++i) {
            int64_t max_j = -1;
            const A128 ai = a[i];
            PackedCell ci;
            ci.x_lo = (uint32_t)ai.x;
            ci.y_lo = (uint16_t)(uint32_t)ai.y;
            ci.span = (uint8_t)(ai.y >> 32 & 0xff);
            ci.seg = (uint8_t)((ai.y & PMX_SEED_SEG_MASK) >> PMX_SEED_SEG_SHIFT);
            ci.f = 0; ci.p1 = 0; ci.t = 0;
            const A128 ri = packed_anchor(ci);
            int32_t max_f = (int32_t)ci.span, n_skip = 0;
            while (st < i && (ai.x >> 32 != x_st >> 32 || ai.x > x_st + (uint64_t)max_dist_x)) {
                ++st;
                x_st = st < i ? a[st].x : ai.x;
            }
            if (i - st > max_iter) { st = i - max_iter; x_st = a[st].x; }
            const bool use_mask = i - st <= 64;
            uint64_t mark = 0;   // bit (k - st): anchor k is the predecessor of an anchor already visited for this i
            int64_t end_j = st - 1;
            bool stop = false;
            if (use_mask) {
                // window of at most 64 anchors: marks in the register mask, the whole body is selects
                const int32_t i32 = (int32_t)i, st32 = (int32_t)st;
                int32_t mj = -1, ej = st32 - 1;
                for (int32_t g = (i32 - 1) >> 2; i32 > st32 && !stop && g >= (st32 >> 2); --g) {
                    CellWords cw[4];
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_) cw[s_] = packed_cell_words(pk4, g, s_);
#pragma unroll
                    for (int s_ = 3; s_ >= 0; --s_) {
                        const int32_t j = g * 4 + s_;
                        const CellWords q = cw[s_];
                        const int32_t sc0 = chain_score_sel(ci.x_lo, (int32_t)ci.y_lo, (int32_t)ci.seg, q.w0, (int32_t)(q.w1 & 0xffffu),
                                                            (int32_t)(q.w1 >> 24), (int32_t)(q.w1 >> 16 & 0xffu), max_dist_x, max_dist_y, bw,
                                                            chn_pen_gap, chn_pen_skip, n_seg);
                        const bool valid = !stop && j < i32 && j >= st32 && sc0 != INT32_MIN;
                        const int32_t sc = sc0 + (int32_t)q.w2;
                        const bool better = valid && sc > max_f;
                        const bool marked = valid && !better && (mark >> ((j - st32) & 63) & 1) != 0;
                        max_f = better ? sc : max_f;
                        mj = better ? j : mj;
                        n_skip += (better && n_skip > 0) ? -1 : 0;
                        n_skip += marked ? 1 : 0;
                        const bool brk = marked && n_skip > max_skip;   // the reference breaks before marking p[j]
                        ej = brk ? j : ej;
                        stop = stop || brk;
                        const int32_t kk = (int32_t)(q.w3 & 0xffffu) - 1 - st32;   // p[j] relative to st (p1 == 0: none)
                        mark |= (valid && !brk && (q.w3 & 0xffffu) != 0u && kk >= 0) ? 1ULL << (kk & 63) : 0ULL;
                    }
                }
                max_j = mj;
                end_j = ej;
            } else
            for (int64_t g = (i - 1) >> 2; i > st && !stop && g >= (st >> 2); --g) {   // wide window: marks in the cells' t field
                const Cell4 G = pk4[g];
#pragma unroll
                for (int s_ = 3; s_ >= 0; --s_) {
                    const int64_t j = g * 4 + s_;
                    if (stop || j >= i || j < st) continue;
                    const PackedCell cj = G.c[s_];
                    int32_t sc = chain_score(ri, packed_anchor(cj), max_dist_x, max_dist_y, bw, chn_pen_gap, chn_pen_skip, n_seg);
                    if (sc == INT32_MIN) continue;
                    sc += cj.f;
                    if (sc > max_f) {
                        max_f = sc;
                        max_j = j;
                        if (n_skip > 0) --n_skip;
                    } else if (packed_cell(pk4, j).t == (uint16_t)i) {
                        if (++n_skip > max_skip) { stop = true; end_j = j; continue; }
                    }
                    if (cj.p1) packed_cell(pk4, (int64_t)cj.p1 - 1).t = (uint16_t)i;
                }
            }
            // (unsigned, as lchain.c:197 compares: across a strand or target change the difference is huge and max_ii starts over)
            if (max_ii < 0 || (ai.x - x_mi) > (uint64_t)(int64_t)max_dist_x) {
                int32_t mx = INT32_MIN;
                max_ii = -1;
                for (int64_t j = i - 1; j >= st; --j) {
                    const int32_t fj = packed_cell(pk4, j).f;
                    if (mx < fj) { mx = fj; max_ii = j; }
                }
                if (max_ii >= 0) { x_mi = a[max_ii].x; f_mi = mx; }
            }
            if (max_ii >= 0 && max_ii < end_j) {
                const PackedCell cm = packed_cell(pk4, max_ii);
                const int32_t tmp = chain_score(ri, packed_anchor(cm), max_dist_x, max_dist_y, bw, chn_pen_gap, chn_pen_skip, n_seg);
                if (tmp != INT32_MIN && max_f < tmp + cm.f) { max_f = tmp + cm.f; max_j = max_ii; }
            }
            {
                const int32_t vm = max_j >= 0 ? c[max_j].v : 0;
                ChainCell co;
                co.f = max_f;
                co.p = (int32_t)max_j;
                co.t = 0;
                co.v = max_j >= 0 && vm > max_f ? vm : max_f;
                c[i] = co;
                ci.f = max_f;
                ci.p1 = (uint16_t)(max_j + 1);
                packed_cell(pk4, i) = ci;
            }
            if (max_ii < 0 || ((ai.x - x_mi) <= (uint64_t)(int64_t)max_dist_x && f_mi < max_f)) { max_ii = i; x_mi = ai.x; f_mi = max_f; }
        }
    }
    wave_sync();

    PMX_STAMP(W, 18);
    chain_finish(W, n, min_cnt, min_sc, max_drop);
}

// mg_chain_backtrack (lchain.c:27-76) + compact_a (:78-111) on the filled cells W.cc[0..n): shared by the two fills
// (chain_dp above, chain_rmq in aln_rmq.hpp)
PMX_HDN void chain_finish(Work& W, const int64_t n, const int min_cnt, const int min_sc, const int32_t max_drop) {
    PMX_LDS(&W);
    Ptr<A128> a = W.a; PMX_LDS(a);
    Ptr<ChainCell> c = W.cc; PMX_LDS(c);
    // ---- backtrack (lchain.c:27-76)
    Ptr<A128> z = W.z; PMX_LDS(z);
    int64_t n_z = 0;
#if PMX_W == 64 && defined(__HIP_DEVICE_COMPILE__)
    // (long reads: ~1,800 cells in the wave's HBM slab -- 64 at a time, places from a ballot, same order)
    for (int64_t i0 = 0; i0 < n; i0 += 64) {
        const int64_t i = i0 + lane_id();
        const int32_t f = i < n ? c[i].f : 0;
        const bool take = i < n && f >= min_sc;
        const unsigned long long m = __ballot(take);
        if (take) {
            const int64_t at = n_z + (int64_t)__builtin_popcountll(m & ((1ULL << lane_id()) - 1ULL));
            A128 e; e.x = (uint64_t)(int64_t)f; e.y = (uint64_t)i;
            z[at] = e;
        }
        n_z += (int64_t)__builtin_popcountll(m);
    }
    wave_sync();
#else
    for (int64_t i = 0; i < n; ++i)
        if (c[i].f >= min_sc) { z[n_z].x = (uint64_t)(int64_t)c[i].f; z[n_z].y = (uint64_t)i; ++n_z; }
#endif
    if (n_z == 0) { W.n_a = 0; return; }
    radix_sort_128x(z, z + n_z, &W.status);
    int64_t n_v = 0;
    int32_t n_u = 0;
    Ptr<uint64_t> u = W.u; PMX_LDS(u);
    if (n <= 64) {
        // Up to 64 anchors: the t[] marks of mg_chain_backtrack / mg_chain_bk_end (0 unused, 1 in an emitted trace,
        // 2 on the walk in progress) are one bit mask in a register, and the three passes over a chain (walk to the
        // drop point, un-mark, trace up to the best prefix) collapse into one walk: every visited anchor is written
        // to v[] as it is met, and the prefix that ends at the best-scoring cut is kept.
        //  * the walk stops at an anchor already used, at the chain start, or max_drop below the best cut;
        //  * the score of the kept prefix is the best cut's s = f[end] - f[cut] (the value mg_chain_bk_end maximised);
        //  * the reference marks the kept prefix as used even when the chain is then rejected: same here.
        uint64_t used = 0;
        for (int64_t k = n_z - 1; k >= 0; --k) {
            A128 zk;
            for (; k >= 0; --k) {   // scan to the next unused chain end: the lanes of a wave meet at the walk
                zk = z[k];
                if (!(used >> zk.y & 1)) break;
            }
            if (k < 0) break;
            const int32_t zx = (int32_t)zk.x;
            const int64_t n_v0 = n_v;
            int64_t i = (int64_t)zk.y;
            int32_t max_s = 0;
            uint64_t walk = 0, keep = 0;   // anchors visited so far / visited before the best cut
            do {
                walk |= 1ULL << i;
                c[n_v0 + (int64_t)__builtin_popcountll(walk) - 1].v = (int32_t)i;
                i = c[i].p;
                const int32_t s = i < 0 ? zx : zx - c[i].f;
                if (s > max_s) { max_s = s; keep = walk; }
                else if (max_s - s > max_drop) break;
            } while (i >= 0 && !(used >> i & 1));
            const int64_t cnt = (int64_t)__builtin_popcountll(keep);
            used |= keep;
            n_v = n_v0 + cnt;
            if (max_s >= min_sc && cnt > 0 && cnt >= min_cnt) {
                if (n_u < W.caps.max_reg * 4) u[n_u++] = (uint64_t)(uint32_t)max_s << 32 | (uint64_t)cnt;
                else { W.status |= PMX_ST_OVERFLOW; n_v = n_v0; }
            } else n_v = n_v0;
        }
    } else {
        // More than 64 anchors: the same single walk with the "used" marks in the cells' t field (0 free, 1 in a kept chain).
        // The reference walks a chain to its drop point with temporary marks, un-marks it and traces the best prefix in a
        // second pass; the predecessor links strictly decrease, so a walk never meets its own marks and one pass that
        // remembers how many of the visited anchors precede the best cut is the same thing.
#if PMX_W == 64 && defined(__HIP_DEVICE_COMPILE__)
        // Wave kernels (long reads: the cells live in the wave's HBM slab, every access a memory round trip).  Everything
        // but the walk along the predecessor links is spread over the lanes -- clearing the marks, finding the next chain
        // end that is still free (64 candidates per round trip: after the main chain of a long read nearly every remaining
        // end is used), marking a kept chain -- and the walk itself reads ONE 16-byte cell per step (f, p and t of the
        // predecessor arrive together) instead of three dependent fields.
        for (int64_t i = lane_id(); i < n; i += 64) c[i].t = 0;
        wave_sync();
        for (int64_t k = n_z - 1; k >= 0; --k) {
            A128 zk;
            {
                bool found = false;
                while (k >= 0) {
                    const int64_t kk = k - lane_id();
                    A128 e; e.x = 0; e.y = 0;
                    bool free_end = false;
                    if (kk >= 0) { e = z[kk]; free_end = c[(int64_t)e.y].t == 0; }
                    const unsigned long long m = __ballot(free_end);
                    if (m == 0ULL) { k -= 64; continue; }
                    const int first = __builtin_ctzll(m);
                    k -= first;
                    zk.x = rl64(e.x, first);
                    zk.y = rl64(e.y, first);
                    found = true;
                    break;
                }
                if (!found) break;
            }
            int64_t i = (int64_t)zk.y;
            const int32_t zx = (int32_t)zk.x;
            const int64_t n_v0 = n_v;
            int64_t seen = 0, keep = 0;
            int32_t max_s = 0;
            ChainCell cur = c[i];
            do {
                c[n_v0 + seen].v = (int32_t)i;
                ++seen;
                i = cur.p;
                ChainCell nx; nx.f = 0; nx.p = -1; nx.t = 0; nx.v = 0;
                if (i >= 0) nx = c[i];   // (one load: the walk needs f and t of the predecessor now and its p next)
                const int32_t s_ = i < 0 ? zx : zx - nx.f;
                if (s_ > max_s) { max_s = s_; keep = seen; }
                else if (max_s - s_ > max_drop) break;
                if (!(i >= 0 && nx.t == 0)) break;
                cur = nx;
            } while (true);
            wave_sync();
            for (int64_t q = lane_id(); q < keep; q += 64) c[c[n_v0 + q].v].t = 1;   // used, whether or not the chain is then accepted
            wave_sync();
#else
        for (int64_t i = 0; i < n; ++i) c[i].t = 0;
        wave_sync();
        for (int64_t k = n_z - 1; k >= 0; --k) {
            const A128 zk = z[k];
            int64_t i = (int64_t)zk.y;
            if (c[i].t != 0) continue;
            const int32_t zx = (int32_t)zk.x;
            const int64_t n_v0 = n_v;
            int64_t seen = 0, keep = 0;
            int32_t max_s = 0;
            do {
                c[n_v0 + seen].v = (int32_t)i;
                ++seen;
                i = c[i].p;
                const int32_t s_ = i < 0 ? zx : zx - c[i].f;
                if (s_ > max_s) { max_s = s_; keep = seen; }
                else if (max_s - s_ > max_drop) break;
            } while (i >= 0 && c[i].t == 0);
            wave_sync();
            for (int64_t q = 0; q < keep; ++q) c[c[n_v0 + q].v].t = 1;   // used, whether or not the chain is then accepted
            wave_sync();
#endif
            n_v = n_v0 + keep;
            if (max_s >= min_sc && keep > 0 && keep >= min_cnt) {
                if (n_u < W.caps.max_reg * 4) u[n_u++] = (uint64_t)(uint32_t)max_s << 32 | (uint64_t)keep;
                else { W.status |= PMX_ST_OVERFLOW; n_v = n_v0; }
            } else n_v = n_v0;
        }
    }
    if (n_u == 0) { W.n_a = 0; W.n_u = 0; return; }

    PMX_STAMP(W, 19);
    // ---- compact (lchain.c:78-111): chains reversed into ascending order, then sorted by target position
    Ptr<A128> b = W.a2; PMX_LDS(b);
    int64_t kk = 0;
    for (int32_t i = 0; i < n_u; ++i) {
        const int64_t k0 = kk;
        const int32_t ni = (int32_t)u[i];
        for (int32_t j = 0; j < ni; ++j) b[kk++] = a[c[k0 + (ni - j - 1)].v];
    }
    Ptr<A128> wv = W.z; PMX_LDS(wv);   // z[] is free again
    kk = 0;
    for (int32_t i = 0; i < n_u; ++i) {
        wv[i].x = b[kk].x;
        wv[i].y = (uint64_t)kk << 32 | (uint32_t)i;
        kk += (int32_t)u[i];
    }
    radix_sort_128x(wv, wv + n_u, &W.status);
    Ptr<uint64_t> u2 = W.u2; PMX_LDS(u2);
    kk = 0;
    for (int32_t i = 0; i < n_u; ++i) {
        const int32_t j = (int32_t)wv[i].y, nn = (int32_t)u[j];
        u2[i] = u[j];
        const int64_t src = (int64_t)(wv[i].y >> 32);
        for (int32_t q = 0; q < nn; ++q) a[kk + q] = b[src + q];
        kk += nn;
    }
    for (int32_t i = 0; i < n_u; ++i) u[i] = u2[i];
    W.n_a = kk;
    W.n_u = n_u;
}

}  // namespace aln
}  // namespace pmx
