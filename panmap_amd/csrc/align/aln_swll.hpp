// ALIGN stage: the local alignment score minimap2 uses to look for inversions -- ksw_ll_qinit + ksw_ll_i16
// (ksw2_ll_sse.c:37-158), called by the z-drop test (align.c:74-86) and by mm_align1_inv (align.c:835-885).
//
// The reference is Farrar's striped Smith-Waterman on 16-bit lanes.  Its H values are the plain affine-gap local
// alignment scores (the lazy-F loop repairs the cells a vertical gap crosses between stripes; the E values it leaves
// uncorrected never change an H), so what is restated is the recurrence and the reference's CHOICE among equal
// maxima, which does depend on the striping:
//   * the query is padded to qlen8 = 8 * ceil(qlen / 8) positions, the pads score 0 against every base (they can carry
//     the maximum one column further, and then they are what the reference reports);
//   * te = the last target column whose maximum is >= every earlier column's (ksw2_ll_sse.c:148);
//   * qe = among the positions of that column holding the maximum, the last one in the striped memory order: position
//     p sits at index (p mod slen) * 8 + p / slen (slen = qlen8 / 8).
// Scores stay below 2^15 (q, t <= max_gap, a <= 2), so the reference's saturating 16-bit arithmetic never clips; the
// floor at zero of its unsigned subtractions is the local-alignment floor.
//
// Storage: three rotating H columns (previous, current, the column of the last maximum) and one E column of 16-bit
// values in W.H, idle between DP calls.  Wave kernels: 64 query positions per step, the vertical gap state F crosses
// the lanes as a prefix maximum of (H - gap open + position * gap extend).
#pragma once
#include "aln_types.hpp"

namespace pmx {
namespace aln {

PMX_HD int sw_simple_score(const Opt& o, uint32_t t, uint32_t q) {   // ksw_gen_simple_mat(5, a, b, sc_ambi) entry
    const int amb = o.sc_ambi > 0 ? -o.sc_ambi : o.sc_ambi;
    const int mis = o.b > 0 ? -o.b : o.b;
    const int mch = o.a < 0 ? -o.a : o.a;
    return (t > 3 || q > 3) ? amb : (t == q ? mch : mis);
}

// -> score; *qe, *te as ksw_ll_i16 leaves them.  qf(j), tf(i): base codes 0..4.  false in *ok: W.H is too small.
template <class QF, class TF>
PMX_HDN int sw_ll(Work& W, const Opt& o, int qlen, QF qf, int tlen, TF tf, int* qe, int* te, bool* ok) {
    PMX_LDS(&W);
    *qe = *te = -1;
    *ok = true;
    const int slen = (qlen + 7) / 8, qlen8 = slen * 8;
    const int stride = qlen8 + 8;
    if ((size_t)stride * 4 * sizeof(uint16_t) > (size_t)(W.caps.max_tlen + 32) * sizeof(int32_t)) { *ok = false; return 0; }
    Ptr<uint16_t> buf = ptr_cast<uint16_t>(W.H); PMX_LDS(buf);
    const int gapoe = o.q + o.e, gape = o.e;
    int gmax = 0, t_end = -1;
    int prev = 0, keep = 2;   // column buffers 0..2; E is buffer 3
#if PMX_W > 1
    const int lane = lane_id();
    for (int j = lane; j < qlen8; j += PMX_W) { buf[(size_t)prev * stride + j] = 0; buf[(size_t)keep * stride + j] = 0; buf[(size_t)3 * stride + j] = 0; }
    wave_sync();
    for (int i = 0; i < tlen; ++i) {
        int cur = 0;
        while (cur == prev || cur == keep) ++cur;
        const uint32_t tb = (uint32_t)tf(i);
        int carry = 0, imax = 0;   // F entering the block's first row
        for (int j0 = 0; j0 < qlen8; j0 += PMX_W) {
            const int j = j0 + lane;
            const bool act = j < qlen8;
            int hd = 0, e = 0, s = 0;
            if (act) {
                hd = j > 0 ? (int)buf[(size_t)prev * stride + j - 1] : 0;
                e = (int)buf[(size_t)3 * stride + j];
                s = j < qlen ? sw_simple_score(o, tb, (uint32_t)qf(j)) : 0;
            }
            int h0 = hd + s;
            h0 = h0 > e ? h0 : e;
            // F of row l in this block = max(carry, max over l' < l of (h0[l'] - gapoe + (l' + 1) * gape)) - l * gape
            const int v = act ? h0 - gapoe + (lane + 1) * gape : INT32_MIN / 2;
            int x = v;
            for (int d = 1; d < PMX_W; d <<= 1) {
                const int y = __shfl_up(x, d);
                if (lane >= d) x = y > x ? y : x;
            }
            int ex = __shfl_up(x, 1);
            if (lane == 0) ex = INT32_MIN / 2;
            ex = ex > carry ? ex : carry;
            const int f = ex - lane * gape;
            int h = h0 > f ? h0 : f;
            h = h > 0 ? h : 0;
            int en = e - gape;
            en = en > h - gapoe ? en : h - gapoe;
            en = en > 0 ? en : 0;
            if (act) {
                buf[(size_t)cur * stride + j] = (uint16_t)h;
                buf[(size_t)3 * stride + j] = (uint16_t)en;
            }
            int incl = __shfl(x, PMX_W - 1);
            incl = incl > carry ? incl : carry;
            carry = incl - PMX_W * gape;
            carry = carry > 0 ? carry : 0;
            int m = act ? h : 0;
            for (int d = PMX_W / 2; d > 0; d >>= 1) { const int y = __shfl_xor(m, d); m = y > m ? y : m; }
            imax = m > imax ? m : imax;
        }
        wave_sync();
        if (imax >= gmax) { gmax = imax; t_end = i; keep = cur; }
        prev = cur;
    }
    // the last position in striped order that holds the maximum
    int best = -1;
    for (int p = lane; p < qlen8; p += PMX_W)
        if ((int)buf[(size_t)keep * stride + p] == gmax) {
            const int idx = (p % slen) * 8 + p / slen;
            if (idx > best) best = idx;
        }
    for (int d = PMX_W / 2; d > 0; d >>= 1) { const int y = __shfl_xor(best, d); best = y > best ? y : best; }
    wave_sync();
    if (best >= 0) *qe = best / 8 + best % 8 * slen;
#else
    for (int j = 0; j < qlen8; ++j) { buf[(size_t)prev * stride + j] = 0; buf[(size_t)keep * stride + j] = 0; buf[(size_t)3 * stride + j] = 0; }
    for (int i = 0; i < tlen; ++i) {
        int cur = 0;
        while (cur == prev || cur == keep) ++cur;
        const uint32_t tb = (uint32_t)tf(i);
        int f = 0, imax = 0, hd = 0;
        for (int j = 0; j < qlen8; ++j) {
            const int e = (int)buf[(size_t)3 * stride + j];
            const int s = j < qlen ? sw_simple_score(o, tb, (uint32_t)qf(j)) : 0;
            int h = hd + s;
            h = h > e ? h : e;
            h = h > f ? h : f;
            h = h > 0 ? h : 0;
            hd = (int)buf[(size_t)prev * stride + j];
            buf[(size_t)cur * stride + j] = (uint16_t)h;
            int en = e - gape;
            en = en > h - gapoe ? en : h - gapoe;
            buf[(size_t)3 * stride + j] = (uint16_t)(en > 0 ? en : 0);
            f -= gape;
            f = f > h - gapoe ? f : h - gapoe;
            f = f > 0 ? f : 0;
            imax = h > imax ? h : imax;
        }
        if (imax >= gmax) { gmax = imax; t_end = i; keep = cur; }
        prev = cur;
    }
    int best = -1;
    for (int p = 0; p < qlen8; ++p)
        if ((int)buf[(size_t)keep * stride + p] == gmax) {
            const int idx = (p % slen) * 8 + p / slen;
            if (idx > best) best = idx;
        }
    if (best >= 0) *qe = best / 8 + best % 8 * slen;
#endif
    *te = t_end;
    return gmax;
}

}  // namespace aln
}  // namespace pmx
