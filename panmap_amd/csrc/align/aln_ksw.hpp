// ALIGN stage, part 4: dual-affine-gap extension / global alignment in the Suzuki-Kasahara difference
// formulation, one anti-diagonal at a time, cells of a diagonal spread over the lanes of the wave.
//
// Restates ksw_extd2_sse (ksw2_extd2_sse.c:27-401) including everything that is observable in its
// results: wrapping int8 difference arithmetic, the 16-cell rounding of the computed range [st,en]
// (cells outside the true band are computed from whatever the arrays hold, exactly like the SSE lanes
// do), boundary injection, the exact-max H[] tracking with its 4-way tie order, approximate max, z-drop,
// end bonus, and ksw_backtrack (ksw2.h:127-162).  Integer DP: VALU + LDS, no MFMA.
#pragma once
#include "aln_types.hpp"

namespace pmx {
namespace aln {

PMX_HD void ez_reset(Ez& ez) {   // ksw_reset_extz (ksw2.h:164-169)
    ez.max_q = ez.max_t = ez.mqe_t = ez.mte_q = -1;
    ez.max = 0;
    ez.score = ez.mqe = ez.mte = PMX_KSW_NEG_INF;
    ez.n_cigar = 0;
    ez.zdropped = 0;
    ez.reach_end = 0;
}

// ksw_apply_zdrop (ksw2.h:171-188), is_rot == 1
PMX_HD int ez_apply_zdrop(Ez& ez, int32_t H, int r, int t, int zdrop, int8_t e) {
    if (H > (int32_t)ez.max) {
        ez.max = (uint32_t)H;
        ez.max_t = t;
        ez.max_q = r - t;
    } else if (t >= ez.max_t && r - t >= ez.max_q) {
        const int tl = t - ez.max_t, ql = (r - t) - ez.max_q;
        const int l = tl > ql ? tl - ql : ql - tl;
        if (zdrop >= 0 && (int32_t)ez.max - H > zdrop + l * e) {
            ez.zdropped = 1;
            return 1;
        }
    }
    return 0;
}

PMX_HD void push_cigar(Ptr<uint32_t> cigar, int* n_cigar, int cap, uint32_t op, int len, uint32_t* status) {   // ksw_push_cigar
    if (*n_cigar == 0 || op != (cigar[*n_cigar - 1] & 0xf)) {
        if (*n_cigar < cap) cigar[(*n_cigar)++] = (uint32_t)len << 4 | op;
        else *status |= PMX_ST_OVERFLOW;
    } else cigar[*n_cigar - 1] += (uint32_t)len << 4;
}

// ksw_backtrack (ksw2.h:127-162) with is_rot = 1, min_intron_len = 0
PMX_HDN void ksw_backtrack(Work& W, int is_rev, Ptr<const uint8_t> p, Ptr<const int32_t> off, Ptr<const int32_t> off_end, int n_col, int i0, int j0,
                          int* n_cigar_) {
    PMX_LDS(&W); PMX_LDS(off); PMX_LDS(off_end);   // p = traceback matrix: global memory
    int n_cigar = 0, i = i0, j = j0, state = 0;
    Ptr<uint32_t> cigar = W.cig_tmp; PMX_LDS(cigar);
    const int cap = W.caps.max_cigar;
    while (i >= 0 && j >= 0) {
        int force_state = -1;
        const int r = i + j;
        if (i < off[r]) force_state = 2;
        if (i > off_end[r]) force_state = 1;
        const uint32_t tmp = force_state < 0 ? p[(size_t)r * n_col + i - off[r]] : 0;
        if (state == 0) state = tmp & 7;
        else if (!(tmp >> (state + 2) & 1)) state = 0;
        if (state == 0) state = tmp & 7;
        if (force_state >= 0) state = force_state;
        if (state == 0) { push_cigar(cigar, &n_cigar, cap, 0, 1, &W.status); --i; --j; }
        else if (state == 1 || state == 3) { push_cigar(cigar, &n_cigar, cap, 2, 1, &W.status); --i; }
        else { push_cigar(cigar, &n_cigar, cap, 1, 1, &W.status); --j; }
    }
    if (i >= 0) push_cigar(cigar, &n_cigar, cap, 2, i + 1, &W.status);
    if (j >= 0) push_cigar(cigar, &n_cigar, cap, 1, j + 1, &W.status);
    if (!is_rev)
        for (i = 0; i < n_cigar >> 1; ++i) { const uint32_t t = cigar[i]; cigar[i] = cigar[n_cigar - 1 - i]; cigar[n_cigar - 1 - i] = t; }
    *n_cigar_ = n_cigar;
}

#if PMX_W > 1
// max over the wave of a 64-bit key
__device__ __forceinline__ int64_t wave_max_i64(int64_t v) {
    for (int o = 32; o > 0; o >>= 1) {
        const int64_t other = __shfl_xor(v, o);
        v = other > v ? other : v;
    }
    return v;
}
#else
PMX_HD int64_t wave_max_i64(int64_t v) { return v; }
#endif

// ksw_extd2_sse.  query/target hold nt4 codes; with_cigar always on.  Results in ez and W.cig_tmp.
// FAST (wave-per-read kernels, long reads): the arrays are the small LDS copy (Work::dp_fast) and the compiler is told
// so -- through generic pointers every access is a flat instruction, and at ~1.2 M of them per 10 kb read the kernel was
// bound by the rate the texture-address unit takes them (16 clocks each), not by their latency.
// RIGHT = (flag & PMX_EZ_RIGHT) as a compile-time constant: with the gap-alignment rule chosen by a (uniform) run-time test
// the compiler issues BOTH variants of the two 10-instruction blocks under execution masks, a quarter of the inner loop.
template <bool FAST, bool RIGHT, class QP, class TP>
PMX_HDN void ksw_extd2_t(Work& W, int qlen, QP query, int tlen, TP target, const int8_t* mat, int8_t q, int8_t e,
                        int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, Ez& ez) {
    PMX_LDS(&W); PMX_LDS(query); PMX_LDS(target);
    const int lane = lane_id();
    const int approx_max = !!(flag & PMX_EZ_APPROX_MAX);
    ez_reset(ez);
    if (qlen <= 0 || tlen <= 0) return;
    if (q2 + e2 < q + e) { int8_t t_ = q; q = q2; q2 = t_; t_ = e; e = e2; e2 = t_; }
    const int qe = q + e;
    const int8_t sc_mch = mat[0], sc_mis = mat[1], sc_N = mat[24] == 0 ? (int8_t)-e2 : mat[24];
    if (w < 0) w = tlen > qlen ? tlen : qlen;
    const int wl = w, wr = w;
    const int tlen_ = (tlen + 15) / 16;
    int n_col_ = qlen < tlen ? qlen : tlen;
    n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
    const int n_col = n_col_ * 16;
    {   // "otherwise, we won't see any mismatches" (ksw2_extd2_sse.c:100)
        int min_sc = mat[1];
        for (int t = 1; t < 25; ++t) min_sc = min_sc < mat[t] ? min_sc : mat[t];
        if (-min_sc > 2 * (q + e)) return;
    }
    if (tlen_ * 16 > W.caps.max_tlen || qlen > W.caps.max_tlen || (size_t)(qlen + tlen - 1) * (size_t)n_col > W.tb_cap) {
        W.status |= PMX_ST_OVERFLOW;
        ez.zdropped = 1;
        return;
    }
    int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
    if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
    const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;

    Ptr<int8_t> u = W.du, v = W.dv, x = W.dx, y = W.dy, x2 = W.dx2, y2 = W.dy2, s = W.ds;
    Ptr<uint8_t> sf = W.sf, qr = W.qr;
    Ptr<int32_t> H = W.H;
    Ptr<int32_t> off = W.off, off_end = W.off_end;
    PMX_LDS(u); PMX_LDS(v); PMX_LDS(x); PMX_LDS(y); PMX_LDS(x2); PMX_LDS(y2); PMX_LDS(s);
    PMX_LDS(sf); PMX_LDS(qr); PMX_LDS(H); PMX_LDS(off); PMX_LDS(off_end);
    Ptr<uint8_t> p = W.tb;
    const bool tb_lds = FAST ? false : PMX_TB_IS_LDS(p);   // uniform (FAST: the traceback of a long read's DP is in its HBM slab)
    (void)tb_lds;
    const int T16 = tlen_ * 16;
#if !defined(PMX_INTERLEAVED)
    if (FAST) {   // small DP of a long read: the LDS copy of the arrays
        constexpr int TF = PMX_DP_FAST_TLEN + 32;
        int8_t* d = W.dp_fast;
        PMX_LDS_HERE(d);
        u = d; v = d + TF; x = d + 2 * TF; y = d + 3 * TF; x2 = d + 4 * TF; y2 = d + 5 * TF; s = d + 6 * TF;
        sf = (uint8_t*)(d + 7 * TF);
        qr = (uint8_t*)(d + 8 * TF);   // TF + 64 bytes
        PMX_LDS_HERE(u); PMX_LDS_HERE(v); PMX_LDS_HERE(x); PMX_LDS_HERE(y); PMX_LDS_HERE(x2); PMX_LDS_HERE(y2); PMX_LDS_HERE(s);
        PMX_LDS_HERE(sf); PMX_LDS_HERE(qr);
    }
#endif
    // initial fill (ksw2_extd2_sse.c:107-126): every lane takes a stride
    for (int t = lane; t < T16; t += PMX_W) {
        u[t] = v[t] = x[t] = y[t] = (int8_t)(-q - e);
        x2[t] = y2[t] = (int8_t)(-q2 - e2);
        if (!approx_max) H[t] = PMX_KSW_NEG_INF;
    }
    for (int t = lane; t < T16 + 16; t += PMX_W) {
        s[t] = 0;
        sf[t] = t < tlen ? target[t] : 0;
    }
    const int Q16 = (qlen + 15) / 16 * 16;
    for (int t = lane; t < Q16 + 32; t += PMX_W) qr[t] = t < qlen ? query[qlen - 1 - t] : 0;
    wave_sync();

    int last_st = -1, last_en = -1;
    int32_t H0 = 0, last_H0_t = 0;
    for (int r = 0; r < qlen + tlen - 1; ++r) {
        int st = 0, en = tlen - 1;
        if (st < r - qlen + 1) st = r - qlen + 1;
        if (en > r) en = r;
        if (st < (r - wr + 1) >> 1) st = (r - wr + 1) >> 1;
        if (en > (r + wl) >> 1) en = (r + wl) >> 1;
        if (st > en) { ez.zdropped = 1; break; }
        const int st0 = st, en0 = en;
        st = st / 16 * 16;
        en = (en + 16) / 16 * 16 - 1;
        // boundary conditions (ksw2_extd2_sse.c:150-166)
        int8_t x1, x21, v1;
        if (st > 0) {
            if (st - 1 >= last_st && st - 1 <= last_en) { x1 = x[st - 1]; x21 = x2[st - 1]; v1 = v[st - 1]; }
            else { x1 = (int8_t)(-q - e); x21 = (int8_t)(-q2 - e2); v1 = (int8_t)(-q - e); }
        } else {
            x1 = (int8_t)(-q - e);
            x21 = (int8_t)(-q2 - e2);
            v1 = r == 0 ? (int8_t)(-q - e) : r < long_thres ? (int8_t)-e : r == long_thres ? (int8_t)long_diff : (int8_t)-e2;
        }
        wave_sync();   // x1/x21/v1 were read before anybody overwrites; boundary stores below
        if (en >= r) {
            y[r] = (int8_t)(-q - e);
            y2[r] = (int8_t)(-q2 - e2);
            u[r] = r == 0 ? (int8_t)(-q - e) : r < long_thres ? (int8_t)-e : r == long_thres ? (int8_t)long_diff : (int8_t)-e2;
        }
        // scores (loop fission, :168-189): 16-wide chunks starting at st0
        Ptr<const uint8_t> qrr = qr + (qlen - 1 - r);
        const int s_end = st0 + ((en0 - st0) / 16 + 1) * 16;
        for (int t = st0 + lane; t < s_end; t += PMX_W) {
            const uint8_t sq = sf[t], sq2 = qrr[t];
            int8_t val = sq == sq2 ? sc_mch : sc_mis;
            if (sq == 4 || sq2 == 4) val = sc_N;
            s[t] = val;
        }
        wave_sync();
        // core recurrence on [st, en], highest chunk first so that every chunk still sees the previous
        // diagonal in x[t-1], v[t-1], x2[t-1]
        off[r] = st;
        off_end[r] = en;
        const uint32_t pr_off = (uint32_t)(r * n_col - st);
        const int n_chunk = (en - st + PMX_W) / PMX_W;
        for (int c = n_chunk - 1; c >= 0; --c) {
            const int t = st + c * PMX_W + lane;
            const bool act = t <= en;
            int8_t z = 0, a = 0, b = 0, a2 = 0, b2 = 0, vt1 = 0, ut = 0;
            if (act) {
                z = s[t];
                int8_t xt1, x2t1;
                if (FAST) {   // the element before an array's first is the last of the array in front of it: loads first, selects after
                    const int8_t xm = x[t - 1], vm = v[t - 1], x2m = x2[t - 1];
                    xt1 = t == st ? x1 : xm;
                    vt1 = t == st ? v1 : vm;
                    x2t1 = t == st ? x21 : x2m;
                } else {
                    xt1 = t == st ? x1 : x[t - 1];
                    vt1 = t == st ? v1 : v[t - 1];
                    x2t1 = t == st ? x21 : x2[t - 1];
                }
                ut = u[t];
                a = (int8_t)(xt1 + vt1);
                b = (int8_t)(y[t] + ut);
                a2 = (int8_t)(x2t1 + vt1);
                b2 = (int8_t)(y2[t] + ut);
            }
            wave_sync();   // all loads of the old diagonal done before any store
            if (act) {
                uint8_t d;
                if (!RIGHT) {   // gap left-alignment (:228-268)
                    d = a > z ? 1 : 0;
                    z = z > a ? z : a;
                    d = b > z ? 2 : d;
                    z = z > b ? z : b;
                    d = a2 > z ? 3 : d;
                    z = z > a2 ? z : a2;
                    d = b2 > z ? 4 : d;
                    z = z > b2 ? z : b2;
                    z = z < sc_mch ? z : sc_mch;
                } else {                        // gap right-alignment (:269-321)
                    d = z > a ? 0 : 1;
                    z = z > a ? z : a;
                    d = z > b ? d : 2;
                    z = z > b ? z : b;
                    d = z > a2 ? d : 3;
                    z = z > a2 ? z : a2;
                    d = z > b2 ? d : 4;
                    z = z > b2 ? z : b2;
                    z = z < sc_mch ? z : sc_mch;
                }
                u[t] = (int8_t)(z - vt1);
                v[t] = (int8_t)(z - ut);
                int8_t tmp = (int8_t)(z - q);
                a = (int8_t)(a - tmp);
                b = (int8_t)(b - tmp);
                tmp = (int8_t)(z - q2);
                a2 = (int8_t)(a2 - tmp);
                b2 = (int8_t)(b2 - tmp);
                if (!RIGHT) {
                    x[t] = (int8_t)((a > 0 ? a : 0) - qe);
                    d |= a > 0 ? 0x08 : 0;
                    y[t] = (int8_t)((b > 0 ? b : 0) - qe);
                    d |= b > 0 ? 0x10 : 0;
                    x2[t] = (int8_t)((a2 > 0 ? a2 : 0) - (q2 + e2));
                    d |= a2 > 0 ? 0x20 : 0;
                    y2[t] = (int8_t)((b2 > 0 ? b2 : 0) - (q2 + e2));
                    d |= b2 > 0 ? 0x40 : 0;
                } else {
                    x[t] = (int8_t)((0 > a ? 0 : a) - qe);
                    d |= 0 > a ? 0 : 0x08;
                    y[t] = (int8_t)((0 > b ? 0 : b) - qe);
                    d |= 0 > b ? 0 : 0x10;
                    x2[t] = (int8_t)((0 > a2 ? 0 : a2) - (q2 + e2));
                    d |= 0 > a2 ? 0 : 0x20;
                    y2[t] = (int8_t)((0 > b2 ? 0 : b2) - (q2 + e2));
                    d |= 0 > b2 ? 0 : 0x40;
                }
                PMX_TB_STORE(p, tb_lds, pr_off + (uint32_t)t, d);
            }
            wave_sync();
        }
        if (!approx_max) {   // exact max through the int32 H[] band (:323-366)
            int32_t max_H, max_t;
            if (r > 0) {
                const int32_t h_en0 = en0 > 0 ? H[en0 - 1] + u[en0] : H[en0] + v[en0];
                wave_sync();
                // candidate order among equal H: en0 first, then the four t-classes of the vector loop
                // (smaller class, then smaller t), then the scalar tail in ascending t
                const int en1 = st0 + (en0 - st0) / 4 * 4;
                int64_t best = (int64_t)((uint64_t)(uint32_t)h_en0 << 32 | 0x7fffffffu);
                for (int t = st0 + lane; t < en0; t += PMX_W) {
                    const int32_t h = H[t] + (int32_t)v[t];
                    H[t] = h;
                    uint32_t prio;   // larger = preferred
                    if (t < en1) prio = 0x7ffffffeu - ((uint32_t)((t - st0) & 3) << 24) - (uint32_t)((t - st0) >> 2);
                    else prio = 0x7ffffffeu - (5u << 24) - (uint32_t)(t - st0);
                    const int64_t key = (int64_t)((uint64_t)(uint32_t)h << 32 | prio);
                    best = key > best ? key : best;
                }
                if (lane == 0 || PMX_W == 1) H[en0] = h_en0;
                best = wave_max_i64(best);
                wave_sync();
                max_H = (int32_t)(best >> 32);
                const uint32_t prio = (uint32_t)best;
                if (prio == 0x7fffffffu) max_t = en0;
                else {
                    const uint32_t d = 0x7ffffffeu - prio;
                    const uint32_t cls = d >> 24, idx = d & 0xffffffu;
                    max_t = cls >= 5 ? st0 + (int)idx : st0 + (int)(idx * 4 + cls);
                }
            } else {
                if (lane == 0 || PMX_W == 1) H[0] = v[0] - qe;
                wave_sync();
                max_H = H[0];
                max_t = 0;
            }
            if (en0 == tlen - 1 && H[en0] > ez.mte) { ez.mte = H[en0]; ez.mte_q = r - en0; }
            if (r - st0 == qlen - 1 && H[st0] > ez.mqe) { ez.mqe = H[st0]; ez.mqe_t = st0; }
            if (ez_apply_zdrop(ez, max_H, r, max_t, zdrop, e2)) break;
            if (r == qlen + tlen - 2 && en0 == tlen - 1) ez.score = H[tlen - 1];
        } else {             // approximate max along one path (:367-383)
            if (r > 0) {
                if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
                    const int32_t d0 = v[last_H0_t], d1 = u[last_H0_t + 1];
                    if (d0 > d1) H0 += d0;
                    else { H0 += d1; ++last_H0_t; }
                } else if (last_H0_t >= st0 && last_H0_t <= en0) {
                    H0 += v[last_H0_t];
                } else {
                    ++last_H0_t;
                    H0 += u[last_H0_t];
                }
            } else { H0 = v[0] - qe; last_H0_t = 0; }
            if ((flag & PMX_EZ_APPROX_DROP) && ez_apply_zdrop(ez, H0, r, last_H0_t, zdrop, e2)) break;
            if (r == qlen + tlen - 2 && en0 == tlen - 1) ez.score = H0;
        }
        last_st = st;
        last_en = en;
    }
    wave_sync();
    // backtrack (:388-399)
    const int rev_cigar = !!(flag & PMX_EZ_REV_CIGAR);
    if (!ez.zdropped && !(flag & PMX_EZ_EXTZ_ONLY)) {
        ksw_backtrack(W, rev_cigar, p, off, off_end, n_col, tlen - 1, qlen - 1, &ez.n_cigar);
    } else if (!ez.zdropped && (flag & PMX_EZ_EXTZ_ONLY) && ez.mqe + end_bonus > (int)ez.max) {
        ez.reach_end = 1;
        ksw_backtrack(W, rev_cigar, p, off, off_end, n_col, ez.mqe_t, qlen - 1, &ez.n_cigar);
    } else if (ez.max_t >= 0 && ez.max_q >= 0) {
        ksw_backtrack(W, rev_cigar, p, off, off_end, n_col, ez.max_t, ez.max_q, &ez.n_cigar);
    }
    wave_sync();
}


}  // namespace aln
}  // namespace pmx
#include "aln_ksw_rows.hpp"
namespace pmx {
namespace aln {

template <class QP, class TP>
PMX_HD void ksw_extd2(Work& W, int qlen, QP query, int tlen, TP target, const int8_t* mat, int8_t q, int8_t e,
                      int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, Ez& ez) {
#if !defined(PMX_INTERLEAVED) && !defined(PMX_ALL_LDS)
#if PMX_W == 64 && defined(__HIP_DEVICE_COMPILE__)
    // long reads: row by row with the columns in registers (aln_ksw_rows.hpp) whenever the band never cuts the matrix
    const unsigned long long pt0 = W.prof && W.dp_fast ? (unsigned long long)clock64() : 0ULL;   // diagnostic (long reads): which kernel ran the DP
    auto account = [&](int slot) {
        if (W.prof && W.dp_fast && lane_id() == 0) {
            atomicAdd(&W.prof[slot], 1ULL);
            atomicAdd(&W.prof[slot + 1], (unsigned long long)qlen * (unsigned long long)tlen);
            atomicAdd(&W.prof[slot + 2], (unsigned long long)clock64() - pt0);
        }
    };
    if (W.dp_fast && W.caps.dp_fast_tlen == PMX_DP_FAST_TLEN && !W.no_rows_dp) {
        // (the LDS copy of the DP arrays, 9 x 640 + 64 bytes: window + query first, the replay arrays behind them)
        const size_t area = (size_t)9 * (PMX_DP_FAST_TLEN + 32) + 64, main_b = ksw_rows_lds_main(qlen);
        if (main_b <= area &&
            ksw_extd2_rows(W, W.dp_fast, main_b, W.dp_fast + main_b, area - main_b, qlen, query, tlen, target, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, ez)) {
            account(23);
            return;
        }
    }
#else
    auto account = [&](int) {};
#endif
    if (W.dp_fast && W.caps.dp_fast_tlen == PMX_DP_FAST_TLEN && (tlen + 15) / 16 * 16 <= PMX_DP_FAST_TLEN && qlen <= PMX_DP_FAST_TLEN) {
        if (flag & PMX_EZ_RIGHT) ksw_extd2_t<true, true>(W, qlen, query, tlen, target, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, ez);
        else ksw_extd2_t<true, false>(W, qlen, query, tlen, target, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, ez);
        account(26);
        return;
    }
    if (flag & PMX_EZ_RIGHT) ksw_extd2_t<false, true>(W, qlen, query, tlen, target, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, ez);
    else ksw_extd2_t<false, false>(W, qlen, query, tlen, target, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, ez);
    account(29);
#else
#if PMX_W == 64 && defined(__HIP_DEVICE_COMPILE__) && defined(PMX_ALL_LDS)
    // wave-per-pair tier, all-LDS layout: the row-by-row kernel on the DP arrays' own LDS (du | sf | qr are contiguous: window +
    // query copy; H | off_ are: the replay arrays; tseq between them holds the target and is not touched)
    if (!W.no_rows_dp) {
        const size_t main_b = (size_t)7 * (W.caps.max_tlen + 32) + (size_t)(W.caps.max_tlen + 32) + (size_t)(W.caps.max_tlen + 64);
        const size_t arr_b = (size_t)4 * (W.caps.max_tlen + 32) + (size_t)8 * (W.caps.max_qlen + W.caps.max_tlen);
        if (((W.caps.max_tlen + 32) & 15) == 0 &&   // (the arrays are packed back to back only when their sizes are multiples of 16)
            ksw_extd2_rows(W, (int8_t*)W.du, main_b, (int8_t*)W.H, arr_b, qlen, query, tlen, target, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, ez))
            return;
    }
#endif
    if (flag & PMX_EZ_RIGHT) ksw_extd2_t<false, true>(W, qlen, query, tlen, target, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, ez);
    else ksw_extd2_t<false, false>(W, qlen, query, tlen, target, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, ez);
#endif
}

#if PMX_W == 64
// ------------------------------------------------------------------------------------------------------
// ksw_extd2 for short targets, REGISTER-resident: lane L of the wave owns target columns t = L + 64*c
// (c < NC), so the seven int8 difference arrays and H[] of ksw2_extd2_sse.c are per-lane registers, the
// t-1 neighbours of the recurrence are one lane away (shuffle of the previous diagonal's values) and the
// scalar reads the reference makes on its arrays (x[st-1], H[en0-1], u[en0], v[last_H0_t], ...) are
// v_readlane at a uniform index.  No LDS round trips or barriers inside the anti-diagonal loop except the
// traceback byte each active lane stores.  Semantics are those of ksw_extd2 above, statement for statement
// (same 16-cell rounding of [st,en], same stale s[] cells, same candidate order of the exact max).
// Requires ((tlen + 15) / 16) * 16 <= 64 * NC.
__device__ __forceinline__ int rl_i32(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
// lane L receives lane L-1's value (lane 0: 0): DPP wave_shr:1, one VALU op, no LDS crossbar round trip
__device__ __forceinline__ int wave_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false); }
// wave-wide signed max through DPP (row_shr 1/2/4/8 scan inside each row of 16, row_bcast:15 / :31 across rows;
// the total lands in lane 63): six VALU ops instead of six ds_bpermute round trips through the LDS crossbar
__device__ __forceinline__ int wave_max_i32x(int v) {
    const int lo = INT32_MIN;
#define PMX_DPP_MAX(ctrl, rmask)                                                        \
    do {                                                                                \
        const int o_ = __builtin_amdgcn_update_dpp(lo, v, (ctrl), (rmask), 0xf, false); \
        v = o_ > v ? o_ : v;                                                            \
    } while (0)
    PMX_DPP_MAX(0x111, 0xf);
    PMX_DPP_MAX(0x112, 0xf);
    PMX_DPP_MAX(0x114, 0xf);
    PMX_DPP_MAX(0x118, 0xf);
    PMX_DPP_MAX(0x142, 0xa);
    PMX_DPP_MAX(0x143, 0xc);
#undef PMX_DPP_MAX
    return __builtin_amdgcn_readlane(v, 63);
}

template <int NC>
__device__ __forceinline__ int rd_col(const int (&a)[NC], int idx) {   // a[] indexed by target column, uniform idx
    if (NC == 1) return rl_i32(a[0], idx);
    if (NC == 2) return idx < 64 ? rl_i32(a[0], idx) : rl_i32(a[NC - 1], idx - 64);
    return idx < 64 ? rl_i32(a[0], idx) : idx < 128 ? rl_i32(a[NC > 1 ? 1 : 0], idx - 64) : rl_i32(a[NC - 1], idx - 128);
}

template <int NC, bool TB_IN_LDS, bool RIGHT>
__device__ void ksw_extd2_reg_t(Work& W, int qlen, const uint8_t* query, int tlen, const uint8_t* target, const int8_t* mat, int8_t q, int8_t e,
                                int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, Ez& ez) {
    static_assert(NC >= 1 && NC <= 3, "one to three target columns per lane");
    PMX_LDS(&W); PMX_LDS(query); PMX_LDS(target);
    const int lane = lane_id();
    const int approx_max = !!(flag & PMX_EZ_APPROX_MAX);
    ez_reset(ez);
    if (qlen <= 0 || tlen <= 0) return;
    if (q2 + e2 < q + e) { int8_t t_ = q; q = q2; q2 = t_; t_ = e; e = e2; e2 = t_; }
    const int qe = q + e;
    const int8_t sc_mch = mat[0], sc_mis = mat[1], sc_N = mat[24] == 0 ? (int8_t)-e2 : mat[24];
    if (w < 0) w = tlen > qlen ? tlen : qlen;
    const int wl = w, wr = w;
    const int tlen_ = (tlen + 15) / 16;
    int n_col_ = qlen < tlen ? qlen : tlen;
    n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
    const int n_col = n_col_ * 16;
    {
        int min_sc = mat[1];
        for (int t = 1; t < 25; ++t) min_sc = min_sc < mat[t] ? min_sc : mat[t];
        if (-min_sc > 2 * (q + e)) return;
    }
    if (tlen_ * 16 > 64 * NC || qlen + tlen > W.caps.max_qlen + W.caps.max_tlen || (size_t)(qlen + tlen - 1) * (size_t)n_col > W.tb_cap) {
        W.status |= PMX_ST_OVERFLOW;
        ez.zdropped = 1;
        return;
    }
    int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
    if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
    const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
    int32_t *off = W.off, *off_end = W.off_end;
    PMX_LDS(off); PMX_LDS(off_end);
    uint8_t* p = W.tb;

    // per-column state (int8 values of the reference kept sign-extended in 32-bit registers)
    int u[NC], v[NC], x[NC], y[NC], x2[NC], y2[NC], s[NC], H[NC], sf[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int t = c * 64 + lane;
        u[c] = v[c] = x[c] = y[c] = (int8_t)(-q - e);
        x2[c] = y2[c] = (int8_t)(-q2 - e2);
        H[c] = PMX_KSW_NEG_INF;
        s[c] = 0;
        sf[c] = t < tlen ? target[t] : 0;
    }
    const int8_t init_ue = (int8_t)(-q - e), init_ue2 = (int8_t)(-q2 - e2);

    int last_st = -1, last_en = -1;
    int32_t H0 = 0, last_H0_t = 0;
    for (int r = 0; r < qlen + tlen - 1; ++r) {
        int st = 0, en = tlen - 1;
        if (st < r - qlen + 1) st = r - qlen + 1;
        if (en > r) en = r;
        if (st < (r - wr + 1) >> 1) st = (r - wr + 1) >> 1;
        if (en > (r + wl) >> 1) en = (r + wl) >> 1;
        if (st > en) { ez.zdropped = 1; break; }
        const int st0 = st, en0 = en;
        st = st / 16 * 16;
        en = (en + 16) / 16 * 16 - 1;
        // this diagonal's query bases, issued first so the LDS latency hides behind the boundary work
        // (for NC > 1 every per-column section skips the columns that lie outside its range with a uniform branch:
        //  a diagonal of a 150 x 150 problem touches one or two of the three columns)
        const int s_end = st0 + ((en0 - st0) / 16 + 1) * 16;
        int sq2v[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            sq2v[c] = 0;
            if (NC > 1 && (c * 64 + 63 < st0 || c * 64 >= s_end)) continue;
            const int t = c * 64 + lane;
            sq2v[c] = (t <= r && r - t < qlen) ? (int)query[r - t] : 0;
        }
        // boundary conditions (ksw2_extd2_sse.c:150-166)
        const int8_t gap_head = r == 0 ? init_ue : r < long_thres ? (int8_t)-e : r == long_thres ? (int8_t)long_diff : (int8_t)-e2;
        int8_t x1, x21, v1;
        if (st > 0) {
            if (st - 1 >= last_st && st - 1 <= last_en) {
                x1 = (int8_t)rd_col<NC>(x, st - 1); x21 = (int8_t)rd_col<NC>(x2, st - 1); v1 = (int8_t)rd_col<NC>(v, st - 1);
            } else { x1 = init_ue; x21 = init_ue2; v1 = init_ue; }
        } else { x1 = init_ue; x21 = init_ue2; v1 = gap_head; }
        if (en >= r) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (c * 64 + lane == r) { y[c] = init_ue; y2[c] = init_ue2; u[c] = gap_head; }
        }
        // scores of [st0, s_end); cells of [st, st0) keep whatever s[] held (as the SSE code does)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (NC > 1 && (c * 64 + 63 < st0 || c * 64 >= s_end)) continue;
            const int t = c * 64 + lane;
            if (t >= st0 && t < s_end) {
                const int sq = sf[c];
                const int sq2 = t <= r ? sq2v[c] : 0;   // qr[] padding is 0 beyond the query
                int8_t val = sq == sq2 ? sc_mch : sc_mis;
                if (sq == 4 || sq2 == 4) val = sc_N;
                s[c] = val;
            }
        }
        // the t-1 neighbours, read from the previous diagonal's state before anything is updated
        int xs[NC], vs[NC], x2s[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            xs[c] = vs[c] = x2s[c] = 0;
            if (NC > 1 && (c * 64 + 63 < st || c * 64 > en)) continue;
            xs[c] = wave_shr1(x[c]);
            vs[c] = wave_shr1(v[c]);
            x2s[c] = wave_shr1(x2[c]);
            if (c > 0) {   // column c's lane 0 continues after column c-1's lane 63
                const int bx = rl_i32(x[c - 1], 63), bv = rl_i32(v[c - 1], 63), bx2 = rl_i32(x2[c - 1], 63);
                if (lane == 0) { xs[c] = bx; vs[c] = bv; x2s[c] = bx2; }
            }
        }
        // core recurrence on [st, en] (columns entirely outside the range are skipped with a uniform branch)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (NC > 1 && (c * 64 + 63 < st || c * 64 > en)) continue;
            const int t = c * 64 + lane;
            if (t >= st && t <= en) {
                int8_t z = (int8_t)s[c];
                const int8_t xt1 = t == st ? x1 : (int8_t)xs[c];
                const int8_t vt1 = t == st ? v1 : (int8_t)vs[c];
                const int8_t x2t1 = t == st ? x21 : (int8_t)x2s[c];
                const int8_t ut = (int8_t)u[c];
                int8_t a = (int8_t)(xt1 + vt1);
                int8_t b = (int8_t)((int8_t)y[c] + ut);
                int8_t a2 = (int8_t)(x2t1 + vt1);
                int8_t b2 = (int8_t)((int8_t)y2[c] + ut);
                uint8_t d;
                if (!RIGHT) {
                    d = a > z ? 1 : 0;
                    z = z > a ? z : a;
                    d = b > z ? 2 : d;
                    z = z > b ? z : b;
                    d = a2 > z ? 3 : d;
                    z = z > a2 ? z : a2;
                    d = b2 > z ? 4 : d;
                    z = z > b2 ? z : b2;
                    z = z < sc_mch ? z : sc_mch;
                } else {
                    d = z > a ? 0 : 1;
                    z = z > a ? z : a;
                    d = z > b ? d : 2;
                    z = z > b ? z : b;
                    d = z > a2 ? d : 3;
                    z = z > a2 ? z : a2;
                    d = z > b2 ? d : 4;
                    z = z > b2 ? z : b2;
                    z = z < sc_mch ? z : sc_mch;
                }
                u[c] = (int8_t)(z - vt1);
                v[c] = (int8_t)(z - ut);
                int8_t tmp = (int8_t)(z - q);
                a = (int8_t)(a - tmp);
                b = (int8_t)(b - tmp);
                tmp = (int8_t)(z - q2);
                a2 = (int8_t)(a2 - tmp);
                b2 = (int8_t)(b2 - tmp);
                if (!RIGHT) {
                    x[c] = (int8_t)((a > 0 ? a : 0) - qe);
                    d |= a > 0 ? 0x08 : 0;
                    y[c] = (int8_t)((b > 0 ? b : 0) - qe);
                    d |= b > 0 ? 0x10 : 0;
                    x2[c] = (int8_t)((a2 > 0 ? a2 : 0) - (q2 + e2));
                    d |= a2 > 0 ? 0x20 : 0;
                    y2[c] = (int8_t)((b2 > 0 ? b2 : 0) - (q2 + e2));
                    d |= b2 > 0 ? 0x40 : 0;
                } else {
                    x[c] = (int8_t)((0 > a ? 0 : a) - qe);
                    d |= 0 > a ? 0 : 0x08;
                    y[c] = (int8_t)((0 > b ? 0 : b) - qe);
                    d |= 0 > b ? 0 : 0x10;
                    x2[c] = (int8_t)((0 > a2 ? 0 : a2) - (q2 + e2));
                    d |= 0 > a2 ? 0 : 0x20;
                    y2[c] = (int8_t)((0 > b2 ? 0 : b2) - (q2 + e2));
                    d |= 0 > b2 ? 0 : 0x40;
                }
                PMX_TB_STORE(p, TB_IN_LDS, (uint32_t)(r * n_col + (t - st)), d);
            }
        }
        if (lane == 0) { off[r] = st; off_end[r] = en; }
        if (!approx_max) {   // exact max through H[] (:323-366)
            int32_t max_H, max_t;
            if (r > 0) {
                const int32_t h_en0 = en0 > 0 ? rd_col<NC>(H, en0 - 1) + (int8_t)rd_col<NC>(u, en0) : rd_col<NC>(H, en0) + (int8_t)rd_col<NC>(v, en0);
                const int en1 = st0 + (en0 - st0) / 4 * 4;
                // H[] update, then the maximum with the SSE loop's candidate order: en0 first, then the four
                // t-classes (t - st0) mod 4 of the vector part [st0, en1) in class order, each by ascending t,
                // then the scalar tail [en1, en0) by ascending t.  Value by a 32-bit wave max, position from
                // ballots of the lanes that hold it.
                int hv[NC], lmax = INT32_MIN;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    hv[c] = INT32_MIN;
                    if (NC > 1 && (c * 64 + 63 < st0 || c * 64 > en0)) continue;
                    const int t = c * 64 + lane;
                    if (t >= st0 && t < en0) {
                        const int32_t h = H[c] + (int32_t)(int8_t)v[c];
                        H[c] = h;
                        hv[c] = h;
                        lmax = h > lmax ? h : lmax;
                    }
                    if (t == en0) H[c] = h_en0;
                }
                const int wmax = wave_max_i32x(lmax);
                if (h_en0 >= wmax) { max_H = h_en0; max_t = en0; }
                else {
                    // position: every lane that holds the maximum offers its candidate priority (larger = earlier
                    // in the SSE loop's order); one more wave max picks the winner
                    uint32_t lp = 0;
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const int t = c * 64 + lane;
                        if (hv[c] == wmax && t >= st0 && t < en0) {
                            const uint32_t pr = t < en1 ? 0x7ffffffeu - ((uint32_t)((t - st0) & 3) << 24) - (uint32_t)((t - st0) >> 2)
                                                        : 0x7ffffffeu - (5u << 24) - (uint32_t)(t - st0);
                            lp = pr > lp ? pr : lp;
                        }
                    }
                    const uint32_t wp = (uint32_t)wave_max_i32x((int)lp);
                    const uint32_t dd = 0x7ffffffeu - wp;
                    const uint32_t cls = dd >> 24, idx = dd & 0xffffffu;
                    max_H = wmax;
                    max_t = cls >= 5 ? st0 + (int)idx : st0 + (int)(idx * 4 + cls);
                }
            } else {
                if (lane == 0) H[0] = (int8_t)v[0] - qe;
                max_H = rl_i32(H[0], 0);
                max_t = 0;
            }
            const int32_t H_en0 = rd_col<NC>(H, en0), H_st0 = rd_col<NC>(H, st0);
            if (en0 == tlen - 1 && H_en0 > ez.mte) { ez.mte = H_en0; ez.mte_q = r - en0; }
            if (r - st0 == qlen - 1 && H_st0 > ez.mqe) { ez.mqe = H_st0; ez.mqe_t = st0; }
            if (ez_apply_zdrop(ez, max_H, r, max_t, zdrop, e2)) break;
            if (r == qlen + tlen - 2 && en0 == tlen - 1) ez.score = rd_col<NC>(H, tlen - 1);
        } else {             // approximate max along one path (:367-383)
            if (r > 0) {
                if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
                    const int32_t d0 = (int8_t)rd_col<NC>(v, last_H0_t), d1 = (int8_t)rd_col<NC>(u, last_H0_t + 1);
                    if (d0 > d1) H0 += d0;
                    else { H0 += d1; ++last_H0_t; }
                } else if (last_H0_t >= st0 && last_H0_t <= en0) {
                    H0 += (int8_t)rd_col<NC>(v, last_H0_t);
                } else {
                    ++last_H0_t;
                    H0 += (int8_t)rd_col<NC>(u, last_H0_t);
                }
            } else { H0 = (int8_t)rd_col<NC>(v, 0) - qe; last_H0_t = 0; }
            if ((flag & PMX_EZ_APPROX_DROP) && ez_apply_zdrop(ez, H0, r, last_H0_t, zdrop, e2)) break;
            if (r == qlen + tlen - 2 && en0 == tlen - 1) ez.score = H0;
        }
        last_st = st;
        last_en = en;
        if (W.prof) ++W.dp_run_calls;   // diagnostic: anti-diagonals actually filled (the serve kernel reports them)
    }
    wave_sync();
    if (W.prof) W.prof_t = (unsigned long long)clock64();   // diagnostic: the serve kernel reports fill vs traceback
    const int rev_cigar = !!(flag & PMX_EZ_REV_CIGAR);
    if (!ez.zdropped && !(flag & PMX_EZ_EXTZ_ONLY)) {
        ksw_backtrack(W, rev_cigar, p, off, off_end, n_col, tlen - 1, qlen - 1, &ez.n_cigar);
    } else if (!ez.zdropped && (flag & PMX_EZ_EXTZ_ONLY) && ez.mqe + end_bonus > (int)ez.max) {
        ez.reach_end = 1;
        ksw_backtrack(W, rev_cigar, p, off, off_end, n_col, ez.mqe_t, qlen - 1, &ez.n_cigar);
    } else if (ez.max_t >= 0 && ez.max_q >= 0) {
        ksw_backtrack(W, rev_cigar, p, off, off_end, n_col, ez.max_t, ez.max_q, &ez.n_cigar);
    }
    wave_sync();
}

template <int NC, bool TB_IN_LDS>
__device__ __forceinline__ void ksw_extd2_reg(Work& W, int qlen, const uint8_t* query, int tlen, const uint8_t* target, const int8_t* mat, int8_t q,
                                              int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, Ez& ez) {
    if (flag & PMX_EZ_RIGHT) ksw_extd2_reg_t<NC, TB_IN_LDS, true>(W, qlen, query, tlen, target, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, ez);
    else ksw_extd2_reg_t<NC, TB_IN_LDS, false>(W, qlen, query, tlen, target, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, ez);
}
#elif defined(__HIPCC__)
// (device-only; declared so the host pass of the kernels parses)
template <int NC, bool TB_IN_LDS>
__device__ void ksw_extd2_reg(Work& W, int qlen, const uint8_t* query, int tlen, const uint8_t* target, const int8_t* mat, int8_t q, int8_t e, int8_t q2,
                   int8_t e2, int w, int zdrop, int end_bonus, int flag, Ez& ez);
#endif


#if PMX_W > 1
__device__ __forceinline__ int wave_sum_i32(int v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
    for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(v, o); v = other > v ? other : v; }
    return v;
}
#else
PMX_HD int wave_sum_i32(int v) { return v; }
PMX_HD int wave_max_i32(int v) { return v; }
#endif

// Number of positions i < n where a[i] != b[i] or a base is ambiguous (code >= 4), saturating early is
// not needed: n <= a few hundred.  Lane-parallel.
PMX_HD int count_diff(Ptr<const uint8_t> a, Ptr<const uint8_t> b, int n) {
    PMX_LDS(a); PMX_LDS(b);
    int d = 0;
    for (int i = lane_id(); i < n; i += PMX_W) d += (a[i] != b[i] || a[i] > 3) ? 1 : 0;
    return wave_sum_i32(d);
}

// ksw_extd2 with shortcuts whose results are provably what the DP returns (DESIGN.md "DP shortcuts"):
//  (1) extension (EXTZ_ONLY) of a query that equals the target prefix base for base (no ambiguous base),
//      band and z-drop not binding: the main diagonal is the unique optimum -> max = qlen*a at
//      (qlen-1,qlen-1), reach_end iff end_bonus > 0, CIGAR = qlen M;
//  (1b) the same with exactly ONE mismatch (at query position p) and (a+b) < min(q+e,q2+e2): any path that
//      ends off the main diagonal pays a gap (>= gmin) and aligns at most min(i,j)+1 pairs, so it scores
//      < a*(m+1) - (a+b) <= H(m,m); any gapped path back onto the diagonal pays two gaps.  Hence every
//      maximum the DP tracks sits on the diagonal with H(m,m) = a*(m+1) - (m >= p ? a+b : 0): the running
//      max is the first of { a*p at (p-1,p-1), a*qlen-(a+b) at the end } to reach the larger value (strict
//      '>' keeps the earlier on ties), mqe = a*qlen-(a+b) at t = qlen-1, the traceback is all-diagonal;
//  (2) global alignment in the approximate-max first pass of two equal-length sequences without
//      ambiguous bases whose Hamming distance d satisfies d*(a+b) <= a + 2*min(q+e,q2+e2) (strictly less when
//      gaps are right-aligned): no gapped alignment scores more than the gap-free one for any prefix pair on the
//      main diagonal, so the traceback is all-diagonal -> score = len*a - d*(a+b), CIGAR = len M.
// Everything else runs the DP.
// QF / TF: callables returning the query / target base at position i (so the bases can come from the work
// arena, straight from the reference in global memory, or be read back to front for a left extension)
// The decisions of the shortcuts, separated from how the mismatch statistics were obtained (a per-base scan here, a
// packed XOR in the compact tier, aln_compact.hpp): both tiers answer from the SAME code.
// Extension (EXTZ_ONLY): d = positions i < qlen where the bases differ or one is ambiguous, pf / pm = the smallest /
// largest such position.  cig0 receives the single CIGAR operation when ez.n_cigar == 1.
template <class QF, class TF>
PMX_HD bool ksw_shortcut_ext_decide(int qlen, int tlen, int d, int pf, int pm, QF& qf, TF& tf, int a, int b, int8_t q, int8_t e, int8_t q2,
                                    int8_t e2, int zdrop, int end_bonus, Ez& ez, uint32_t* cig0) {
    const int g1 = q + e, g2 = q2 + e2;
    const int gmin = g1 < g2 ? g1 : g2, gmax = g1 > g2 ? g1 : g2;
    if (d == 0) {
        ez_reset(ez);
        ez.max = (uint32_t)(qlen * a);
        ez.max_t = ez.max_q = qlen - 1;
        ez.mqe = qlen * a;
        ez.mqe_t = qlen - 1;
        if (tlen == qlen) { ez.mte = qlen * a; ez.mte_q = qlen - 1; }
        ez.reach_end = ez.mqe + end_bonus > (int)ez.max ? 1 : 0;
        *cig0 = (uint32_t)qlen << 4;
        ez.n_cigar = 1;
        return true;
    }
    if (d == 1 && a + b < gmin) {
        const int pos = pm;
        if (qf(pos) <= 3 && tf(pos) <= 3) {   // a real mismatch, not an ambiguous base
            ez_reset(ez);
            const int hend = qlen * a - (a + b);
            if (pos >= 1) { ez.max = (uint32_t)(pos * a); ez.max_t = ez.max_q = pos - 1; }
            if (hend > (int)ez.max) { ez.max = (uint32_t)hend; ez.max_t = ez.max_q = qlen - 1; }
            ez.mqe = hend;
            ez.mqe_t = qlen - 1;
            if (tlen == qlen) { ez.mte = hend; ez.mte_q = qlen - 1; }
            ez.reach_end = ez.mqe + end_bonus > (int)ez.max ? 1 : 0;
            const int len = ez.reach_end ? qlen : ez.max_q + 1;
            if (len > 0) { *cig0 = (uint32_t)len << 4; ez.n_cigar = 1; }
            return true;
        }
    }
    // (1c) exactly TWO mismatches p1 < p2 (no ambiguous base), both at least 6 bases before the query end.
    // A path with two or more gaps costs >= 2*gmin > 2(a+b) more than it can win back, a single gap longer
    // than G = the largest k with min(q+ke, q2+ke2) <= 2(a+b) likewise; a single gap of length <= G can only
    // rival the diagonal if it dodges BOTH mismatches without meeting a new one, i.e. if the diagonal shifted
    // by that gap matches the target everywhere between p1 and p2.  If every such shifted diagonal (both
    // directions, 1..G) has a mismatch in [p1, p2], every cell off the main diagonal scores strictly below
    // the diagonal cell of its row/column minimum, and the results follow from
    // H(m,m) = a(m+1) - (a+b)[m >= p1] - (a+b)[m >= p2] as in (1b).  (checked against the DP on millions of
    // random problems with tandem repeats: tests/test_align_host.py)
    if (d == 2 && 2 * (a + b) < 2 * gmin && zdrop >= 2 * gmax + a + 2 * (a + b)) {
        const int p1 = pf, p2 = pm;
        bool ok = p2 <= qlen - 6 && qf(p1) <= 3 && tf(p1) <= 3 && qf(p2) <= 3 && tf(p2) <= 3;
        int G = 0;
        for (int k = 1; k <= 64 && ok; ++k) {
            const int gk = (q + k * e) < (q2 + k * e2) ? (q + k * e) : (q2 + k * e2);
            if (gk <= 2 * (a + b)) G = k;
            else break;
        }
        for (int k = 1; k <= G && ok; ++k) {
            // query shifted forward by k against the target (k query bases inserted before the shifted run),
            // and the target shifted forward by k (k target bases deleted): a mismatch must exist at some
            // u in [p1, p2]; a shifted diagonal that runs off a sequence before p2 cannot reach that far
            bool hit_i = false, hit_d = false;
            for (int u = p1; u <= p2; ++u) {
                if (u + k >= qlen || qf(u + k) != tf(u) || qf(u + k) > 3) hit_i = true;
                if (u + k >= tlen || qf(u) != tf(u + k) || tf(u + k) > 3) hit_d = true;
            }
            ok = hit_i && hit_d;
        }
        if (ok) {
            ez_reset(ez);
            const int ab = a + b;
            const int h1 = p1 * a, h2 = p2 * a - ab, hend = qlen * a - 2 * ab;   // H at m = p1-1, p2-1, qlen-1
            if (p1 >= 1) { ez.max = (uint32_t)h1; ez.max_t = ez.max_q = p1 - 1; }
            if (h2 > (int)ez.max) { ez.max = (uint32_t)h2; ez.max_t = ez.max_q = p2 - 1; }
            if (hend > (int)ez.max) { ez.max = (uint32_t)hend; ez.max_t = ez.max_q = qlen - 1; }
            ez.mqe = hend;
            ez.mqe_t = qlen - 1;
            if (tlen == qlen) { ez.mte = hend; ez.mte_q = qlen - 1; }
            ez.reach_end = ez.mqe + end_bonus > (int)ez.max ? 1 : 0;
            const int len = ez.reach_end ? qlen : ez.max_q + 1;
            if (len > 0) { *cig0 = (uint32_t)len << 4; ez.n_cigar = 1; }
            return true;
        }
    }
    return false;
}
// pre-conditions shared by every shortcut, and the ones of the extension / gap-fill class
PMX_HD bool ksw_shortcut_applicable(int qlen, int tlen, int a, int b, int gmin, int w) {
    return qlen > 0 && tlen > 0 && a > 0 && b > 0 && (w < 0 || (w >= qlen && w >= tlen)) && b <= 2 * gmin;
}
PMX_HD bool ksw_shortcut_is_ext(int qlen, int tlen, int a, int b, int gmax, int zdrop, int flag) {
    return (flag & PMX_EZ_EXTZ_ONLY) && tlen >= qlen && zdrop >= 2 * gmax + a + (a + b);
}
PMX_HD bool ksw_shortcut_is_fill(int qlen, int tlen, int flag) {
    return !(flag & PMX_EZ_EXTZ_ONLY) && (flag & PMX_EZ_APPROX_MAX) && !(flag & PMX_EZ_APPROX_DROP) && qlen == tlen;
}
// Gap fill in the approximate-max first pass: d differing positions (an ambiguous query base counts), amb ambiguous ones
PMX_HD bool ksw_shortcut_fill_decide(int qlen, int d, int amb, int a, int b, int gmin, int flag, Ez& ez, uint32_t* cig0) {
    // gap-free = a*n - d(a+b); any gapped global path has >= 2 gaps and <= n-1 pairs: <= a(n-1) - 2*gmin.  With
    // left-aligned gaps (no RIGHT flag) a tie is harmless: the traceback takes a gap only when it is strictly
    // better (`d = a > z ? 1 : 0`), and the score is the same.
    const bool right = (flag & PMX_EZ_RIGHT) != 0;
    if (amb == 0 && (right ? d * (a + b) < a + 2 * gmin : d * (a + b) <= a + 2 * gmin)) {
        ez_reset(ez);
        ez.score = qlen * a - d * (a + b);
        *cig0 = (uint32_t)qlen << 4;
        ez.n_cigar = 1;
        return true;
    }
    return false;
}

template <class QF, class TF>
PMX_HD bool ksw_shortcut_f(Work& W, int qlen, QF& qf, int tlen, TF& tf, const int8_t* mat, int8_t q, int8_t e, int8_t q2, int8_t e2, int w,
                           int zdrop, int end_bonus, int flag, Ez& ez) {
    PMX_LDS(&W);
    Ptr<uint32_t> cig_tmp = W.cig_tmp; PMX_LDS(cig_tmp);
    const int a = mat[0], b = -mat[1];
    const int g1 = q + e, g2 = q2 + e2;
    const int gmin = g1 < g2 ? g1 : g2, gmax = g1 > g2 ? g1 : g2;
    if (!ksw_shortcut_applicable(qlen, tlen, a, b, gmin, w)) return false;
    uint32_t cig0 = 0;
    bool done = false;
    if (ksw_shortcut_is_ext(qlen, tlen, a, b, gmax, zdrop, flag)) {
        // d = differing or ambiguous positions among the first qlen; pf / pm: smallest / largest such position
        int d = 0, pm = -1, pf = INT32_MAX;
        for (int i = lane_id(); i < qlen; i += PMX_W) {
            const uint32_t cq = qf(i), ct = tf(i);
            if (cq != ct || cq > 3 || ct > 3) { ++d; pm = i; pf = i < pf ? i : pf; }
        }
        d = wave_sum_i32(d);
        if (d > 0 && d <= 2) { pm = wave_max_i32(pm); pf = -wave_max_i32(-pf); }
        done = ksw_shortcut_ext_decide(qlen, tlen, d, pf, pm, qf, tf, a, b, q, e, q2, e2, zdrop, end_bonus, ez, &cig0);
    } else if (ksw_shortcut_is_fill(qlen, tlen, flag)) {
        // differing positions, and ambiguous bases (any of those forces the DP)
        int d = 0, amb = 0;
        for (int i = lane_id(); i < qlen; i += PMX_W) {
            const uint32_t cq = qf(i), ct = tf(i);
            d += (cq != ct || cq > 3) ? 1 : 0;
            amb += (cq > 3 || ct > 3) ? 1 : 0;
        }
        d = wave_sum_i32(d);
        amb = wave_sum_i32(amb);
        done = ksw_shortcut_fill_decide(qlen, d, amb, a, b, gmin, flag, ez, &cig0);
    }
    if (done) {
        if (ez.n_cigar > 0) cig_tmp[0] = cig0;
        wave_sync();
    }
    return done;
}

// base readers for ksw_shortcut_f
template <class R>
struct FwdBases {
    R r;
    PMX_HD explicit FwdBases(R r_) : r(r_) {}
    PMX_HD uint32_t operator()(int i) { return r[i]; }
};
template <class R>
struct RevBases {   // position i of the reversed sequence of length n
    R r;
    int n;
    PMX_HD RevBases(R r_, int n_) : r(r_), n(n_) {}
    PMX_HD uint32_t operator()(int i) { return r[n - 1 - i]; }
};

PMX_HD bool ksw_shortcut(Work& W, int qlen, Ptr<const uint8_t> query, int tlen, Ptr<const uint8_t> target, const int8_t* mat, int8_t q, int8_t e,
                         int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, Ez& ez) {
    PMX_LDS(query); PMX_LDS(target);
    FwdBases<ByteReader> qf{ByteReader(query)}, tf{ByteReader(target)};
    return ksw_shortcut_f(W, qlen, qf, tlen, tf, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, ez);
}

// DP cells of one ksw call as SURVEY 8d counts them: q * min(t, 2w+1)
PMX_HD uint64_t dp_cells(int qlen, int tlen, int w) {
    if (qlen <= 0 || tlen <= 0) return 0;
    const int64_t band = w < 0 ? (int64_t)tlen : (int64_t)2 * w + 1;
    return (uint64_t)qlen * (uint64_t)(band < tlen ? band : tlen);
}

PMX_HD void ksw_extd2_auto(Work& W, int qlen, Ptr<const uint8_t> query, int tlen, Ptr<const uint8_t> target, const int8_t* mat, int8_t q, int8_t e,
                           int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, Ez& ez) {
    W.last_dp_shortcut = 0;
    if (!W.skip_shortcut && ksw_shortcut(W, qlen, query, tlen, target, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, ez)) {
        W.last_dp_shortcut = 1;
        return;
    }
#if defined(PMX_THREAD_PER_PAIR) && (defined(__HIP_DEVICE_COMPILE__) || defined(PMX_HOSTSIM_TPP))
    // the thread-per-pair kernel never runs a DP itself: it serves the call from the pair's result list, or posts it as a
    // request.  After the first request of a pass the pair is lost for this pass (PMX_ST_NEED_DP), but the pass goes on with
    // NEUTRAL results (nothing aligned, no Z-drop) so that the later DP calls of the pair -- the other extension, the other
    // regions, the other mate: their inputs come from the chains, not from earlier results -- are posted in the same pass and
    // served together.  Results are kept by call index; the replay is deterministic, so its c-th call is this pass's c-th
    // call unless an earlier result changes the flow (a Z-drop re-run), which the key check catches (the pair then goes
    // to the wave tier).
    // ... except a small one whose arrays fit the thread's own slab (the common case: an extension over a
    // mismatch near a read end): a scalar DP in this lane is cheaper than a request + replay round
    {
        int wb = w < 0 ? (tlen > qlen ? tlen : qlen) : w;
        int n_col = qlen < tlen ? qlen : tlen;
        n_col = (((n_col < wb + 1 ? n_col : wb + 1) + 15) / 16 + 1) * 16;
        if (!(W.status & PMX_ST_NEED_WAVE) && qlen > 0 && tlen > 0 && (tlen + 15) / 16 * 16 <= W.caps.max_tlen && qlen <= W.caps.max_tlen &&
            (size_t)(qlen + tlen - 1) * (size_t)n_col <= W.tb_cap) {
            ksw_extd2(W, qlen, query, tlen, target, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, ez);
            return;
        }
    }
    const uint32_t key = (uint32_t)qlen | (uint32_t)tlen << 10 | (uint32_t)(flag & 0xff) << 20;
    if (!(W.status & PMX_ST_NEED_WAVE)) {
        const int c = W.dp_calls++;
        if (c < W.dp_n_cached) {
            const DpRes& R = W.dp_res[c];
            if (R.key == key) {
                ez = R.ez;
                for (int i = 0; i < ez.n_cigar; ++i) W.cig_tmp[i] = R.cigar[i];
                return;
            }
            W.status |= PMX_ST_NEED_WAVE;
        } else {
            bool posted = false;
            const int e_ = c - W.dp_n_cached;   // request entry of this pass
            if (W.dp_req_base && c < PMX_DP_MAX_CALLS && e_ < PMX_DP_REQ_PER_PASS && ((qlen + 15) & ~15) + tlen <= PMX_DP_SEQ_BYTES) {
                if (W.dp_slot < 0 && W.dp_slot_ctr) {
                    const unsigned long long sl = atomicAdd(W.dp_slot_ctr, 1ULL);
                    if (sl < W.dp_slot_cap) {
                        W.dp_slot = (int64_t)sl;
                        for (int j = 0; j < PMX_DP_REQ_PER_PASS; ++j)
                            reinterpret_cast<DpReq*>(W.dp_req_base + ((size_t)sl * PMX_DP_REQ_PER_PASS + (size_t)j) * sizeof(DpReq))->call = 0xffffffffu;
                    }
                }
                if (W.dp_slot >= 0) {
                    DpReq* rq = reinterpret_cast<DpReq*>(W.dp_req_base + ((size_t)W.dp_slot * PMX_DP_REQ_PER_PASS + (size_t)e_) * sizeof(DpReq));
                    rq->qlen = qlen; rq->tlen = tlen; rq->w = w; rq->zdrop = zdrop; rq->end_bonus = end_bonus; rq->flag = flag;
                    rq->call = (uint32_t)c; rq->key = key;
                    uint8_t* sq = rq->seq;
                    for (int i = 0; i < qlen; ++i) sq[i] = query[i];
                    sq += (qlen + 15) & ~15;
                    for (int i = 0; i < tlen; ++i) sq[i] = target[i];
                    posted = true;
                    if (c == W.dp_post_end) W.dp_post_end = c + 1;
                }
            }
            if (!(W.status & PMX_ST_NEED_DP)) {   // the first miss of the pass
                if (posted) { W.status_pre = W.status; W.status |= PMX_ST_NEED_DP; }
                else W.status |= PMX_ST_NEED_WAVE;
            }   // (a later call that cannot be posted is met again, as a first miss, by the replay)
        }
    }
    ez_reset(ez);
    if (W.status & PMX_ST_NEED_WAVE) ez.zdropped = 1;   // this pass is over; after a posted request: neutral result, the pass goes on
#else
    ++W.dp_run_calls;
    W.dp_run_cells += dp_cells(qlen, tlen, w);
    ksw_extd2(W, qlen, query, tlen, target, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, ez);
#endif
}

}  // namespace aln
}  // namespace pmx
