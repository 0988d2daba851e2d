// ksw_extd2 for the wave-per-read kernels (long reads), ROW BY ROW: the 64 lanes of the wave own SW consecutive target
// columns each, with the columns' state in registers, and walk the query rows one after the other -- lane k starts row i one
// step after lane k-1 finished it and receives the row's right edge (x, v, x2, H of that lane's last column) through one DPP
// wave_shr:1 per value.  No LDS round trip and no barrier inside the fill, where the anti-diagonal kernel (ksw_extd2_t)
// pays two of each per diagonal: the gap fills between the anchors of a 10 kb read (a few hundred bases each way) ran at
// a fifth of that kernel's own instruction bound.
//
// Same scheme, same exactness argument as the grouped service of the short reads (align_kernel_dpg.hip): the cell function
// is ksw2_extd2_sse.c:168-321 on the same neighbours, only the visiting order differs, and that cannot change a value as
// long as the band never cuts the matrix -- the function is entered only when w >= max(qlen, tlen) - 1 (the long-read
// presets fill gaps with bw_long = 30,001).  What the reference derives diagonal by diagonal in order (exact maximum with
// the SSE candidate order, mqe / mte, Z-drop and its early exit) is replayed after the fill from three LDS arrays; the
// approximate-maximum mode of the first pass needs none of them (its H0 walks one path to the corner: the score is H there).
// The traceback matrix is row-major in the wave's HBM slab; the walk reads it through 64-row x 32-column windows that the
// lanes fetch together (one HBM round trip per >= 32 steps instead of one per step).
#pragma once
#include "aln_ksw_cell.hpp"
#include "aln_types.hpp"

#if PMX_W == 64 && defined(__HIP_DEVICE_COMPILE__)
namespace pmx {
namespace aln {

#define PMX_ROWS_BIAS (1 << 18)
#define PMX_ROWS_MAX_SW 16
#define PMX_ROWS_WIN_BYTES 2048   // 64 rows x 32 columns

// LDS the kernel needs: the traceback window and the query copy in one area, the replay arrays of the exact mode in another
PMX_HD size_t ksw_rows_lds_main(int qlen) { return (size_t)PMX_ROWS_WIN_BYTES + (size_t)((qlen + 15) & ~15); }
PMX_HD size_t ksw_rows_lds_arrays(int qlen, int tlen, bool exact) { return exact ? (size_t)8 * (size_t)(qlen + tlen) + 8 : 0; }

__device__ __forceinline__ int rows_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false); }   // wave_shr:1

template <int SW, bool RIGHT, bool EXACT, class QP, class TP>
__device__ __attribute__((noinline)) void ksw_extd2_rows_t(Work& W, int8_t* lds, int8_t* lds_arr, int qlen, QP query, int tlen, TP target, int q, int e, int q2, int e2,
                                                           int sc_mch, int sc_mis, int sc_N, int zdrop, int end_bonus, int flag, Ez& ez) {
    constexpr int ROW = 64 * SW;
    const int k = (int)(threadIdx.x & 63u);
    PMX_LDS_HERE(lds); PMX_LDS_HERE(lds_arr);
    uint8_t* win = reinterpret_cast<uint8_t*>(lds);
    uint8_t* qs = win + PMX_ROWS_WIN_BYTES;
    int32_t* lastcol = reinterpret_cast<int32_t*>(lds_arr);
    int32_t* lastrow = lastcol + qlen;
    uint32_t* diag = reinterpret_cast<uint32_t*>(lastrow + tlen);
    uint32_t* spare = diag + qlen + tlen;   // where the columns past the target put what they compute
    // (the traceback matrix of a long read's DP is in the wave's HBM slab: say so -- through a generic pointer the row's bytes leave
    //  as flat_store, which counts on lgkmcnt too, and the next step's LDS read of the query base waits for the HBM round trip)
    typedef __attribute__((address_space(1))) uint8_t g_u8;
    typedef __attribute__((address_space(1))) uint32_t g_u32;
    typedef __attribute__((address_space(1))) uint4 g_u4;
    g_u8* tb = (g_u8*)(uint8_t*)W.tb;

    const int qe = q + e, qe2 = q2 + e2;
    const KswCellParams cellp{q, q2, qe, qe2, sc_mch, sc_mis, sc_N};
    const int init_ue = -qe, init_ue2 = -qe2;
    int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
    if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
    const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
    auto gap_head = [&](int r) { return r == 0 ? init_ue : r < long_thres ? -e : r == long_thres ? long_diff : -e2; };

    const int t0 = k * SW;
    for (int i = k; i < qlen; i += 64) qs[i] = (uint8_t)query[i];
    int u[SW], y[SW], y2[SW];
    uint32_t sfw[SW / 4];
#pragma unroll
    for (int c4 = 0; c4 < SW / 4; ++c4) {
        uint32_t wd = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int t = t0 + 4 * c4 + b;
            wd |= (t < tlen ? (uint32_t)target[t] : 0u) << (8 * b);
        }
        sfw[c4] = wd;
    }
#pragma unroll
    for (int c = 0; c < SW; ++c) {   // first row: the boundary values of ksw2_extd2_sse.c:160-166
        u[c] = gap_head(t0 + c);
        y[c] = init_ue;
        y2[c] = init_ue2;
    }
    if (EXACT)
        for (int j = k; j < qlen + tlen; j += 64) diag[j] = 0;
    __syncthreads();

    const int n_steps = qlen + 63;
    int xo = 0, vo = 0, x2o = 0, Ho = 0;   // the right edge of the row this lane finished last
    int H0 = -qe;                          // H of column 0 (lane 0): v[0] - qe on the first row, += v after (:326-340)
    int hcol = 0;                          // H of the last column, row by row (its owner lane)
    const bool owns_last = (tlen - 1) / SW == k;
    int qb_next = (int)qs[0];   // the query base of the lane's next row, requested one step ahead (an LDS round trip per step otherwise)
    for (int tau = 0; tau < n_steps; ++tau) {
        int xl = rows_shr1(xo), vl = rows_shr1(vo), x2l = rows_shr1(x2o), Hl = rows_shr1(Ho);
        const int i = tau - k;
        const int qb = qb_next;
        qb_next = (int)qs[i + 1 > 0 ? i + 1 : 0];
        // (opaque copies, and one empty asm statement per cell below: see align_kernel_dpg.hip -- without them the compiler hoists
        //  every per-column invariant out of the row loop and interleaves the cells of a row, and the live values spill)
        int t0v = t0, tlv = tlen, qlv = qlen;
        asm volatile("" : "+v"(t0v), "+v"(tlv), "+v"(qlv));
#pragma unroll
        for (int c4 = 0; c4 < SW / 4; ++c4) asm volatile("" : "+v"(sfw[c4]));
        if (i >= 0 && i < qlv && t0v < tlv) {
            if (k == 0) { xl = init_ue; x2l = init_ue2; vl = gap_head(i); Hl = 0; }
            const int rem_q = qlv - 1 - i;
            int tcur = t0v;
            uint32_t* drow = diag + i + t0v;
            int32_t* lr = lastrow + t0v;
            uint32_t tbw[SW / 4];
#pragma unroll
            for (int c4 = 0; c4 < SW / 4; ++c4) tbw[c4] = 0;
#pragma unroll
            for (int c = 0; c < SW; ++c) {
                const int t = tcur;
                const int sq = (int)(sfw[c >> 2] >> (8 * (c & 3)) & 0xffu);
                const int ut = u[c];
                int un, vn, xn, yn, x2n, y2n;
                uint32_t d;
                ksw_cell<RIGHT>(cellp, sq, qb, xl, vl, x2l, ut, y[c], y2[c], un, vn, xn, yn, x2n, y2n, d);
                // H(t, q) = H(t-1, q) + u (the same number as the reference's H[t] += v: both are H(t-1, q-1) + z); column 0: += v
                const int Hn = c == 0 ? (k == 0 ? H0 + vn : Hl + un) : Hl + un;
                if (c == 0) H0 = Hn;
                hcol = t == tlv - 1 ? Hn : hcol;
                u[c] = un; y[c] = yn; y2[c] = y2n;
                xl = xn; vl = vn; x2l = x2n; Hl = Hn;
                ++tcur;
                tbw[c >> 2] |= d << (8 * (c & 3));
                if (EXACT) {
                    *(t < tlv ? lr + c : reinterpret_cast<int32_t*>(spare)) = Hn;   // every row leaves its H here: what stays is the last row's
                    // candidate of its diagonal's maximum: value, then the SSE loop's candidate order (en0 first, the vector part
                    // [st0, en1) class by class, the scalar tail [en1, en0) last) -- one comparable word, collected with LDS atomics
                    const int dt = t < rem_q ? t : rem_q;                // t - st0
                    const int dn = tlv - 1 - t < i ? tlv - 1 - t : i;    // en0 - t
                    const int n3 = (dt + dn) & 3;
                    const uint32_t kv = 1u + ((uint32_t)(dt & 3) << 8 | (uint32_t)(dt >> 2));   // <= 1,024 (targets of up to 1,024 columns)
                    const uint32_t kt = 1025u + (uint32_t)dt;
                    const uint32_t kp = (dn <= n3 ? kt : kv) & (uint32_t)-(int)(dn != 0);
                    atomicMax(t < tlv ? drow + c : spare + 1, (uint32_t)(Hn + PMX_ROWS_BIAS) << 12 | (4095u - kp));
                }
                asm volatile("" : "+v"(xl), "+v"(vl), "+v"(x2l), "+v"(Hl), "+v"(tbw[c >> 2]), "+v"(u[c]), "+v"(y[c]), "+v"(y2[c]), "+v"(tcur), "+v"(hcol));
            }
            xo = xl; vo = vl; x2o = x2l; Ho = Hl;
            g_u32* trow = (g_u32*)(tb + (size_t)i * ROW + t0v);
#pragma unroll
            for (int c4 = 0; c4 < SW / 4; ++c4) trow[c4] = tbw[c4];
            if (EXACT && owns_last) lastcol[i] = hcol;
        }
    }
    __threadfence_block();
    __syncthreads();

    // replay of the per-diagonal bookkeeping (every lane, on the same values: ez stays wave-uniform)
    ez_reset(ez);
    if (EXACT) {
        for (int r = 0; r < qlen + tlen - 1; ++r) {
            const int st0 = r - qlen + 1 > 0 ? r - qlen + 1 : 0, en0 = tlen - 1 < r ? tlen - 1 : r;
            const uint32_t key = diag[r];
            const int32_t max_H = (int32_t)(key >> 12) - PMX_ROWS_BIAS;
            const uint32_t kp = 4095u - (key & 4095u);
            int max_t;
            if (kp == 0) max_t = en0;
            else if (kp <= 1024u) { const uint32_t v = kp - 1u; max_t = st0 + (int)((v & 255u) << 2 | v >> 8); }
            else max_t = st0 + (int)(kp - 1025u);
            if (en0 == tlen - 1) { const int32_t h = lastcol[r - en0]; if (h > ez.mte) { ez.mte = h; ez.mte_q = r - en0; } }
            if (r - st0 == qlen - 1) { const int32_t h = lastrow[st0]; if (h > ez.mqe) { ez.mqe = h; ez.mqe_t = st0; } }
            if (ez_apply_zdrop(ez, max_H, r, max_t, zdrop, (int8_t)e2)) break;
            if (r == qlen + tlen - 2) ez.score = lastcol[qlen - 1];
        }
    } else ez.score = __builtin_amdgcn_readlane(hcol, __builtin_amdgcn_readfirstlane((tlen - 1) / SW));   // the approximate maximum walks one path to the corner: H there (:367-383)
    // ksw_backtrack (ksw2.h:127-162); no cell of the walk lies outside the band here.  All lanes walk together.
    int bi = -1, bj = -1;
    if (!ez.zdropped && !(flag & PMX_EZ_EXTZ_ONLY)) { bi = tlen - 1; bj = qlen - 1; }
    else if (!ez.zdropped && (flag & PMX_EZ_EXTZ_ONLY) && ez.mqe + end_bonus > (int)ez.max) { ez.reach_end = 1; bi = ez.mqe_t; bj = qlen - 1; }
    else if (ez.max_t >= 0 && ez.max_q >= 0) { bi = ez.max_t; bj = ez.max_q; }
    int n_cigar = 0;
    if (bi >= 0 && bj >= 0) {
        uint32_t* cig = W.cig_tmp;   // (LDS or the wave's slab, by the layout)
        const int cap = W.caps.max_cigar;
        bool ovf = false;
        int cur_op = -1, cur_len = 0;   // (ksw_push_cigar with the open operation kept in registers)
        // (the walk meets the operations last to first.  A list wanted first to last is written from the end of the buffer
        //  backwards and moved to the front 64 entries per round trip afterwards -- the buffer is in the wave's HBM slab for a
        //  long read, and swapping it around in place was a dependent load and store per pair of entries)
        const bool backwards = !(flag & PMX_EZ_REV_CIGAR);
        auto flush = [&]() {
            if (cur_op < 0) return;
            if (n_cigar < cap) { cig[backwards ? cap - 1 - n_cigar : n_cigar] = (uint32_t)cur_len << 4 | (uint32_t)cur_op; ++n_cigar; }
            else ovf = true;
        };
        auto push = [&](int op, int len) {
            if (op == cur_op) cur_len += len;
            else { flush(); cur_op = op; cur_len = len; }
        };
        int wi0 = 1 << 30, wj0 = 1 << 30;
        int i = bi, j = bj, state = 0;
        while (i >= 0 && j >= 0) {
            if (i < wi0 || j < wj0) {   // (the walk only ever moves up and to the left)
                wi0 = (i & ~15) - 16 > 0 ? (i & ~15) - 16 : 0;
                wj0 = j - 63 > 0 ? j - 63 : 0;
                if (wj0 + k <= j) {
                    const g_u4* src = (const g_u4*)(tb + (size_t)(wj0 + k) * ROW + wi0);
                    const uint4 v0 = src[0], v1 = src[1];
                    *reinterpret_cast<uint4*>(win + k * 32) = v0;
                    *reinterpret_cast<uint4*>(win + k * 32 + 16) = v1;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            }
            const uint32_t tmp = win[(j - wj0) * 32 + (i - wi0)];
            if (state == 0) state = (int)(tmp & 7u);
            else if (!(tmp >> (state + 2) & 1u)) state = 0;
            if (state == 0) state = (int)(tmp & 7u);
            if (state == 0) { push(0, 1); --i; --j; }
            else if (state == 1 || state == 3) { push(2, 1); --i; }
            else { push(1, 1); --j; }
        }
        if (i >= 0) push(2, i + 1);
        if (j >= 0) push(1, j + 1);
        flush();
        if (ovf) W.status |= PMX_ST_OVERFLOW;
        if (backwards && n_cigar < cap) {   // (n_cigar == cap: the list already starts at the front)
            __syncthreads();
            for (int a0 = 0; a0 < n_cigar; a0 += 64) {
                const int a = a0 + k;
                uint32_t v = 0;
                if (a < n_cigar) v = cig[cap - n_cigar + a];
                if (a < n_cigar) cig[a] = v;   // (a chunk's sources lie above every entry written so far: cap - n_cigar > 0)
            }
        }
    }
    ez.n_cigar = n_cigar;
    __syncthreads();
}

// The dispatcher: true = the request was served here.  Taken: band never cutting the matrix, scoring parameters that keep the
// reference's int8 lanes far from wrapping (plain 32-bit arithmetic stands for them), a target of at most 1,024 columns, the
// traceback matrix within the wave's slab, window + query copy within `lds_bytes` of LDS at `lds`, the replay arrays of the exact
// mode within `arr_bytes` at `lds_arr`.
template <class QP, class TP>
__device__ __forceinline__ bool ksw_extd2_rows(Work& W, int8_t* lds, size_t lds_bytes, int8_t* lds_arr, size_t arr_bytes, int qlen, QP query, int tlen, TP target,
                                               const int8_t* mat, int q, int e, int q2, int e2, int w, int zdrop, int end_bonus, int flag, Ez& ez) {
    if (qlen < 1 || tlen < 1 || tlen > 64 * PMX_ROWS_MAX_SW) return false;
    if (__builtin_amdgcn_is_shared((const void*)(const uint8_t*)W.tb)) return false;   // (a traceback matrix in LDS: the DP service's small requests)
    const int longer = qlen > tlen ? qlen : tlen;
    if (!(w < 0 || w >= longer - 1)) return false;
    if (flag & ~(PMX_EZ_RIGHT | PMX_EZ_APPROX_MAX | PMX_EZ_EXTZ_ONLY | PMX_EZ_REV_CIGAR)) return false;
    const bool exact = !(flag & PMX_EZ_APPROX_MAX);
    if (!exact && (flag & PMX_EZ_EXTZ_ONLY)) return false;
    if (q2 + e2 < q + e) { int t_ = q; q = q2; q2 = t_; t_ = e; e = e2; e2 = t_; }
    int min_sc = mat[1], max_abs = 0;
    for (int t = 0; t < 25; ++t) {
        const int v = mat[t];
        if (t >= 1 && v < min_sc) min_sc = v;
        max_abs = (v < 0 ? -v : v) > max_abs ? (v < 0 ? -v : v) : max_abs;
    }
    if (-min_sc > 2 * (q + e)) return false;   // (ksw2_extd2_sse.c:100: the reference returns without aligning)
    if (q < 0 || e < 0 || q2 < 0 || e2 < 0 || 2 * (q2 + e2) + 2 * max_abs > 100) return false;
    const int sw = ((tlen + 63) / 64 + 3) & ~3;   // 4, 8, 12, 16 columns per lane
    if ((size_t)qlen * (size_t)(64 * sw) > W.tb_cap) return false;
    if (ksw_rows_lds_main(qlen) > lds_bytes || ksw_rows_lds_arrays(qlen, tlen, exact) > arr_bytes) return false;
    const int sc_mch = mat[0], sc_mis = mat[1], sc_N = mat[24] == 0 ? -e2 : mat[24];
    const bool right = (flag & PMX_EZ_RIGHT) != 0;
#define PMX_ROWS_CALL(SWV, R, X) ksw_extd2_rows_t<SWV, R, X>(W, lds, lds_arr, qlen, query, tlen, target, q, e, q2, e2, sc_mch, sc_mis, sc_N, zdrop, end_bonus, flag, ez)
#define PMX_ROWS_SW(SWV)                                         \
    do {                                                         \
        if (right) { if (exact) PMX_ROWS_CALL(SWV, true, true); else PMX_ROWS_CALL(SWV, true, false); }   \
        else { if (exact) PMX_ROWS_CALL(SWV, false, true); else PMX_ROWS_CALL(SWV, false, false); }        \
    } while (0)
    if (sw == 4) PMX_ROWS_SW(4);
    else if (sw == 8) PMX_ROWS_SW(8);
    else if (sw == 12) PMX_ROWS_SW(12);
    else PMX_ROWS_SW(16);
#undef PMX_ROWS_SW
#undef PMX_ROWS_CALL
    return true;
}

}  // namespace aln
}  // namespace pmx
#endif
