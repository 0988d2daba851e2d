// ALIGN stage, grouped DP service for gfx950: the ksw_extd2 requests of the thread-per-pair tier (short reads: extensions and
// gap fills of a few dozen to 128 bases), EIGHT LANES PER REQUEST, eight requests per wave.
//
// The wave-per-request service (k_align_dp_serve) walks the anti-diagonals of one matrix with one wave: a diagonal of a
// 100 x 100 extension has 10-50 cells, so most lanes idle, and every diagonal pays the wave-wide maximum and the boundary
// reads (round-2 counters: 250-535 wave instructions per diagonal, 32 GCUPS).  Here a lane owns SW consecutive target
// columns (SW = 4, 8, 12, 16 by target length) with their state in REGISTERS and walks the query rows one after the other;
// lane k of a group starts row i one step after lane k-1 finished it (a systolic pipeline: the only exchange is the row's
// right edge -- x, v, x2 and H of the lane's last column -- handed to the next lane through one DPP row_shr:1 per value).
// Every lane of the wave is busy except during the 7-step skew.
//
// Exactness.  The cell function is ksw2_extd2_sse.c:168-321 applied to the same neighbours; what differs is the ORDER the
// cells are visited in (row by row instead of diagonal by diagonal), which cannot change a cell's value as long as the band
// never cuts the matrix: the service takes a request only if w >= max(qlen, tlen) - 1 (then st/en are the matrix edges on
// every diagonal, and the 16-aligned rounding of the SSE loops only ever touches cells outside the matrix that no cell
// inside depends on).  Everything the reference derives diagonal by diagonal IN ORDER -- the exact maximum with the SSE
// loop's candidate order, mqe / mte, the Z-drop test and its early exit (:323-366) -- is replayed after the fill from three
// small LDS arrays: the per-diagonal maximum (value and candidate rank in one word, collected with LDS atomics), H along the
// last column and H along the last row.  Cells past a Z-drop are computed and ignored (the reference stops; the traceback
// starts at or before the drop, so their bytes are never read).  The int8 lanes of the reference never wrap for the scoring
// parameters the host admits (checked there), so the arithmetic is plain 32-bit.
#include <hip/hip_runtime.h>

#include "align_kernel.h"
#include "align_kernel_dpg.h"
#include "align/aln_ksw_cell.hpp"

namespace pmx {
namespace aln {

#define PMX_DPG_BIAS (1 << 20)

__global__ void __launch_bounds__(256) k_dpg_collect(DpgArgs D) {
    __shared__ uint32_t h[16];
    if (threadIdx.x < 16) h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t n = D.n_slots * PMX_DP_REQ_PER_PASS;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t slot = D.worklist ? (int64_t)D.worklist[idx % D.n_slots] : idx % D.n_slots;
        const uint32_t ent = (uint32_t)(slot * PMX_DP_REQ_PER_PASS + idx / D.n_slots);
        const DpReq* rq = reinterpret_cast<const DpReq*>(D.dp_req_base + (size_t)ent * sizeof(DpReq));
        uint32_t key = (uint32_t)PMX_DPG_NO_BUCKET << 8 | 0xffu;
        if (rq->call != 0xffffffffu) {
            const int qlen = rq->qlen, tlen = rq->tlen, w = rq->w, flag = rq->flag;
            const int longer = qlen > tlen ? qlen : tlen;
            const int kind = flag == PMX_EZ_APPROX_MAX ? 0 : flag == PMX_EZ_EXTZ_ONLY ? 1 : flag == (PMX_EZ_EXTZ_ONLY | PMX_EZ_RIGHT | PMX_EZ_REV_CIGAR) ? 2 : -1;
            if (kind >= 0 && qlen >= 1 && tlen >= 1 && longer <= PMX_DPG_MAXLEN && (w < 0 || w >= longer - 1)) {
                const int cls = (tlen - 1) / (4 * PMX_DPG_G);   // 0..3
                const uint32_t bucket = (uint32_t)(cls * PMX_DPG_KINDS + kind);
                key = bucket << 8 | (uint32_t)(PMX_DPG_MAXLEN - qlen);
                atomicAdd(&h[bucket], 1u);
            } else atomicAdd(&h[PMX_DPG_NO_BUCKET - 1], 1u);   // a request for the wave service
        }
        D.keys[idx] = key;
        D.ids[idx] = ent;
    }
    __syncthreads();
    if (threadIdx.x < 16 && h[threadIdx.x]) atomicAdd(&D.counts[threadIdx.x], h[threadIdx.x]);
}

// The few requests the grouped service left (a side beyond 128 bases, ...) when they are too few to be worth a launch of the
// wave-per-request service (a full launch lasts as long as its longest request, ~0.4 ms): their results are marked invalid, and
// the replay hands those pairs to the wave-per-pair tier, which runs its own DPs.
__global__ void __launch_bounds__(256) k_dpg_refuse_left(DpgArgs D) {
    const int64_t n = D.n_slots * PMX_DP_REQ_PER_PASS;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t slot = D.worklist ? (int64_t)D.worklist[idx % D.n_slots] : idx % D.n_slots;
        DpReq* rq = reinterpret_cast<DpReq*>(D.dp_req_base + ((size_t)slot * PMX_DP_REQ_PER_PASS + (size_t)(idx / D.n_slots)) * sizeof(DpReq));
        const uint32_t call = rq->call;
        if (call == 0xffffffffu) continue;
        if (call < PMX_DP_MAX_CALLS) D.dp_res_base[(size_t)slot * PMX_DP_MAX_CALLS + call].key = 0xffffffffu;
        rq->call = 0xffffffffu;
    }
}

__device__ __forceinline__ int dpg_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); }   // row_shr:1

// (not inlined: twelve instantiations in one kernel body let the compiler hoist every variant's lane-invariant values out of
//  the task loop, ~1,000 spilled registers; as functions each variant has its own allocation)
template <int SW, int KIND>
__device__ __attribute__((noinline)) void dpg_serve(const DpgArgs& D, const uint32_t* ids, int n_here, uint8_t* lds, uint8_t* tb_wave) {
    constexpr int ROW = PMX_DPG_G * SW;   // bytes of one traceback row
    constexpr bool exact = KIND != 0, RIGHT = KIND == 2;
    const int lane = (int)(threadIdx.x & 63u), k = lane & (PMX_DPG_G - 1), g = lane / PMX_DPG_G;
    uint8_t* blk = lds + (size_t)g * PMX_DPG_LDS_PER_REQ;
    uint8_t* qs = blk;
    uint32_t* diag = reinterpret_cast<uint32_t*>(blk + PMX_DPG_MAXLEN);
    int32_t* lastcol = reinterpret_cast<int32_t*>(diag + 2 * PMX_DPG_MAXLEN);
    int32_t* lastrow = lastcol + PMX_DPG_MAXLEN;
    uint32_t* cig = reinterpret_cast<uint32_t*>(lastrow + PMX_DPG_MAXLEN);
    uint8_t* tb = tb_wave + (size_t)g * PMX_DPG_TB_PER_REQ;

    const unsigned long long pt0 = D.prof ? (unsigned long long)clock64() : 0ULL;
    bool valid = g < n_here;
    DpReq* rq = nullptr;
    uint32_t ent = 0;
    int qlen = 0, tlen = 0;
    if (valid) {
        ent = ids[g];
        if (ent < D.n_entries) {
            rq = reinterpret_cast<DpReq*>(D.dp_req_base + (size_t)ent * sizeof(DpReq));
            qlen = rq->qlen;
            tlen = rq->tlen;
        }
        // (what k_dpg_collect admitted; anything else would index past the per-request arrays: counted, left to the wave service)
        if (ent >= D.n_entries || qlen < 1 || tlen < 1 || qlen > PMX_DPG_MAXLEN || tlen > PMX_DPG_G * SW) {
            if (k == 0) atomicAdd(&D.counts[PMX_DPG_NO_BUCKET], 1u);
            valid = false;
            qlen = tlen = 0;
        }
    }
    const int t0 = k * SW;
    const int qe = D.q + D.e, qe2 = D.q2 + D.e2;
    const int init_ue = -qe, init_ue2 = -qe2;
    const KswCellParams cellp{D.q, D.q2, D.q + D.e, D.q2 + D.e2, D.sc_mch, D.sc_mis, D.sc_N};
    auto gap_head = [&](int r) { return r == 0 ? init_ue : r < D.long_thres ? -D.e : r == D.long_thres ? D.long_diff : -D.e2; };

    // the query to LDS (one byte per row and lane later), the lane's target bases to registers
    int u[SW], y[SW], y2[SW];
    uint32_t sfw[SW / 4];   // four bases per register (the compares select the byte)
#pragma unroll
    for (int c4 = 0; c4 < SW / 4; ++c4) sfw[c4] = 0;
    if (valid) {
        *reinterpret_cast<uint4*>(qs + k * 16) = *reinterpret_cast<const uint4*>(rq->seq + k * 16);
        const int t_off = (qlen + 15) & ~15;
#pragma unroll
        for (int c4 = 0; c4 < SW / 4; ++c4) sfw[c4] = *reinterpret_cast<const uint32_t*>(rq->seq + t_off + t0 + 4 * c4);
    }
#pragma unroll
    for (int c = 0; c < SW; ++c) {   // first row: the boundary values of ksw2_extd2_sse.c:160-166
        u[c] = gap_head(t0 + c);
        y[c] = init_ue;
        y2[c] = init_ue2;
    }
    if (exact)
        for (int j = k; j < 2 * PMX_DPG_MAXLEN; j += PMX_DPG_G) diag[j] = 0;
    int n_steps = valid ? qlen + PMX_DPG_G - 1 : 0;
    for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(n_steps, o); n_steps = other > n_steps ? other : n_steps; }
    __syncthreads();

    const unsigned long long pt1 = D.prof ? (unsigned long long)clock64() : 0ULL;
    int xo = 0, vo = 0, x2o = 0, Ho = 0;   // the right edge of the row this lane finished last
    int H0 = -qe;                          // H of column 0 (first lane of the group): H[0] = v[0] - qe on the first row, += v after (:326-340)
    const bool owns_last = valid && (tlen - 1) / SW == k;
    int qb_next = (int)qs[0];   // the query base of the lane's next row, requested one step ahead (an LDS round trip per step otherwise)
    for (int tau = 0; tau < n_steps; ++tau) {
        int xl = dpg_shr1(xo), vl = dpg_shr1(vo), x2l = dpg_shr1(x2o), Hl = dpg_shr1(Ho);
        const int i = tau - k;
        const int qb = qb_next;
        qb_next = (int)qs[i + 1 > 0 ? i + 1 : 0];
        // (opaque copies: everything the cells derive from the lane's column range and the lengths is a one-instruction
        //  value; hoisted out of the row loop as loop invariants -- two or three registers per column -- they spill)
        int t0v = t0, tlv = tlen, qlv = qlen;
        asm volatile("" : "+v"(t0v), "+v"(tlv), "+v"(qlv));
#pragma unroll
        for (int c4 = 0; c4 < SW / 4; ++c4) asm volatile("" : "+v"(sfw[c4]));
        if (i >= 0 && i < qlv && t0v < tlv) {
            if (k == 0) { xl = init_ue; x2l = init_ue2; vl = gap_head(i); Hl = 0; }
            const int rem_q = qlv - 1 - i;
            int hcol = 0, tcur = t0v;
            uint32_t* drow = diag + i + t0v;   // the diagonals of this row's cells
            int32_t* lr = lastrow + t0v;   // every row leaves its H values here: what stays is the last row's
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
                // H of the cell.  The reference adds v along the query (H[t] += v, :326-340); H(t-1, q) + u is the same number
                // (both are H(t-1, q-1) + z: u = z - v(t-1, q), v = z - u(t, q-1)), and that one is already in a register:
                // the left neighbour's H travels along the row with x and v.  Column 0 has no left neighbour: += v there.
                const int Hn = c == 0 ? (k == 0 ? H0 + vn : Hl + un) : Hl + un;
                if (c == 0) H0 = Hn;
                hcol = t == tlv - 1 ? Hn : hcol;
                u[c] = un; y[c] = yn; y2[c] = y2n;
                xl = xn; vl = vn; x2l = x2n; Hl = Hn;
                ++tcur;
                if (exact) lr[c] = Hn;
                tbw[c >> 2] |= d << (8 * (c & 3));
                if (exact) {
                    // the cell as a candidate of its diagonal's maximum: value, then the SSE loop's candidate order (en0 first,
                    // the vector part [st0, en1) class by class, the scalar tail [en1, en0) last), as one comparable word
                    const int dt = t < rem_q ? t : rem_q;                    // t - st0
                    const int dn = tlv - 1 - t < i ? tlv - 1 - t : i;      // en0 - t
                    const int n3 = (dt + dn) & 3;
                    const uint32_t kv = 1u + ((uint32_t)(dt & 3) << 7 | (uint32_t)(dt >> 2));
                    const uint32_t kt = 513u + (uint32_t)dt;
                    const uint32_t kp = (dn <= n3 ? kt : kv) & (uint32_t)-(int)(dn != 0);   // (no branch: en0 itself ranks first)
                    uint32_t key = (uint32_t)(Hn + PMX_DPG_BIAS) << 10 | (1023u - kp);
                    atomicMax(&drow[c], t < tlv ? key : 0u);   // (always: a branch per cell would cut the row into SW blocks)
                }
                // (the cells of a row form one dependency chain.  Left alone the compiler runs the chain of several cells first and
                //  the rest of each -- traceback flags, candidate ranks -- afterwards, and the intermediates of all of them spill
                //  (measured: a quarter of the loop's instructions were scratch traffic); a scheduling barrier does not help, the
                //  order is already fixed when the instructions are selected.  Passing every value a cell hands on through one
                //  empty asm statement makes the cell complete before the next one starts)
                asm volatile("" : "+v"(xl), "+v"(vl), "+v"(x2l), "+v"(Hl), "+v"(tbw[c >> 2]), "+v"(u[c]), "+v"(y[c]), "+v"(y2[c]), "+v"(tcur), "+v"(hcol));
            }
            xo = xl; vo = vl; x2o = x2l; Ho = Hl;
            uint8_t* trow = tb + (size_t)i * ROW + t0v;
            if constexpr (SW == 16) *reinterpret_cast<uint4*>(trow) = make_uint4(tbw[0], tbw[1], tbw[2], tbw[3]);
            else
#pragma unroll
                for (int c4 = 0; c4 < SW / 4; ++c4) reinterpret_cast<uint32_t*>(trow)[c4] = tbw[c4];
            if (owns_last) lastcol[i] = hcol;
        }
    }
    __threadfence_block();
    __syncthreads();
    const unsigned long long pt2 = D.prof ? (unsigned long long)clock64() : 0ULL;
    unsigned long long pt3 = pt2;

    // Replay of the per-diagonal bookkeeping, traceback and hand-over.  All eight lanes of a group run it, in step and on the
    // same values (they would idle otherwise): that way they can fetch the traceback matrix together -- a window of 16 rows x
    // 32 columns at a time into LDS (the per-diagonal array is free by then), one HBM round trip per >= 16 steps of the walk
    // instead of one per step (a 150-step walk was ~150 dependent loads: longer than the fill).  Only lane 0 writes.
    if (valid) {
        Ez ez;
        ez.max_q = ez.max_t = ez.mqe_t = ez.mte_q = -1;
        ez.max = 0;
        ez.score = ez.mqe = ez.mte = PMX_KSW_NEG_INF;
        ez.n_cigar = 0;
        ez.zdropped = 0;
        ez.reach_end = 0;
        const int flag = rq->flag, zdrop = rq->zdrop, end_bonus = rq->end_bonus;
        if (exact) {
            // The reference's loop over the diagonals (running maximum, Z-drop test, mqe / mte, :323-366) is sequential only in
            // the running maximum, and that is a prefix operation: the eight lanes take eight runs of consecutive diagonals.
            //  (1) every lane finds its run's first strict maximum; folding the runs before it (init: max = 0 at (-1, -1))
            //      gives the state the reference has when it enters the run;
            //  (2) with that state the lane walks its run as the reference does and notes the first diagonal where the Z-drop
            //      fires; the smallest such diagonal over the lanes is where the reference stops (every run before it was
            //      walked in full with the true state), and the state there is the result;
            //  (3) mte / mqe are first maxima of H along the last column / row up to that diagonal (inclusive: the reference
            //      updates them before the test), eight entries at a time.
            const int gb = lane & ~(PMX_DPG_G - 1);
            const int R = qlen + tlen - 1;
            const int per = (R + PMX_DPG_G - 1) / PMX_DPG_G;
            const int r0 = k * per, r1 = r0 + per < R ? r0 + per : R;
            auto diag_max = [&](int r, int32_t& H, int& t) {
                const int st0 = r - qlen + 1 > 0 ? r - qlen + 1 : 0, en0 = tlen - 1 < r ? tlen - 1 : r;
                const uint32_t key = diag[r];
                H = (int32_t)(key >> 10) - PMX_DPG_BIAS;
                const uint32_t kp = 1023u - (key & 1023u);
                if (kp == 0) t = en0;
                else if (kp <= 512u) { const uint32_t v = kp - 1u; t = st0 + (int)((v & 127u) << 2 | v >> 7); }
                else t = st0 + (int)(kp - 513u);
            };
            int32_t cV = INT32_MIN;
            int cT = -1, cR = -1;
            for (int r = r0; r < r1; ++r) {
                int32_t H; int t;
                diag_max(r, H, t);
                if (H > cV) { cV = H; cT = t; cR = r; }
            }
            int32_t mx = 0;
            int mt = -1, mq = -1;
            for (int j = 0; j < PMX_DPG_G - 1; ++j) {
                const int32_t V = __shfl(cV, gb + j);
                const int T = __shfl(cT, gb + j), Rj = __shfl(cR, gb + j);
                if (j < k && V > mx) { mx = V; mt = T; mq = Rj - T; }
            }
            int rb = INT32_MAX;
            for (int r = r0; r < r1; ++r) {
                int32_t H; int t;
                diag_max(r, H, t);
                if (H > mx) { mx = H; mt = t; mq = r - t; }
                else if (t >= mt && r - t >= mq) {
                    const int tl = t - mt, ql = (r - t) - mq;
                    const int l = tl > ql ? tl - ql : ql - tl;
                    if (zdrop >= 0 && mx - H > zdrop + l * D.e2) { rb = r; break; }
                }
            }
            int r_stop = INT32_MAX, owner = PMX_DPG_G - 1;   // no Z-drop: the last lane's state is the state after the last diagonal
            for (int j = 0; j < PMX_DPG_G; ++j) {
                const int rj = __shfl(rb, gb + j);
                if (rj < r_stop) { r_stop = rj; owner = j; }
            }
            ez.max = (uint32_t)__shfl(mx, gb + owner);
            ez.max_t = __shfl(mt, gb + owner);
            ez.max_q = __shfl(mq, gb + owner);
            ez.zdropped = r_stop != INT32_MAX ? 1 : 0;
            const int r_lim = ez.zdropped ? r_stop : R - 1;
            {
                int nq = r_lim - tlen + 2; nq = nq < qlen ? nq : qlen;      // rows whose last-column cell lies on a diagonal <= r_lim
                int nt = r_lim - qlen + 2; nt = nt < tlen ? nt : tlen;      // columns whose last-row cell does
                int32_t be = PMX_KSW_NEG_INF, bq = PMX_KSW_NEG_INF;
                int be_at = -1, bq_at = -1;
                for (int q_ = k; q_ < nq; q_ += PMX_DPG_G) { const int32_t h = lastcol[q_]; if (h > be) { be = h; be_at = q_; } }
                for (int t_ = k; t_ < nt; t_ += PMX_DPG_G) { const int32_t h = lastrow[t_]; if (h > bq) { bq = h; bq_at = t_; } }
                for (int j = 0; j < PMX_DPG_G; ++j) {   // first maximum = the largest value at the smallest position
                    const int32_t e_ = __shfl(be, gb + j), q2_ = __shfl(bq, gb + j);
                    const int ea = __shfl(be_at, gb + j), qa = __shfl(bq_at, gb + j);
                    if (ea >= 0 && (e_ > ez.mte || (e_ == ez.mte && ea < ez.mte_q))) { ez.mte = e_; ez.mte_q = ea; }
                    if (qa >= 0 && (q2_ > ez.mqe || (q2_ == ez.mqe && qa < ez.mqe_t))) { ez.mqe = q2_; ez.mqe_t = qa; }
                }
            }
            if (!ez.zdropped) ez.score = lastcol[qlen - 1];
        } else ez.score = lastcol[qlen - 1];   // the approximate maximum follows one path to the corner: H there (:367-383)
        if (D.prof) pt3 = (unsigned long long)clock64();
        // ksw_backtrack (ksw2.h:127-162): no cell of the walk lies outside the band here
        int bi = -1, bj = -1;
        if (!ez.zdropped && !(flag & PMX_EZ_EXTZ_ONLY)) { bi = tlen - 1; bj = qlen - 1; }
        else if (!ez.zdropped && (flag & PMX_EZ_EXTZ_ONLY) && ez.mqe + end_bonus > (int)ez.max) { ez.reach_end = 1; bi = ez.mqe_t; bj = qlen - 1; }
        else if (ez.max_t >= 0 && ez.max_q >= 0) { bi = ez.max_t; bj = ez.max_q; }
        bool bad = false;
        int n_cigar = 0;
        if (bi >= 0 && bj >= 0) {
            // (ksw_push_cigar with the open operation kept in registers)
            int cur_op = -1, cur_len = 0;
            auto flush = [&]() {
                if (cur_op < 0) return;
                if (n_cigar < PMX_DP_MAX_CIGAR) { if (k == 0) cig[n_cigar] = (uint32_t)cur_len << 4 | (uint32_t)cur_op; ++n_cigar; }
                else bad = true;
            };
            auto push = [&](int op, int len) {
                if (op == cur_op) cur_len += len;
                else { flush(); cur_op = op; cur_len = len; }
            };
            uint8_t* win = reinterpret_cast<uint8_t*>(diag);   // [16][32]
            int wi0 = 1 << 30, wj0 = 1 << 30;
            int i = bi, j = bj, state = 0;
            while (i >= 0 && j >= 0 && !bad) {
                if (i < wi0 || j < wj0) {   // (the walk only ever moves up and to the left)
                    wi0 = (i & ~15) - 16 > 0 ? (i & ~15) - 16 : 0;
                    wj0 = j - 15 > 0 ? j - 15 : 0;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int row = k + 8 * h;
                        const uint4* src = reinterpret_cast<const uint4*>(tb + (size_t)(wj0 + row) * ROW + wi0);
                        const uint4 v0 = src[0], v1 = src[1];
                        *reinterpret_cast<uint4*>(win + row * 32) = v0;
                        *reinterpret_cast<uint4*>(win + row * 32 + 16) = v1;
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
            if (k == 0 && !(flag & PMX_EZ_REV_CIGAR))
                for (int a = 0; a < n_cigar >> 1; ++a) { const uint32_t t_ = cig[a]; cig[a] = cig[n_cigar - 1 - a]; cig[n_cigar - 1 - a] = t_; }
        }
        ez.n_cigar = n_cigar;
        if (k == 0) {
            const uint32_t call = rq->call;
            if (call < PMX_DP_MAX_CALLS) {
                DpRes& R = D.dp_res_base[(size_t)(ent / PMX_DP_REQ_PER_PASS) * PMX_DP_MAX_CALLS + call];
                R.ez = ez;
                R.key = bad ? 0xffffffffu : rq->key;
                if (!bad) for (int a = 0; a < n_cigar; ++a) R.cigar[a] = cig[a];
            }
            if (D.stats) {
                atomicAdd(&D.stats[0], 1ULL);
                atomicAdd(&D.stats[1], (unsigned long long)dp_cells(qlen, tlen, rq->w));
            }
            if (!D.shadow) rq->call = 0xffffffffu;   // served
        }
    }
    __syncthreads();
    if (D.prof && lane == 0) {
        const unsigned long long pt4 = (unsigned long long)clock64();
        atomicAdd(&D.prof[0], pt1 - pt0); atomicAdd(&D.prof[1], pt2 - pt1); atomicAdd(&D.prof[2], pt3 - pt2); atomicAdd(&D.prof[3], pt4 - pt3);
        atomicAdd(&D.prof[4], 1ULL); atomicAdd(&D.prof[5], (unsigned long long)n_steps);
    }
}

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) k_align_dp_group(DpgArgs D) {
    extern __shared__ __attribute__((aligned(16))) uint8_t dpg_lds[];
    uint8_t* tb_wave = D.tb + (size_t)blockIdx.x * PMX_DPG_TB_BYTES;
    // buckets: requests of one (columns per lane, kind) class sit together in the sorted list; a wave takes eight of ONE bucket
    constexpr uint32_t per_wave = 64 / PMX_DPG_G;
    uint32_t n_tasks = 0;
    for (int b = 0; b < PMX_DPG_BUCKETS; ++b) n_tasks += (D.counts[b] + per_wave - 1) / per_wave;
    // Tasks are drawn from a counter (counts[PMX_DPG_WORK], zeroed with the bucket counts), the widest classes first and
    // within a class the longest requests first: a resident grid striding over the task list gave every wave the same
    // number of tasks whatever they cost and left the sixteen-column classes -- the list's end -- for last.
    for (;;) {
        uint32_t task = 0;
        if ((threadIdx.x & 63u) == 0) task = atomicAdd(&D.counts[PMX_DPG_WORK], 1u);
        task = (uint32_t)__builtin_amdgcn_readfirstlane((int)task);
        if (task >= n_tasks) break;
        uint32_t fb = 0, wb = 0, nb = 0;
        int b = PMX_DPG_BUCKETS - 1;
        for (;; --b) {   // (ends: task < n_tasks)
            nb = D.counts[b];
            const uint32_t wn = (nb + per_wave - 1) / per_wave;
            if (task < wb + wn) break;
            wb += wn;
        }
        for (int c = 0; c < b; ++c) fb += D.counts[c];
        const uint32_t at = (task - wb) * per_wave;
        const int n_here = (int)(nb - at < per_wave ? nb - at : per_wave);
        const uint32_t* ids = D.sorted_ids + fb + at;
        switch (b) {
            case 0: dpg_serve<4, 0>(D, ids, n_here, dpg_lds, tb_wave); break;
            case 1: dpg_serve<4, 1>(D, ids, n_here, dpg_lds, tb_wave); break;
            case 2: dpg_serve<4, 2>(D, ids, n_here, dpg_lds, tb_wave); break;
            case 3: dpg_serve<8, 0>(D, ids, n_here, dpg_lds, tb_wave); break;
            case 4: dpg_serve<8, 1>(D, ids, n_here, dpg_lds, tb_wave); break;
            case 5: dpg_serve<8, 2>(D, ids, n_here, dpg_lds, tb_wave); break;
            case 6: dpg_serve<12, 0>(D, ids, n_here, dpg_lds, tb_wave); break;
            case 7: dpg_serve<12, 1>(D, ids, n_here, dpg_lds, tb_wave); break;
            case 8: dpg_serve<12, 2>(D, ids, n_here, dpg_lds, tb_wave); break;
            case 9: dpg_serve<16, 0>(D, ids, n_here, dpg_lds, tb_wave); break;
            case 10: dpg_serve<16, 1>(D, ids, n_here, dpg_lds, tb_wave); break;
            default: dpg_serve<16, 2>(D, ids, n_here, dpg_lds, tb_wave); break;
        }
    }
}

}  // namespace aln
}  // namespace pmx
