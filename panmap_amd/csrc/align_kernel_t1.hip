// ALIGN stage, tier-1 kernel for gfx950: the compact layout keeps every work array except the traceback
// matrix in LDS, and this translation unit tells the compiler so (PMX_ALL_LDS -> PMX_LDS() assumes), so
// the per-pair bookkeeping compiles to ds_read/ds_write instead of flat_load/flat_store (the flat path
// was the measured bottleneck: ~9k flat memory instructions per pair, 56% of wave cycles waiting).
#define PMX_ALL_LDS 1
#include "align_kernel_body.hpp"

namespace pmx {
namespace aln {

__global__ void __launch_bounds__(64, 2) k_align_reads_t1(AlignArgs A) { align_reads_body<2>(A); }
__global__ void __launch_bounds__(64, 4) k_align_reads_t1_w4(AlignArgs A) { align_reads_body<4>(A); }

// DP service of the thread-per-pair kernel: one wave per posted request.  Class 1 (short targets: nearly all
// requests) runs the register-resident ksw_extd2_reg; the rest run the anti-diagonal-parallel ksw_extd2 with
// its arrays in LDS (traceback matrix in LDS when it fits, else in the wave's HBM slab).  The result is
// appended to the slot's result list.
__global__ void __launch_bounds__(64, 3) k_align_dp_serve(AlignArgs A) {
    if (A.dp_left && A.dp_left[0] + A.dp_left[1] == 0) return;   // the grouped service (align_kernel_dpg.hip) took every request
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    Work& W = *reinterpret_cast<Work*>(lds);
    PMX_LDS(&W);
    uint8_t* fast = lds + PMX_ALIGN_WORK_BYTES;
    uint8_t* slow = A.slow_base + (size_t)blockIdx.x * A.slow_stride;
    const int lane = (int)(threadIdx.x & 63u);
    // work items = request entries: PMX_DP_REQ_PER_PASS per pair slot, most of them empty (call != its place means: nothing
    // posted there in this pass, or already served)
    for (int64_t it = blockIdx.x; it < A.n_items * PMX_DP_REQ_PER_PASS; it += gridDim.x) {
        // (entry-major: with the slot-major order and a grid that is a multiple of the entries per slot, every wave would
        //  see one entry number only -- and nearly all requests sit in entry 0)
        const int64_t slot = A.worklist ? (int64_t)A.worklist[it % A.n_items] : it % A.n_items;
        DpReq* rq = reinterpret_cast<DpReq*>(A.dp_req_base + ((size_t)slot * PMX_DP_REQ_PER_PASS + (size_t)(it / A.n_items)) * sizeof(DpReq));
        if (rq->call == 0xffffffffu) continue;   // (wave-uniform)
        __syncthreads();
        bind_work(W, A.layout, fast, slow);
        W.prof = A.prof;
        W.prof_t = 0;
        W.dp_run_calls = 0;
        W.no_rows_dp = A.no_rows_dp;
        const int qlen = rq->qlen, tlen = rq->tlen;
        const int t_off = (qlen + 15) & ~15;
        const size_t tb_need = dp_request_tb_bytes(qlen, tlen, rq->w);
        if (A.dp_class != 0) {
            const bool small = qlen <= A.dp_small_qlen && tlen <= A.dp_small_tlen && tb_need <= (size_t)A.dp_small_tb;
            if (small != (A.dp_class == 1)) continue;   // the other launch serves it
        }
        if (tb_need <= A.layout.tb_fast_cap) {   // traceback matrix in LDS when it fits
            W.tb = fast + A.layout.tb_fast.off;
            if (W.tb_cap < A.layout.tb_fast_cap) W.tb_cap = A.layout.tb_fast_cap;
        }
        uint8_t* sq = W.qseq[0][0]; PMX_LDS(sq);
        __syncthreads();
        for (int i = lane; i < t_off + tlen; i += 64) sq[i] = rq->seq[i];
        __syncthreads();
        const unsigned long long tp1 = A.prof ? (unsigned long long)clock64() : 0ULL;
        Ez ez;
#define PMX_REG_DP(NC, TBL)                                                                                                            \
    ksw_extd2_reg<NC, TBL>(W, qlen, sq, tlen, sq + t_off, A.opt.mat, (int8_t)A.opt.q, (int8_t)A.opt.e, (int8_t)A.opt.q2, (int8_t)A.opt.e2, \
                           rq->w, rq->zdrop, rq->end_bonus, rq->flag, ez)
        if (A.dp_class == 1) {
            const bool tbl = tb_need <= A.layout.tb_fast_cap;
            if (tlen <= 64) { if (tbl) PMX_REG_DP(1, true); else PMX_REG_DP(1, false); }
            else if (tlen <= 128) { if (tbl) PMX_REG_DP(2, true); else PMX_REG_DP(2, false); }
            else { if (tbl) PMX_REG_DP(3, true); else PMX_REG_DP(3, false); }
        }
#undef PMX_REG_DP
        else
            ksw_extd2(W, qlen, sq, tlen, sq + t_off, A.opt.mat, (int8_t)A.opt.q, (int8_t)A.opt.e, (int8_t)A.opt.q2, (int8_t)A.opt.e2, rq->w,
                      rq->zdrop, rq->end_bonus, rq->flag, ez);
        __syncthreads();
        const unsigned long long tp2 = A.prof ? (unsigned long long)clock64() : 0ULL;
        if (lane == 0) {
            const uint32_t* cg = W.cig_tmp; PMX_LDS(cg);
            dp_store_result(A, slot, rq, ez, cg, W.status);
            rq->call = 0xffffffffu;   // served
            if (A.stats) {
                atomicAdd(&A.stats[0], 1ULL);
                atomicAdd(&A.stats[1], (unsigned long long)dp_cells(qlen, tlen, rq->w));
            }
        }
        if (A.prof && lane == 0) {
            atomicAdd(&A.prof[12], W.prof_t ? tp2 - W.prof_t : 0ULL);   // traceback part of the DP (register kernel)
            atomicAdd(&A.prof[13], tp2 - tp1);
            atomicAdd(&A.prof[14], (unsigned long long)clock64() - tp2);
            atomicAdd(&A.prof[15], (unsigned long long)(W.dp_run_calls ? W.dp_run_calls : qlen + tlen - 1));   // filled anti-diagonals (register kernel) / their bound
        }
    }
}

}  // namespace aln
}  // namespace pmx
