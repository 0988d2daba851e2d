// Body of the ALIGN kernels (included by align_kernel.hip and align_kernel_t1.hip): one wave64 (= one
// workgroup) per read pair, persistent over the batch.  The per-pair pipeline is
// align/aln_map.hpp::map_frag; the Work descriptor lives in LDS; which work arrays live in LDS and which
// in the per-wave HBM slab is the host planner's choice (aln_host.hpp).
#pragma once
#include <hip/hip_runtime.h>

#include "align/aln_host.hpp"
#include "align_kernel.h"

namespace pmx {
namespace aln {

template <int WAVES_PER_SIMD>
__device__ __forceinline__ void align_reads_body(const AlignArgs& A) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    Work& W = *reinterpret_cast<Work*>(lds);
    PMX_LDS(&W);
    uint8_t* fast = lds + PMX_ALIGN_WORK_BYTES;
    uint8_t* slow = A.slow_base + (size_t)blockIdx.x * A.slow_stride;
    const int lane = (int)(threadIdx.x & 63u);
    const int n_segs = A.paired ? 2 : 1;

    // work distribution: a strided loop, or (work_queue != NULL: long reads, whose ~25 items per wave differ by tens of
    // milliseconds) the next item from a device counter, so that a wave that drew quick reads takes more of them
    int64_t it = (int64_t)blockIdx.x - (int64_t)gridDim.x;
    for (;;) {
        if (A.work_queue) {
            unsigned long long nx = 0;
            if (lane == 0) nx = atomicAdd(A.work_queue, 1ULL);
            it = (int64_t)((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)nx) |
                           (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(nx >> 32)) << 32);
        } else it += gridDim.x;
        if (it >= A.n_items) break;
        int64_t item = A.worklist ? (int64_t)A.worklist[it] : it;
        int64_t ho_slot = -1;
        if (A.dp_slot_pairs) {   // worklist holds DP-service slots
            ho_slot = item;
            item = (int64_t)A.dp_slot_pairs[item];
            if (item == 0xffffffffLL) continue;   // slot whose pair already went to the retry list
        }
        __syncthreads();
        bind_work(W, A.layout, fast, slow);
        W.n_segs = n_segs;
        W.prof = A.prof;
        W.sk_no_lane_ring = A.sk_no_lane_ring;
        W.no_rows_dp = A.no_rows_dp;
        W.mv_ready = 0;
        if (A.mv_handover && ho_slot >= 0 && ho_slot < (int64_t)A.mv_slots) {   // minimizers left by the thread-per-pair kernel
            const A128* src = A.mv_handover + (size_t)ho_slot * A.mv_stride;
            const A128 hd = src[0];
            if (hd.y == ((uint64_t)A.mv_epoch << 32 | (uint32_t)item) && (int)hd.x <= W.caps.max_mini) {
                A128* mvp = W.mv; PMX_LDS(mvp);
                for (int j = lane; j < (int)hd.x; j += 64) mvp[j] = src[1 + j];
                W.n_mv = (int)hd.x;
                W.mv_ready = 1;
            }
        }
        if (A.prof) { W.prof_t = (unsigned long long)clock64(); for (int k = 0; k < 24; ++k) W.prof_acc[k] = 0; }
        bool too_long = false;
        for (int s = 0; s < n_segs; ++s) {
            const int64_t r = A.paired ? 2 * item + s : item;
            const int64_t len = A.off[r + 1] - A.off[r];
            if (len > A.layout.caps.max_qlen) too_long = true;
            W.qlen[s] = (int)len;
        }
        __syncthreads();
        if (!too_long) {
            for (int s = 0; s < n_segs; ++s) {
                const int64_t r = A.paired ? 2 * item + s : item;
                const int len = W.qlen[s];
                const uint64_t* rw = A.words + A.woff[r];
                const uint32_t* ra = A.amb + A.woff[r];
                uint8_t* fwd = W.qseq[s][0];
                uint8_t* rev = W.qseq[s][1];
                PMX_LDS(fwd); PMX_LDS(rev);
                const bool rc = A.revcomp_mate2 && s == 1;
                for (int i = lane; i < len; i += 64) {
                    const uint32_t code = (uint32_t)(rw[i >> 5] >> (2 * (i & 31))) & 3u;
                    const uint32_t am = (ra[i >> 5] >> (i & 31)) & 1u;
                    const uint8_t c = am ? (code == 3 ? 3 : 4) : (uint8_t)code;
                    const uint8_t cc = c < 4 ? (uint8_t)(3 - c) : (uint8_t)4;
                    if (!rc) { fwd[i] = c; rev[len - 1 - i] = cc; }
                    else { fwd[len - 1 - i] = cc; rev[i] = c; }
                }
            }
            __syncthreads();
            PMX_STAMP(W, 0);
            map_frag(W, A.opt, A.ri);
            __syncthreads();
        } else {
            W.status |= PMX_ST_OVERFLOW;
            W.n_regs[0] = W.n_regs[1] = 0;
        }
        if (A.retry_list && (W.status & PMX_ST_OVERFLOW)) {   // tier 1: hand the pair to the general-capacity launch
            if (lane == 0) A.retry_list[atomicAdd(A.retry_count, 1ULL)] = (uint32_t)item;
            continue;
        }
        const bool mapped = !too_long && frag_is_mapped(W, A.paired);
        for (int s = 0; s < n_segs; ++s) {
            const int64_t r = A.paired ? 2 * item + s : item;
            AlnRecord rec;
            memset(&rec, 0, sizeof(rec));
            rec.flags = (uint16_t)(W.status & 3u);
            if (mapped) {
                rec.mapped = 1;
                const Reg* gp_ = W.regs[s]; PMX_LDS(gp_);
                const Reg& g = gp_[0];
                if (g.has_p) {
                    rec.flags |= PMX_REC_HAS_ALN;
                    rec.rs = g.rs; rec.re = g.re; rec.qs = g.qs; rec.qe = g.qe;
                    rec.mapq = g.mapq; rec.rev = g.rev; rec.proper_frag = g.proper_frag;
                    rec.n_cigar = (uint16_t)g.n_cigar;
                    rec.score = g.dp_max;
                    __syncthreads();
                    if (lane == 0) W.tmp64 = atomicAdd(A.cigar_used, (unsigned long long)g.n_cigar);
                    __syncthreads();
                    const uint64_t coff = W.tmp64;
                    rec.cigar_off = (uint32_t)coff;
                    if (coff + g.n_cigar <= A.cigar_cap && g.n_cigar <= 0xffffu) {   // (n_cigar is 16 bits wide in the record)
                        const uint32_t* cg = reg_cigar(W, g); PMX_LDS(cg);
                        for (uint32_t i = lane; i < g.n_cigar; i += 64) A.cigars[coff + i] = cg[i];
                    } else {
                        rec.flags |= PMX_REC_OVERFLOW;
                        rec.n_cigar = 0;
                    }
                }
            }
            if (lane == 0) A.records[r] = rec;
            if (A.edits && lane == 0) A.edits[r] = too_long ? W.qlen[s] : read_errors(W, s);
        }
        if (A.stats && lane == 0 && W.dp_run_calls) {
            atomicAdd(&A.stats[0], (unsigned long long)W.dp_run_calls);
            atomicAdd(&A.stats[1], (unsigned long long)W.dp_run_cells);
            if (!A.dp_slot_pairs) atomicAdd(&A.stats[2], 1ULL);   // (pairs that came from a DP slot are counted by the slot allocator)
        }
        if (A.prof) {
            PMX_STAMP(W, 11);
            __syncthreads();
            if (lane < 23 && (lane < 12 || lane >= 16 || !A.dp_req_base)) atomicAdd(&A.prof[lane], W.prof_acc[lane]);   // (12..15: the DP service's own counters)
        }
    }
}

}  // namespace aln
}  // namespace pmx
