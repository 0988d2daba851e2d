// ALIGN stage, compact tier kernel for gfx950: THREAD per read pair, the pair's whole work state in LDS (the first form of
// the two-kernel chain kernel: 48 anchors, 288 bytes, word-interleaved across the wave: 19.6 KB per wave with the penalty
// tables, eight waves per CU -- 384 bytes / six waves when the reference is longer than 32,767 bases; the second form and
// the fused kernels: all 56 anchors the hand-over carries, 336 / 448 bytes, seven / five waves) and registers --
// align/aln_compact.hpp.
// It takes every pair of the batch first.  A pair it finishes has its records written here; a pair outside the tier's
// envelope is appended to the bail list and run by the general thread-per-pair kernel (align_kernel_tpp.hip) and its
// DP service.  HBM traffic per pair: the packed read words in (38 B per 150 bp read, + the ambiguity words), index /
// reference probes (L2 resident), two 32-byte records and two CIGAR words out.
#define PMX_THREAD_PER_PAIR 1
#include <hip/hip_runtime.h>

#include "align/aln_compact.hpp"
#include "align/aln_host.hpp"
#include "align_kernel.h"
#include "device/dev_util.hpp"

namespace pmx {
namespace aln {

// The pair of launch position `it` (both kernels of the two-kernel form walk the positions the same way)
__device__ __forceinline__ int64_t compact_item(const AlignArgs& A, int64_t it, CRead* rd, const uint32_t** amb) {
    const int64_t item = A.pair_perm ? (int64_t)A.pair_perm[it] : it;
    if (A.recs) {
        // the mates' records lie side by side: one aligned 128-byte stretch per pair instead of pieces of two arrays and
        // four offset look-ups (the pairs are visited in locality order, i.e. scattered)
        const uint8_t* rec = A.recs + (size_t)item * 128;
        for (int s = 0; s < 2; ++s) {
            rd[s].w = reinterpret_cast<const uint64_t*>(rec + 64 * s);
            rd[s].len = (int)*reinterpret_cast<const uint32_t*>(rec + 64 * s + 60);
            rd[s].flip = A.revcomp_mate2 && s == 1;
            amb[s] = reinterpret_cast<const uint32_t*>(rec + 64 * s + 40);
        }
        return item;
    }
    for (int s = 0; s < 2; ++s) {
        const int64_t r = 2 * item + s;
        const int64_t len = A.off[r + 1] - A.off[r];
        rd[s].w = A.words + A.woff[r];
        rd[s].len = len > 0x7fffffff ? 0x7fffffff : (int)len;
        rd[s].flip = A.revcomp_mate2 && s == 1;
        amb[s] = A.amb + A.woff[r];
    }
    return item;
}

// First kernel of the two-kernel form: sketch + index probes of every pair, thread per pair.  The only LDS is the
// minimizer queue (112 bytes per pair, 7 KB per wave), so the CU holds as many waves as the registers allow (the fused
// kernel: seven, by its 21 KB of work state per wave) -- this part is pure integer arithmetic plus a few probes.
// Seeds go to the pair's hand-over words (CSeedOutT), the count (or PMX_C_NSEED_BAIL) to cseed_n[position].
template <class PT>
__device__ __forceinline__ void compact_seeds_body(const AlignArgs& A) {
    extern __shared__ __attribute__((aligned(16))) uint32_t c_lds[];
    const int lane = (int)(threadIdx.x & 63u);
    CSeedOutT<PT> so;
    so.q = (c_u32*)c_lds + lane;
    so.st = (c_u32*)c_lds + PMX_C_SEEDQ * 2 * 64 + lane;   // eight staging words per lane behind the queues
    const int64_t n_threads = (int64_t)gridDim.x * 64;
    for (int64_t it0 = (int64_t)blockIdx.x * 64; it0 < A.n_items; it0 += n_threads) {   // (uniform trip count: the drains are wave-wide)
        const int64_t it = it0 + lane;
        so.out = (c_g32*)(A.cseeds + (size_t)it * CSeedOutT<PT>::kPairWords);
        int n_s = 0, n_s0 = 0, rc = PMX_C_DONE;
        if (it < A.n_items) {
            CRead rd[2];
            const uint32_t* amb[2];
            compact_item(A, it, rd, amb);
            unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            rc = compact_seed_pair(so, A.opt, A.ri, rd, amb, &n_s, &n_s0, pacc, A.prof != nullptr);
            if (A.prof && lane == 0)
                for (int k = 0; k < 2; ++k) atomicAdd(&A.prof[k], pacc[k]);
            A.cseed_n[it] = (uint16_t)(rc == PMX_C_DONE ? (n_s | n_s0 << 8) : (int)PMX_C_NSEED_BAIL);
        }
    }
}
__global__ void __launch_bounds__(64) k_compact_seeds16(AlignArgs A) { compact_seeds_body<uint16_t>(A); }
__global__ void __launch_bounds__(64) k_compact_seeds32(AlignArgs A) { compact_seeds_body<uint32_t>(A); }

// The pairs the seeds kernel gave up on, as a list of pair ids in launch order (one atomic per wave)
__global__ void __launch_bounds__(256) k_compact_list_seed_bails(AlignArgs A) {
    const int lane = (int)(threadIdx.x & 63u);
    for (int64_t it0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) & ~63LL; it0 < A.n_items; it0 += (int64_t)gridDim.x * blockDim.x) {
        const int64_t it = it0 + lane;
        const bool hit = it < A.n_items && A.cseed_n[it] == PMX_C_NSEED_BAIL;
        const unsigned long long mask = __ballot(hit);
        if (!mask) continue;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(A.early_count, (unsigned long long)__popcll(mask));
        base = __shfl(base, 0);
        if (hit) A.early_list[base + __popcll(mask & ((1ULL << lane) - 1ULL))] = A.pair_perm ? A.pair_perm[it] : (uint32_t)it;
    }
}

// What a pair of either form leaves behind: an entry of the bail list (one atomic per wave), or its two records and CIGAR
// words (one arena atomic per wave).  Called by every lane of the wave.
__device__ __forceinline__ void compact_emit(const AlignArgs& A, int lane, int64_t item, bool bail, bool done, const CResult& res) {
    const unsigned long long bmask = __ballot(bail);
    if (bmask) {
        unsigned long long bbase = 0;
        if (lane == 0) bbase = atomicAdd(A.retry_count, (unsigned long long)__popcll(bmask));
        bbase = __shfl(bbase, 0);
        if (bail) A.retry_list[bbase + __popcll(bmask & ((1ULL << lane) - 1ULL))] = (uint32_t)item;
    }
    // CIGAR arena: one word per mapped mate, one atomic per wave
    const uint32_t mine = done && res.mapped ? 2u : 0u;
    uint32_t incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    const uint32_t wave_total = __shfl(incl, 63);
    unsigned long long wave_base = 0;
    if (wave_total) {
        if (lane == 0) wave_base = atomicAdd(A.cigar_used, (unsigned long long)wave_total);
        wave_base = __shfl(wave_base, 0);
    }
    if (done) {
        uint64_t coff = wave_base + (incl - mine);
        for (int s = 0; s < 2; ++s) {
            AlnRecord rec;
            memset(&rec, 0, sizeof(rec));
            if (res.mapped) {
                const CMate& t = res.m[s];
                rec.mapped = 1;
                rec.flags = PMX_REC_HAS_ALN;
                rec.rs = t.rs; rec.re = t.re; rec.qs = t.qs; rec.qe = t.qe;
                rec.mapq = t.mapq; rec.rev = t.rev; rec.proper_frag = t.proper_frag;
                rec.n_cigar = 1;
                rec.score = t.dp_max;
                rec.cigar_off = (uint32_t)coff;
                if (coff < A.cigar_cap) A.cigars[coff] = t.cigar;
                else { rec.flags |= PMX_REC_OVERFLOW; rec.n_cigar = 0; }
                ++coff;
            }
            A.records[2 * item + s] = rec;
            if (A.edits) A.edits[2 * item + s] = res.edit[s];
        }
    }
}

template <class PT, bool PRESEEDED>
__device__ __forceinline__ void align_compact_body(const AlignArgs& A) {
    extern __shared__ __attribute__((aligned(16))) uint32_t c_lds[];
    const int lane = (int)(threadIdx.x & 63u);
    typedef CMemT<PT, PRESEEDED ? PMX_C_CAP1 : PMX_C_CAP> MT;   // (the seeds kernel's queue overlay needs the full block: fused form)
    MT m;
    // gap-penalty tables behind the lanes' work memory (see CPenTab): 1152 bytes per wave
    c_u8* pen = (c_u8*)((c_u32*)c_lds + MT::kWords * 64);
    CPenTab tab;
    tab.same = pen; tab.diff = pen + PMX_C_PEN_SAME;
    for (int dd = lane; dd < PMX_C_PEN_DIFF; dd += 64) {
        int ps, pd;
        c_pen_values(A.opt.chn_pen_gap, dd, &ps, &pd);
        if (dd < PMX_C_PEN_SAME) pen[dd] = (uint8_t)(ps < 255 ? ps : 255);
        pen[PMX_C_PEN_SAME + dd] = (uint8_t)(pd < 255 ? pd : 255);
    }
    __syncthreads();
    static_assert(MT::kWords == (PRESEEDED ? (sizeof(PT) == 2 ? PMX_C_LANE_WORDS16_1 : PMX_C_LANE_WORDS32_1) : (sizeof(PT) == 2 ? PMX_C_LANE_WORDS16 : PMX_C_LANE_WORDS32)),
                  "LDS size the host launches with");
    m.base = (c_u32*)c_lds + lane;
    const int64_t n_threads = (int64_t)gridDim.x * 64;
    // every lane of the wave runs the same number of iterations (arena and bail-list slots are claimed once per wave)
    for (int64_t it0 = (int64_t)blockIdx.x * 64; it0 < A.n_items; it0 += n_threads) {
        const int64_t it = it0 + lane;
        int64_t item = -1;
        int rc = PMX_C_DONE;
        CResult res;
        res.mapped = 0;
        CRead rd[2];
        const uint32_t* amb[2];
        unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (PRESEEDED) {
            // two-kernel form: the pairs' seeds wait in the hand-over words; a wave copies seed i of its 64 pairs with
            // one contiguous load per word, four seeds requested together
            int n_s = 0, n_s0 = 0;
            if (it < A.n_items) {
                item = compact_item(A, it, rd, amb);
                const uint32_t c = A.cseed_n[it];
                if (c == PMX_C_NSEED_BAIL) {
                    if (A.seed_bails_listed) item = -1;   // already with the general tiers (k_compact_list_seed_bails)
                    else rc = PMX_C_BAIL;
                } else { n_s = (int)(c & 0xffu); n_s0 = (int)(c >> 8); }
            }
            CSeedOutT<PT> so;
            so.q = nullptr; so.st = nullptr;
            so.out = (c_g32*)(A.cseeds + (size_t)it * CSeedOutT<PT>::kPairWords);
            for (int i0 = 0; __ballot(i0 < n_s) != 0ULL; i0 += 4) {
                uint32_t x[4] = {0, 0, 0, 0}, y[4] = {0, 0, 0, 0};
                if (i0 < n_s) so.get4(i0, x, y);
#pragma unroll
                for (int b = 0; b < 4; ++b) if (i0 + b < n_s) m.setSeed(i0 + b, x[b], y[b]);
            }
            if (item >= 0 && rc == PMX_C_DONE) {
                rc = compact_chain_pair(m, A.opt, A.ri, rd, n_s, n_s0, res, tab, pacc, A.edits != nullptr, A.prof != nullptr);
                if (A.prof && lane == 0)
                    for (int k = 2; k < 8; ++k) atomicAdd(&A.prof[k], pacc[k]);
            }
        } else if (it < A.n_items) {
            item = compact_item(A, it, rd, amb);
            rc = compact_map_pair(m, A.opt, A.ri, rd, amb, res, tab, pacc, A.edits != nullptr, A.prof != nullptr);
            if (A.prof && lane == 0)   // lane 0's stamps are the wave's phase timeline (diagnostic runs: PMX_ALIGN_PROF)
                for (int k = 0; k < 8; ++k) atomicAdd(&A.prof[k], pacc[k]);
        }
        // a pair that leaves after its seeds were made: the second form (several regions per mate) takes it from its hand-over words
        bool bail = item >= 0 && rc != PMX_C_DONE;
        if (PRESEEDED && A.multi_list) {
            const bool again = bail && A.cseed_n[it] != PMX_C_NSEED_BAIL;
            const unsigned long long mmask = __ballot(again);
            if (mmask) {
                unsigned long long mbase = 0;
                if (lane == 0) mbase = atomicAdd(A.multi_count, (unsigned long long)__popcll(mmask));
                mbase = __shfl(mbase, 0);
                if (again) A.multi_list[mbase + __popcll(mmask & ((1ULL << lane) - 1ULL))] = (uint32_t)it;
            }
            bail = bail && !again;
        }
        compact_emit(A, lane, item, bail, item >= 0 && rc == PMX_C_DONE, res);
    }
}

// Second form of the compact tier: the pairs of multi_list (launch positions), seeds from their hand-over words, up to
// four fragment chains and several regions per mate (align/aln_compact_multi.hpp).  The list's length is only known on
// the device: a resident grid strides over it, waves beyond its end leave at once.
template <class PT>
__device__ __forceinline__ void align_compact_multi_body(const AlignArgs& A) {
    extern __shared__ __attribute__((aligned(16))) uint32_t c_lds[];
    const int lane = (int)(threadIdx.x & 63u);
    CMemT<PT> m;
    c_u8* pen = (c_u8*)((c_u32*)c_lds + CMemT<PT>::kWords * 64);
    CPenTab tab;
    tab.same = pen; tab.diff = pen + PMX_C_PEN_SAME;
    for (int dd = lane; dd < PMX_C_PEN_DIFF; dd += 64) {
        int ps, pd;
        c_pen_values(A.opt.chn_pen_gap, dd, &ps, &pd);
        if (dd < PMX_C_PEN_SAME) pen[dd] = (uint8_t)(ps < 255 ? ps : 255);
        pen[PMX_C_PEN_SAME + dd] = (uint8_t)(pd < 255 ? pd : 255);
    }
    __syncthreads();
    m.base = (c_u32*)c_lds + lane;
    // the wave's region records and index words (HBM, interleaved across the lanes; see SReg)
    uint32_t* const wsb = A.multi_ws + (size_t)blockIdx.x * ((size_t)PMX_CM_WS_WORDS * 64);
    const SWork mw{reinterpret_cast<SReg*>(wsb + lane), wsb + (size_t)PMX_CM_SREGS * 24 * 64 + lane};
    const int64_t n_list = (int64_t)*A.multi_count;
    const int64_t n_threads = (int64_t)gridDim.x * 64;
    for (int64_t e0 = (int64_t)blockIdx.x * 64; e0 < n_list; e0 += n_threads) {
        const int64_t e = e0 + lane;
        int64_t item = -1;
        int rc = PMX_C_DONE;
        CResult res;
        res.mapped = 0;
        if (e < n_list) {
            const int64_t it = (int64_t)A.multi_list[e];
            CRead rd[2];
            const uint32_t* amb[2];
            item = compact_item(A, it, rd, amb);
            const uint32_t c = A.cseed_n[it];
            const int n_s = (int)(c & 0xffu), n_s0 = (int)(c >> 8);
            CSeedOutT<PT> so;
            so.q = nullptr; so.st = nullptr;
            so.out = (c_g32*)(A.cseeds + (size_t)it * CSeedOutT<PT>::kPairWords);
            for (int i0 = 0; i0 < n_s; i0 += 4) {
                uint32_t x[4], y[4];
                so.get4(i0, x, y);
#pragma unroll
                for (int b = 0; b < 4; ++b) if (i0 + b < n_s) m.setSeed(i0 + b, x[b], y[b]);
            }
            rc = compact_chain_pair<true>(m, A.opt, A.ri, rd, n_s, n_s0, res, tab, nullptr, A.edits != nullptr, false, &mw);
        }
        const bool done = item >= 0 && rc == PMX_C_DONE;
        const unsigned long long dmask = __ballot(done);
        if (dmask && lane == 0) atomicAdd(A.multi_count + 1, (unsigned long long)__popcll(dmask));
        compact_emit(A, lane, item, item >= 0 && rc != PMX_C_DONE, done, res);
    }
}

__global__ void __launch_bounds__(64) k_align_compact16(AlignArgs A) { align_compact_body<uint16_t, true>(A); }
__global__ void __launch_bounds__(64) k_align_compact32(AlignArgs A) { align_compact_body<uint32_t, true>(A); }
__global__ void __launch_bounds__(64) k_align_compact16_fused(AlignArgs A) { align_compact_body<uint16_t, false>(A); }
__global__ void __launch_bounds__(64) k_align_compact32_fused(AlignArgs A) { align_compact_body<uint32_t, false>(A); }
__global__ void __launch_bounds__(64) k_align_compact16_multi(AlignArgs A) { align_compact_multi_body<uint16_t>(A); }
__global__ void __launch_bounds__(64) k_align_compact32_multi(AlignArgs A) { align_compact_multi_body<uint32_t>(A); }

}  // namespace aln
}  // namespace pmx
