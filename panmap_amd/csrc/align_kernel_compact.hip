// ALIGN stage, compact tier kernel for gfx950: THREAD per read pair, the pair's whole work state in LDS (336 bytes,
// word-interleaved across the wave: 21 KB per wave, seven waves per CU; 448 bytes / five waves when the reference is
// longer than 32,767 bases) and registers -- align/aln_compact.hpp.
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

template <class PT>
__device__ __forceinline__ void align_compact_body(const AlignArgs& A) {
    extern __shared__ __attribute__((aligned(16))) uint32_t c_lds[];
    const int lane = (int)(threadIdx.x & 63u);
    CMemT<PT> m;
    // gap-penalty tables behind the lanes' work memory (see CPenTab): 1152 bytes per wave
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
    static_assert(CMemT<PT>::kWords == (sizeof(PT) == 2 ? PMX_C_LANE_WORDS16 : PMX_C_LANE_WORDS32), "LDS size the host launches with");
    m.base = (c_u32*)c_lds + lane;
    const int64_t n_threads = (int64_t)gridDim.x * 64;
    // every lane of the wave runs the same number of iterations (arena and bail-list slots are claimed once per wave)
    for (int64_t it0 = (int64_t)blockIdx.x * 64; it0 < A.n_items; it0 += n_threads) {
        const int64_t it = it0 + lane;
        int64_t item = -1;
        int rc = PMX_C_DONE;
        CResult res;
        res.mapped = 0;
        if (it < A.n_items) {
            item = A.pair_perm ? (int64_t)A.pair_perm[it] : it;
            CRead rd[2];
            const uint32_t* amb[2];
            for (int s = 0; s < 2; ++s) {
                const int64_t r = 2 * item + s;
                const int64_t len = A.off[r + 1] - A.off[r];
                rd[s].w = A.words + A.woff[r];
                rd[s].len = len > 0x7fffffff ? 0x7fffffff : (int)len;
                rd[s].flip = A.revcomp_mate2 && s == 1;
                amb[s] = A.amb + A.woff[r];
            }
            unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            rc = compact_map_pair(m, A.opt, A.ri, rd, amb, res, tab, A.prof ? pacc : nullptr, A.edits != nullptr);
            if (A.prof && lane == 0)   // lane 0's stamps are the wave's phase timeline (diagnostic runs: PMX_ALIGN_PROF)
                for (int k = 0; k < 8; ++k) atomicAdd(&A.prof[k], pacc[k]);
        }
        // bail list: one atomic per wave
        const bool bail = item >= 0 && rc != PMX_C_DONE;
        const unsigned long long bmask = __ballot(bail);
        if (bmask) {
            unsigned long long bbase = 0;
            if (lane == 0) bbase = atomicAdd(A.retry_count, (unsigned long long)__popcll(bmask));
            bbase = __shfl(bbase, 0);
            if (bail) A.retry_list[bbase + __popcll(bmask & ((1ULL << lane) - 1ULL))] = (uint32_t)item;
        }
        // CIGAR arena: one word per mapped mate, one atomic per wave
        const bool done = item >= 0 && rc == PMX_C_DONE;
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
}

__global__ void __launch_bounds__(64) k_align_compact16(AlignArgs A) { align_compact_body<uint16_t>(A); }
__global__ void __launch_bounds__(64) k_align_compact32(AlignArgs A) { align_compact_body<uint32_t>(A); }

}  // namespace aln
}  // namespace pmx
