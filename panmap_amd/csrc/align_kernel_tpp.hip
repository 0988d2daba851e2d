// ALIGN stage, tier-0 kernel for gfx950: THREAD per read pair.  The per-pair bookkeeping of minimap2
// (sketch, seed lookup, chaining, region logic, CIGAR clean-up, mapq, pairing) is scalar, branchy code: run
// redundantly on 64 lanes it wastes the machine (measured: ~130k wave-instructions and 1.6M cycles per
// pair); run as 64 independent pairs per wave it is ordinary SIMT.  This kernel executes the very same
// sources as the host unit-test build (PMX_W = 1) and never runs a DP itself: a pair whose extension is not
// covered by the proved shortcuts (ksw_shortcut) posts the DP inputs as a request and aborts; the
// wave-per-request kernel k_align_dp_serve (align_kernel_t1.hip) computes it; the next round replays the
// pair with the result served from its slot (rounds repeat until no pair posts a request).  Pairs that
// exceed a work-buffer or DP-service capacity go to the wave-per-pair kernels through retry_list.
// Per-thread work arrays live in a private HBM slab.
#define PMX_THREAD_PER_PAIR 1
#include <hip/hip_runtime.h>

#include "align/aln_host.hpp"
#include "align_kernel.h"
#include "device/dev_util.hpp"

namespace pmx {
namespace aln {

// Per-thread memory: the interleaved arena is addressed through IPtr (aln_types.hpp) relative to the wave's
// slab named by A.tpp; `raw` is this lane's base into the wave's strided Reg region (struct Reg).
#ifndef PMX_TPP_OCC
#define PMX_TPP_OCC 4   // waves per SIMD the register allocation targets (latency-bound kernel: occupancy hides L2 round trips)
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PMX_TPP_OCC)))
k_align_reads_tpp(AlignArgs A) {
    static_assert(offsetof(AlignArgs, tpp) == 0, "IPtr::phys reads the arena from the start of the kernarg segment");
    const int64_t n_threads = (int64_t)gridDim.x * 64;
    uint8_t* raw = A.slow_base + (size_t)blockIdx.x * A.slow_stride + ((threadIdx.x & 63u) << 2);   // strided Reg region of this lane
    const int n_segs = A.paired ? 2 : 1;
    // minimizer window ring of this lane in LDS (dynamic LDS = 12 bytes x w x 64 lanes; 0 -> ring in the arena)
    extern __shared__ __attribute__((aligned(16))) uint8_t tpp_lds[];
    uint64_t* ring_x = nullptr;
    uint32_t* ring_y = nullptr;
    if (A.tpp_ring_w > 0) {
        ring_x = reinterpret_cast<uint64_t*>(tpp_lds) + (threadIdx.x & 63u);
        ring_y = reinterpret_cast<uint32_t*>(tpp_lds + (size_t)A.tpp_ring_w * 64 * 8) + (threadIdx.x & 63u);
    }

    const int lane = (int)(threadIdx.x & 63u);
    // every lane of the wave runs the same number of iterations (the CIGAR arena is claimed once per wave)
    // (tried, round 3: a launch too small to fill the chip spread over more waves, eight pairs on each -- the 27.8k bails of a
    // 10M-read batch: align stage 25.5 -> 26.4 ms, real reads 22.5 -> 21.7 ms; not kept)
    for (int64_t it0 = (int64_t)blockIdx.x * 64; it0 < A.n_items; it0 += n_threads) {
        const int64_t it = it0 + lane;
        int64_t item = -1, slot = -1;
        bool emit = false;
        Work W;
        if (it < A.n_items) do {   // `break` = this lane emits no record in this pass
        if (A.dp_round == 0) item = A.pair_perm ? (int64_t)A.pair_perm[it] : it;
        else {
            slot = A.worklist ? (int64_t)A.worklist[it] : it;
            item = (int64_t)A.dp_slot_pairs[slot];
            if (item == 0xffffffffLL) break;
        }
        bind_work(W, A.layout, nullptr, nullptr, raw);
        W.n_segs = n_segs;
        W.mv_ready = 0;
        W.sk_lds_x = ring_x;
        W.sk_lds_y = ring_y;
        W.prof = A.prof;   // diagnostic runs: lane 0's stamps are the wave's phase timeline
        if (A.prof) { W.prof_t = (unsigned long long)clock64(); for (int k = 0; k < 24; ++k) W.prof_acc[k] = 0; }
        W.dp_req_base = A.dp_req_base;
        W.dp_slot_cap = A.dp_slot_cap;
        W.dp_slot = slot;
        if (slot >= 0) {
            W.dp_res = A.dp_res_base + (size_t)slot * PMX_DP_MAX_CALLS;
            W.dp_n_cached = (int)A.dp_ncached[slot];
        } else W.dp_slot_ctr = A.dp_count;
        W.dp_post_end = W.dp_n_cached;
        W.status_pre = 0;
        bool too_long = false;
        for (int s = 0; s < n_segs; ++s) {
            const int64_t r = A.paired ? 2 * item + s : item;
            const int64_t len = A.off[r + 1] - A.off[r];
            if (len > A.layout.caps.max_qlen) too_long = true;
            W.qlen[s] = (int)len;
        }
        if (!too_long) {
            for (int s = 0; s < n_segs; ++s) {
                const int64_t r = A.paired ? 2 * item + s : item;
                const int len = W.qlen[s];
                const uint64_t* rw = A.words + A.woff[r];
                const uint32_t* ra = A.amb + A.woff[r];
                // X = the read as stored (forward strand, or the reverse-strand array when the mate is
                // reverse-complemented on the fly); Y = complement of X read backwards.  Four bases per store.
                const bool rc = A.revcomp_mate2 && s == 1;
                Ptr<uint8_t> X = rc ? W.qseq[s][1] : W.qseq[s][0];
                Ptr<uint8_t> Y = rc ? W.qseq[s][0] : W.qseq[s][1];
                Ptr<uint32_t> X4 = ptr_cast<uint32_t>(X), Y4 = ptr_cast<uint32_t>(Y);
                // four bases at a time: one byte of the packed word expands to one 32-bit word of base codes (the
                // ambiguity nibble is almost always zero), and the reverse complement is produced from the packed
                // words as well (not by reading X back from the arena): byte-reversed, 3 - code
                auto expand4 = [](uint32_t b) { return (b & 3u) | (b & 0xcu) << 6 | (b & 0x30u) << 12 | (b & 0xc0u) << 18; };
                auto code_of = [](uint32_t code, uint32_t am) { return am ? (code == 3u ? 3u : 4u) : code; };
                uint64_t cw = 0;
                uint32_t ca = 0;
                const int n4 = len >> 2;
                for (int g = 0; g < n4; ++g) {
                    if ((g & 7) == 0) { cw = rw[g >> 3]; ca = ra[g >> 3]; }
                    const uint32_t b = (uint32_t)cw & 0xffu, am4 = ca & 0xfu;
                    cw >>= 8; ca >>= 4;
                    uint32_t wd = expand4(b);
                    if (am4) {
                        wd = 0;
                        for (int k = 0; k < 4; ++k) wd |= code_of(b >> (2 * k) & 3u, am4 >> k & 1u) << (8 * k);
                    }
                    X4[g] = wd;
                }
                for (int i = n4 << 2; i < len; ++i) {
                    const uint32_t code = (uint32_t)(rw[i >> 5] >> (2 * (i & 31))) & 3u, am = ra[i >> 5] >> (i & 31) & 1u;
                    X[i] = (uint8_t)code_of(code, am);
                }
                int cur = -1;
                uint64_t w64 = 0;
                uint32_t a32 = 0;
                for (int g = 0; g < n4; ++g) {   // Y[4g .. 4g+3] = complement of X[len-1-4g .. len-4-4g]
                    const int hi = len - 1 - 4 * g, lo = hi - 3;
                    uint32_t wd;
                    if ((lo >> 5) == (hi >> 5)) {
                        if ((hi >> 5) != cur) { cur = hi >> 5; w64 = rw[cur]; a32 = ra[cur]; }
                        const uint32_t b = (uint32_t)(w64 >> (2 * (lo & 31))) & 0xffu, am4 = a32 >> (lo & 31) & 0xfu;
                        if (!am4) wd = 0x03030303u - __builtin_bswap32(expand4(b));
                        else {
                            wd = 0;
                            for (int k = 0; k < 4; ++k) {   // byte k of Y = position hi - k = bit pair (3 - k) of b
                                const uint32_t c = code_of(b >> (2 * (3 - k)) & 3u, am4 >> (3 - k) & 1u);
                                wd |= (c < 4u ? 3u - c : 4u) << (8 * k);
                            }
                        }
                    } else {   // the four bases straddle two packed words
                        wd = 0;
                        for (int k = 0; k < 4; ++k) {
                            const int pp = hi - k;
                            const uint32_t c = code_of((uint32_t)(rw[pp >> 5] >> (2 * (pp & 31))) & 3u, ra[pp >> 5] >> (pp & 31) & 1u);
                            wd |= (c < 4u ? 3u - c : 4u) << (8 * k);
                        }
                    }
                    Y4[g] = wd;
                }
                for (int j = n4 << 2; j < len; ++j) {
                    const int pp = len - 1 - j;
                    const uint32_t c = code_of((uint32_t)(rw[pp >> 5] >> (2 * (pp & 31))) & 3u, ra[pp >> 5] >> (pp & 31) & 1u);
                    Y[j] = (uint8_t)(c < 4u ? 3u - c : 4u);
                }
            }
            PMX_STAMP(W, 0);
            map_frag(W, A.opt, A.ri);
        } else {
            W.status |= PMX_ST_OVERFLOW;
        }
        // after a posted DP request the pass ran on neutral results: only what was known before it counts
        if (W.status & PMX_ST_NEED_DP) W.status = W.status_pre | PMX_ST_NEED_DP;
        if (W.status & (PMX_ST_OVERFLOW | PMX_ST_NEED_WAVE)) {
            A.retry_list[atomicAdd(A.retry_count, 1ULL)] = (uint32_t)item;
            if (A.dp_round == 0 && W.dp_slot >= 0) A.dp_slot_pairs[W.dp_slot] = 0xffffffffu;   // slot taken, pair gone
            break;
        }
        if (W.status & PMX_ST_NEED_DP) {
            A.dp_ncached[W.dp_slot] = (uint32_t)W.dp_post_end;   // what the next pass finds served
            if (A.dp_round == 0) {
                A.dp_slot_pairs[W.dp_slot] = (uint32_t)item;
                if (A.mv_handover && (uint64_t)W.dp_slot < (uint64_t)A.mv_slots && W.n_mv < (int)A.mv_stride) {   // see AlignArgs
                    A128* dst = A.mv_handover + (size_t)W.dp_slot * A.mv_stride;
                    A128 hd;
                    hd.x = (uint64_t)W.n_mv; hd.y = (uint64_t)A.mv_epoch << 32 | (uint32_t)item;
                    dst[0] = hd;
                    Ptr<A128> mvp = W.mv;
                    for (int j = 0; j < W.n_mv; ++j) dst[1 + j] = mvp[j];
                }
            } else A.dp_next_list[atomicAdd(A.dp_count, 1ULL)] = (uint32_t)slot;
            break;
        }
        emit = true;
        } while (0);
        // CIGAR arena: one atomic per wave (a returning atomic per mate on this single word serialised the lanes)
        bool mapped = false;
        uint32_t need[2] = {0, 0};
        if (emit) {
            mapped = frag_is_mapped(W, A.paired);
            for (int s = 0; s < n_segs; ++s)
                if (mapped && W.regs[s][0].has_p) need[s] = W.regs[s][0].n_cigar;
        }
        const uint32_t mine = need[0] + need[1];
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
        uint64_t coff = wave_base + (incl - mine);
        if (emit)
        for (int s = 0; s < n_segs; ++s) {
            const int64_t r = A.paired ? 2 * item + s : item;
            AlnRecord rec;
            memset(&rec, 0, sizeof(rec));
            rec.flags = (uint16_t)(W.status & 3u);
            if (mapped) {
                rec.mapped = 1;
                const Reg& g = W.regs[s][0];
                if (g.has_p) {
                    rec.flags |= PMX_REC_HAS_ALN;
                    rec.rs = g.rs; rec.re = g.re; rec.qs = g.qs; rec.qe = g.qe;
                    rec.mapq = g.mapq; rec.rev = g.rev; rec.proper_frag = g.proper_frag;
                    rec.n_cigar = (uint16_t)g.n_cigar;
                    rec.score = g.dp_max;
                    rec.cigar_off = (uint32_t)coff;
                    if (coff + g.n_cigar <= A.cigar_cap && g.n_cigar <= 0xffffu) {   // (n_cigar is 16 bits wide in the record)
                        Ptr<const uint32_t> cg = reg_cigar(W, g);
                        for (uint32_t i = 0; i < g.n_cigar; ++i) A.cigars[coff + i] = cg[i];
                    } else {
                        rec.flags |= PMX_REC_OVERFLOW;
                        rec.n_cigar = 0;
                    }
                    coff += g.n_cigar;
                }
            }
            A.records[r] = rec;
            if (A.edits) A.edits[r] = read_errors(W, s);
        }
        if (A.prof && emit) {
            PMX_STAMP(W, 11);
            if ((threadIdx.x & 63) == 0)
                for (int k = 0; k < 24; ++k)
                    if (k < 12 || k >= 16) atomicAdd(&A.prof[k], W.prof_acc[k]);
        }
    }
}

}  // namespace aln
}  // namespace pmx
