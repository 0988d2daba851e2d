// Sorting primitives with the reference's exact (unstable, but deterministic) behaviour.
// minimap2 sorts everything with KRADIX_SORT (ksort.h:98-153): insertion sort up to 64 elements, else
// an in-place MSD byte radix sort that falls back to insertion sort on buckets <= 64.  Equal keys keep
// whatever order these procedures leave them in, and that order leaks into results (chain order,
// hit order), so the procedures are restated here step for step.
#pragma once
#include "aln_types.hpp"

namespace pmx {
namespace aln {

struct KeyX {
    PMX_HD static uint64_t key(const A128& a) { return a.x; }
};
struct KeyU64 {
    PMX_HD static uint64_t key(const uint64_t& a) { return a; }
};

template <class T, class K>
PMX_HD void rs_insertsort(Ptr<T> beg, Ptr<T> end) {
    for (Ptr<T> i = beg + 1; i < end; ++i) {
        if (K::key(*i) < K::key(*(i - 1))) {
            Ptr<T> j = i;
            const T tmp = *i;
            for (; j > beg && K::key(tmp) < K::key(*(j - 1)); --j) *j = *(j - 1);
            *j = tmp;
        }
    }
}

#if PMX_W > 1
// n <= 64: one element per lane, rank = number of elements that precede it in a STABLE ascending order.
// The reference's insertion sort is stable, so the result is identical; the n^2/4 dependent LDS moves of
// the insertion sort become n scalar broadcasts.
template <class T, class K>
__device__ __forceinline__ void wave_rank_sort(T* beg, int n) {
    const int lane = lane_id();
    T val;
    uint64_t key = 0;
    if (lane < n) { val = beg[lane]; key = K::key(val); }
    const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
    int rank = 0;
    for (int j = 0; j < n; ++j) {
        const uint32_t jl = (uint32_t)__builtin_amdgcn_readlane((int)klo, j), jh = (uint32_t)__builtin_amdgcn_readlane((int)khi, j);
        const uint64_t kj = (uint64_t)jh << 32 | jl;
        rank += (kj < key || (kj == key && j < lane)) ? 1 : 0;
    }
    wave_sync();
    if (lane < n) beg[rank] = val;
    wave_sync();
}
#endif

// one level of the American-flag pass on byte `s/8`; iterative over an explicit stack of pending
// sub-ranges (the reference recurses; the visiting order of disjoint buckets does not matter)
template <class T, class K>
PMX_HDN void rs_sort_level(Ptr<T> beg, Ptr<T> end, int s, Ptr<T>* stk_b, Ptr<T>* stk_e, int* stk_s, int& sp, int stk_cap, uint32_t* overflow) {
    // bucket boundaries: 256 (begin,end) pairs kept as offsets
    int32_t bb[256], be[256];
    for (int k = 0; k < 256; ++k) bb[k] = be[k] = 0;
    for (Ptr<T> i = beg; i != end; ++i) ++be[(K::key(*i) >> s) & 255];
    for (int k = 1; k < 256; ++k) { be[k] += be[k - 1]; bb[k] = be[k - 1]; }
    for (int k = 0; k < 256;) {
        if (bb[k] != be[k]) {
            int l = (int)((K::key(beg[bb[k]]) >> s) & 255);
            if (l != k) {
                T tmp = beg[bb[k]], swp;
                do {
                    swp = tmp;
                    tmp = beg[bb[l]];
                    beg[bb[l]++] = swp;
                    l = (int)((K::key(tmp) >> s) & 255);
                } while (l != k);
                beg[bb[k]++] = tmp;
            } else ++bb[k];
        } else ++k;
    }
    bb[0] = 0;
    for (int k = 1; k < 256; ++k) bb[k] = be[k - 1];
    if (s) {
        const int s2 = s > 8 ? s - 8 : 0;
        for (int k = 0; k < 256; ++k) {
            const int n = be[k] - bb[k];
            if (n > 64) {
                if (sp < stk_cap) { stk_b[sp] = beg + bb[k]; stk_e[sp] = beg + be[k]; stk_s[sp] = s2; ++sp; }
                else *overflow |= PMX_ST_OVERFLOW;
            } else if (n > 1) {
#if PMX_W > 1
                // (the reference's insertion sort is stable: the rank sort over the lanes leaves the same order, without the
                //  n^2/4 dependent moves -- in the wave's HBM slab for a long read's ~1,800 chain ends, where the thirty-odd
                //  buckets of a level were most of the chaining stage's time)
                wave_rank_sort<T, K>(beg + bb[k], n);
#else
                rs_insertsort<T, K>(beg + bb[k], beg + be[k]);
#endif
            }
        }
    }
}

template <class T, class K>
PMX_HDN void radix_sort(Ptr<T> beg, Ptr<T> end, uint32_t* status) {
    if (end - beg <= 64) {
#if PMX_W > 1
        if (end - beg > 1) wave_rank_sort<T, K>(beg, (int)(end - beg));
#else
        rs_insertsort<T, K>(beg, end);
#endif
        return;
    }
    Ptr<T> stk_b[64];
    Ptr<T> stk_e[64];
    int stk_s[64];
    int sp = 0;
    int s0 = 56;
#if PMX_W > 1
    // A level at which every key holds the same byte moves nothing and hands the whole range to the next level (one bucket
    // of more than 64 elements): start at the first byte in which the keys differ.  (Chain scores, positions within one
    // reference: the top five or six bytes are equal, and every such level was a counting pass plus a walk over the range
    // in the wave's HBM slab -- 13.7 M of a 10 kb read's 86 M cycles went into the sort of its ~1,800 chain ends.)
    {
        uint64_t all_or = 0, all_and = ~0ULL;
        for (int64_t i = lane_id(); i < (int64_t)(end - beg); i += PMX_W) { const uint64_t kx = K::key(beg[i]); all_or |= kx; all_and &= kx; }
        for (int o = PMX_W / 2; o > 0; o >>= 1) { all_or |= __shfl_xor(all_or, o); all_and &= __shfl_xor(all_and, o); }
        const uint64_t diff = all_or ^ all_and;   // bits in which some two keys differ
        while (s0 > 0 && ((diff >> s0) & 255ULL) == 0ULL) s0 -= 8;
    }
#endif
    stk_b[0] = beg; stk_e[0] = end; stk_s[0] = s0; sp = 1;
    while (sp > 0) {
        --sp;
        Ptr<T> b = stk_b[sp];
        Ptr<T> e = stk_e[sp];
        const int s = stk_s[sp];
        rs_sort_level<T, K>(b, e, s, stk_b, stk_e, stk_s, sp, 64, status);
    }
}

PMX_HD void radix_sort_128x(Ptr<A128> beg, Ptr<A128> end, uint32_t* status) { radix_sort<A128, KeyX>(beg, end, status); }
PMX_HD void radix_sort_64(Ptr<uint64_t> beg, Ptr<uint64_t> end, uint32_t* status) { radix_sort<uint64_t, KeyU64>(beg, end, status); }

}  // namespace aln
}  // namespace pmx
