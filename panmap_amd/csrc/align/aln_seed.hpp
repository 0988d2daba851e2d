// ALIGN stage, part 1: minimizer sketch, index lookup, seed filtering and the heap merge that turns
// minimizer occurrences into sorted anchors.
// Reference behaviour: sketch.c:28-143 (mm_sketch, hash64), index.c:81-99 (mm_idx_get), seed.c:28-131
// (mm_seed_collect_all, mm_seed_select, mm_collect_matches), map.c:59-166 (collect_minimizers,
// collect_seed_hits_heap).
#pragma once
#include "aln_sort.hpp"
#include "aln_types.hpp"

namespace pmx {
namespace aln {

// invertible integer hash restricted to 2k bits (sketch.c:28-38)
PMX_HD uint64_t mz_hash64(uint64_t key, uint64_t mask) {
    key = (~key + (key << 21)) & mask;
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8)) & mask;
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4)) & mask;
    key = key ^ key >> 28;
    key = (key + (key << 31)) & mask;
    return key;
}

// (w,k)-minimizers of one segment, appended to W.mv (no HPC).  seq holds nt4 codes (>=4 ambiguous).
// Output layout as the reference: x = hash<<8 | span, y = rid<<32 | lastPos<<1 | strand.
// the window ring of sketch_segment: w entries of (x, y)
struct RingMem {            // in the work arena (or LDS / host memory through plain pointers)
    Ptr<A128> buf;
    PMX_HD A128 get(int j) const { return buf[j]; }
    PMX_HD uint64_t getx(int j) const { return buf[j].x; }
    PMX_HD void set(int j, const A128& v) const { buf[j] = v; }
};
#ifdef PMX_INTERLEAVED
struct RingLds {            // thread-per-pair kernel: [slot][lane] in LDS, y stored as its low 32 bits
    uint64_t* x;
    uint32_t* y;
    uint64_t y_hi;
    __device__ __forceinline__ A128 get(int j) const {
        A128 v;
        v.x = x[j * 64];
        v.y = v.x == UINT64_MAX ? UINT64_MAX : (y_hi | y[j * 64]);
        return v;
    }
    __device__ __forceinline__ uint64_t getx(int j) const { return x[j * 64]; }
    __device__ __forceinline__ void set(int j, const A128& v) const { x[j * 64] = v.x; y[j * 64] = (uint32_t)v.y; }
};
#endif

template <class Ring>
PMX_HD void sketch_segment_t(Work& W, const Ring& buf, Ptr<const uint8_t> seq, int len, int w, int k, uint32_t rid) {
    const uint64_t shift1 = 2 * (uint64_t)(k - 1), mask = (1ULL << 2 * k) - 1;
    uint64_t kmer0 = 0, kmer1 = 0;
    int l = 0, buf_pos = 0, min_pos = 0, kmer_span = 0;
    PMX_LDS(&W); PMX_LDS(seq);
    Ptr<A128> mvp = W.mv; PMX_LDS(mvp);
    A128 mn;
    mn.x = mn.y = UINT64_MAX;
    {
        A128 inv;
        inv.x = inv.y = UINT64_MAX;
        for (int j = 0; j < w; ++j) buf.set(j, inv);
    }
#define PMX_MV_PUSH(val)                                              \
    do {                                                              \
        if (W.n_mv < W.caps.max_mini) mvp[W.n_mv] = (val);            \
        else W.status |= PMX_ST_OVERFLOW;                             \
        if (W.n_mv < W.caps.max_mini) ++W.n_mv;                       \
    } while (0)
    ByteReader seq_r(seq);
    for (int i = 0; i < len; ++i) {
        const int c = (int)seq_r[i];
        A128 info;
        info.x = info.y = UINT64_MAX;
        if (c < 4) {
            kmer_span = l + 1 < k ? l + 1 : k;
            kmer0 = (kmer0 << 2 | (uint64_t)c) & mask;
            kmer1 = (kmer1 >> 2) | (uint64_t)(3 ^ c) << shift1;
            if (kmer0 == kmer1) continue;   // strand-symmetric k-mer: skipped without advancing the window
            const int z = kmer0 < kmer1 ? 0 : 1;
            ++l;
            if (l >= k && kmer_span < 256) {
                info.x = mz_hash64(z ? kmer1 : kmer0, mask) << 8 | (uint64_t)kmer_span;
                info.y = (uint64_t)rid << 32 | (uint32_t)i << 1 | (uint32_t)z;
            }
        } else {
            l = 0;
            kmer_span = 0;
        }
        buf.set(buf_pos, info);
        if (l == w + k - 1 && mn.x != UINT64_MAX) {   // first full window: emit earlier identical k-mers
            for (int j = buf_pos + 1; j < w; ++j)
                { const A128 bj = buf.get(j); if (mn.x == bj.x && bj.y != mn.y) PMX_MV_PUSH(bj); }
            for (int j = 0; j < buf_pos; ++j)
                { const A128 bj = buf.get(j); if (mn.x == bj.x && bj.y != mn.y) PMX_MV_PUSH(bj); }
        }
        if (info.x <= mn.x) {                          // new minimum: flush the old one
            if (l >= w + k && mn.x != UINT64_MAX) PMX_MV_PUSH(mn);
            mn = info;
            min_pos = buf_pos;
        } else if (buf_pos == min_pos) {               // the minimum slid out of the window
            if (l >= w + k - 1 && mn.x != UINT64_MAX) PMX_MV_PUSH(mn);
            mn.x = UINT64_MAX;
            for (int j = buf_pos + 1; j < w; ++j)
                if (mn.x >= buf.getx(j)) { mn = buf.get(j); min_pos = j; }   // >= keeps the right-most
            for (int j = 0; j <= buf_pos; ++j)
                if (mn.x >= buf.getx(j)) { mn = buf.get(j); min_pos = j; }
            if (l >= w + k - 1 && mn.x != UINT64_MAX) {
                for (int j = buf_pos + 1; j < w; ++j)
                    { const A128 bj = buf.get(j); if (mn.x == bj.x && mn.y != bj.y) PMX_MV_PUSH(bj); }
                for (int j = 0; j <= buf_pos; ++j)
                    { const A128 bj = buf.get(j); if (mn.x == bj.x && mn.y != bj.y) PMX_MV_PUSH(bj); }
            }
        }
        if (++buf_pos == w) buf_pos = 0;
    }
    if (mn.x != UINT64_MAX) PMX_MV_PUSH(mn);
#undef PMX_MV_PUSH
}

// The same sketch for the scalar execution models (thread per pair, host) with the window ring in REGISTERS
// (w <= WMAX): sketch_segment_t re-scans the ring entry by entry whenever the minimum slides out of the window, and in
// a wave of 64 reads some lane needs that at nearly every base, so every base paid ~2w dependent ring reads.  Here
// the ring is an unrolled array (static indices; the dynamic slot is written by compare-select), the re-scan is a
// min / arg-max-age reduction over registers, and the "identical minimizer in the window" loops only run when a
// compare mask says there is one.  Entry order, tie rules and the emitted minimizers are exactly sketch_segment_t's:
//  * re-scan order is j = buf_pos+1 .. w-1, 0 .. buf_pos with '>=': among equal x the LAST in that order wins, i.e.
//    the largest age rank (j - buf_pos - 1) mod w;
//  * an entry is "another occurrence" iff its x equals the minimum's and it is not the minimum's slot (y holds the
//    position, unique per slot; invalid entries are excluded by mn.x != UINT64_MAX).
// BaseFn: int(int i) = nt4 code of base i (>= 4: ambiguous); PushFn: void(uint64_t x, uint64_t y) appends a minimizer
// (the caller owns the output list and its overflow handling).  Used by the scalar models of the general pipeline
// (sketch_segment_reg below: bases from the work arena, output to W.mv) and by the compact tier (aln_compact.hpp:
// bases straight from the packed read words, output to its LDS staging list).
template <int WMAX, class BaseFn, class PushFn>
PMX_HD void sketch_core(int len, int w, int k, uint64_t y_hi, BaseFn& base_at, PushFn& push, bool final_push = true) {
    const uint64_t shift1 = 2 * (uint64_t)(k - 1), mask = (1ULL << 2 * k) - 1;
    uint64_t kmer0 = 0, kmer1 = 0;
    int l = 0, buf_pos = 0, min_pos = 0, kmer_span = 0;
    A128 mn;
    mn.x = mn.y = UINT64_MAX;
    uint64_t xs[WMAX];
    uint32_t ys[WMAX];   // low 32 bits of y (the high half is rid; an invalid entry has x == UINT64_MAX)
#pragma unroll
    for (int j = 0; j < WMAX; ++j) { xs[j] = UINT64_MAX; ys[j] = 0xffffffffu; }   // slots >= w stay invalid forever
    auto ring_get = [&](int j) {
        A128 v;
        v.x = UINT64_MAX;
        uint32_t yl = 0xffffffffu;
#pragma unroll
        for (int q = 0; q < WMAX; ++q)
            if (q == j) { v.x = xs[q]; yl = ys[q]; }
        v.y = v.x == UINT64_MAX ? UINT64_MAX : (y_hi | yl);
        return v;
    };
    for (int i = 0; i < len; ++i) {
        const int c = base_at(i);
        A128 info;
        info.x = info.y = UINT64_MAX;
        if (c < 4) {
            kmer_span = l + 1 < k ? l + 1 : k;
            kmer0 = (kmer0 << 2 | (uint64_t)c) & mask;
            kmer1 = (kmer1 >> 2) | (uint64_t)(3 ^ c) << shift1;
            if (kmer0 == kmer1) continue;   // strand-symmetric k-mer: skipped without advancing the window
            const int z = kmer0 < kmer1 ? 0 : 1;
            ++l;
            if (l >= k && kmer_span < 256) {
                info.x = mz_hash64(z ? kmer1 : kmer0, mask) << 8 | (uint64_t)kmer_span;
                info.y = y_hi | (uint32_t)i << 1 | (uint32_t)z;
            }
        } else {
            l = 0;
            kmer_span = 0;
        }
#pragma unroll
        for (int j = 0; j < WMAX; ++j)
            if (j == buf_pos) { xs[j] = info.x; ys[j] = (uint32_t)info.y; }
        if (l == w + k - 1 && mn.x != UINT64_MAX) {   // first full window: emit earlier identical k-mers
            for (int j = buf_pos + 1; j < w; ++j)
                { const A128 bj = ring_get(j); if (mn.x == bj.x && bj.y != mn.y) push(bj.x, bj.y); }
            for (int j = 0; j < buf_pos; ++j)
                { const A128 bj = ring_get(j); if (mn.x == bj.x && bj.y != mn.y) push(bj.x, bj.y); }
        }
        // the three outcomes of sketch_segment_t's if / else-if, with ONE push site and a branch-free re-scan (in a
        // wave of 64 reads some lane needs the re-scan at almost every base, so it is computed unconditionally)
        const bool new_min = info.x <= mn.x;
        const bool slid = !new_min && buf_pos == min_pos;
        if (mn.x != UINT64_MAX && ((new_min && l >= w + k) || (slid && l >= w + k - 1))) push(mn.x, mn.y);
        uint64_t m = UINT64_MAX;
#pragma unroll
        for (int j = 0; j < WMAX; ++j) m = xs[j] < m ? xs[j] : m;
        uint32_t eq = 0;   // slots that hold the minimum (never empty; slots >= w only when everything is invalid)
#pragma unroll
        for (int j = 0; j < WMAX; ++j) eq |= xs[j] == m ? 1u << j : 0u;
        // last in scan order (j = buf_pos+1 .. w-1, 0 .. buf_pos, '>=') = the largest slot <= buf_pos if there is
        // one, else the largest slot
        const uint32_t lo = eq & ((2u << buf_pos) - 1u);
        const int best_j = 31 - __builtin_clz(lo ? lo : eq);
        uint32_t best_y = 0xffffffffu;
#pragma unroll
        for (int j = 0; j < WMAX; ++j)
            if (j == best_j) best_y = ys[j];
        if (new_min) {
            mn = info;
            min_pos = buf_pos;
        } else if (slid) {
            mn.x = m;
            mn.y = m == UINT64_MAX ? UINT64_MAX : (y_hi | best_y);
            min_pos = best_j;
            if (l >= w + k - 1 && mn.x != UINT64_MAX && (eq & ~(1u << min_pos)) != 0u) {   // other occurrences of the minimum
                for (int j = buf_pos + 1; j < w; ++j)
                    if ((eq >> j & 1u) && j != min_pos) { const A128 bj = ring_get(j); push(bj.x, bj.y); }
                for (int j = 0; j <= buf_pos; ++j)
                    if ((eq >> j & 1u) && j != min_pos) { const A128 bj = ring_get(j); push(bj.x, bj.y); }
            }
        }
        if (++buf_pos == w) buf_pos = 0;
    }
    if (final_push && mn.x != UINT64_MAX) push(mn.x, mn.y);
}

// A slice of a long sequence's sketch, computed on its own (the reference index is built by one thread per slice,
// ref_index_kernels.hip): the minimizers that sketch.c:77-143 emits while it processes bases [begin, end) of a sequence
// of `len` bases (the closing push of the last minimum belongs to the slice that ends at len).  With k odd no k-mer is
// its own reverse complement, so the window advances at every base and the state after base i -- the last w entries,
// their minimum (always the newest minimal entry, whatever path led there), the run length since the last ambiguous base
// as far as the thresholds k, w+k-1 and w+k can tell -- is a function of the bases (i - w - k, i]: a run that starts
// w + k + 1 bases early is in the true state by `begin` (tests/test_align_host.py runs the slices against the whole).  emit(x, y): y carries the position in the whole sequence.
template <int WMAX, class BaseFn, class EmitFn>
struct SketchSlice {
    BaseFn& base; EmitFn& emit; int s0, begin, cur;
    PMX_HD int operator()(int i) { cur = i; return base(s0 + i); }
    PMX_HD void operator()(uint64_t x, uint64_t y) { if (s0 + cur >= begin) emit(x, y + ((uint64_t)(uint32_t)s0 << 1)); }
};
template <int WMAX, class BaseFn, class EmitFn>
PMX_HD void sketch_slice(int begin, int end, int len, int w, int k, BaseFn& base_at, EmitFn& emit) {
    const int warm = w + k + 1;
    SketchSlice<WMAX, BaseFn, EmitFn> f{base_at, emit, begin > warm ? begin - warm : 0, begin, 0};
    sketch_core<WMAX>(end - f.s0, w, k, 0, f, f, end >= len);
}

#if PMX_W == 1
template <int WMAX>
PMX_HD void sketch_segment_reg(Work& W, Ptr<const uint8_t> seq, int len, int w, int k, uint32_t rid) {
    PMX_LDS(&W); PMX_LDS(seq);
    Ptr<A128> mvp = W.mv; PMX_LDS(mvp);
    // the output cursor and its bound live in registers for the whole segment (W is memory to the compiler)
    int n_mv = W.n_mv;
    const int max_mini = W.caps.max_mini;
    bool overflow = false;
    ByteReader seq_r(seq);
    auto base_at = [&](int i) { return (int)seq_r[i]; };
    auto push = [&](uint64_t x, uint64_t y) {
        if (n_mv < max_mini) { A128 v; v.x = x; v.y = y; mvp[n_mv++] = v; }
        else overflow = true;
    };
    sketch_core<WMAX>(len, w, k, (uint64_t)rid << 32, base_at, push);
    W.n_mv = n_mv;
    if (overflow) W.status |= PMX_ST_OVERFLOW;
}
#endif

#if PMX_W == 64 && defined(__HIP_DEVICE_COMPILE__)
// Wave-per-pair kernels: the same sketch with the window ring held ACROSS THE LANES (lane j = slot j, w <= 64).  All
// scalar state is wave-uniform (the base is broadcast with readfirstlane, so the k-mer arithmetic runs on the scalar
// unit); the ring write is one predicated move, the re-scan a wave-wide 64-bit minimum plus a ballot, and reading a
// slot a readlane -- no LDS round trips in the per-base loop (the ring-in-LDS form spends ~1200 cycles per base,
// more than half of a tier-1 pair).  Scan order, tie rules and the emitted minimizers are sketch_segment_t's (see
// sketch_segment_reg for the argument).
__device__ __forceinline__ uint64_t sk_rl64(uint64_t v, int l) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return (uint64_t)hi << 32 | lo;
}
__device__ void sketch_segment_wave(Work& W, const uint8_t* seq, int len, int w, int k, uint32_t rid) {
    const uint64_t shift1 = 2 * (uint64_t)(k - 1), mask = (1ULL << 2 * k) - 1;
    const uint64_t y_hi = (uint64_t)rid << 32;
    const int lane = lane_id();
    uint64_t kmer0 = 0, kmer1 = 0;
    int l = 0, buf_pos = 0, min_pos = 0, kmer_span = 0;
    PMX_LDS(&W); PMX_LDS(seq);
    A128* mvp = W.mv; PMX_LDS(mvp);
    int n_mv = W.n_mv;
    const int max_mini = W.caps.max_mini;
    bool overflow = false;
    A128 mn;
    mn.x = mn.y = UINT64_MAX;
    uint64_t rx = UINT64_MAX;     // this lane's ring slot (lanes >= w stay invalid)
    uint32_t ry = 0xffffffffu;
    auto ring_get = [&](int j) {
        A128 v;
        v.x = sk_rl64(rx, j);
        const uint32_t yl = (uint32_t)__builtin_amdgcn_readlane((int)ry, j);
        v.y = v.x == UINT64_MAX ? UINT64_MAX : (y_hi | yl);
        return v;
    };
#define PMX_MV_PUSH(val)                                      \
    do {                                                      \
        if (n_mv < max_mini) { if (lane == 0) mvp[n_mv] = (val); ++n_mv; } \
        else overflow = true;                                 \
    } while (0)
    for (int i = 0; i < len; ++i) {
        const int c = __builtin_amdgcn_readfirstlane((int)seq[i]);
        A128 info;
        info.x = info.y = UINT64_MAX;
        if (c < 4) {
            kmer_span = l + 1 < k ? l + 1 : k;
            kmer0 = (kmer0 << 2 | (uint64_t)c) & mask;
            kmer1 = (kmer1 >> 2) | (uint64_t)(3 ^ c) << shift1;
            if (kmer0 == kmer1) continue;   // strand-symmetric k-mer: skipped without advancing the window
            const int z = kmer0 < kmer1 ? 0 : 1;
            ++l;
            if (l >= k && kmer_span < 256) {
                info.x = mz_hash64(z ? kmer1 : kmer0, mask) << 8 | (uint64_t)kmer_span;
                info.y = y_hi | (uint32_t)i << 1 | (uint32_t)z;
            }
        } else {
            l = 0;
            kmer_span = 0;
        }
        if (lane == buf_pos) { rx = info.x; ry = (uint32_t)info.y; }
        if (l == w + k - 1 && mn.x != UINT64_MAX) {   // first full window: emit earlier identical k-mers
            for (int j = buf_pos + 1; j < w; ++j)
                { const A128 bj = ring_get(j); if (mn.x == bj.x && bj.y != mn.y) PMX_MV_PUSH(bj); }
            for (int j = 0; j < buf_pos; ++j)
                { const A128 bj = ring_get(j); if (mn.x == bj.x && bj.y != mn.y) PMX_MV_PUSH(bj); }
        }
        if (info.x <= mn.x) {                          // new minimum: flush the old one
            if (l >= w + k && mn.x != UINT64_MAX) PMX_MV_PUSH(mn);
            mn = info;
            min_pos = buf_pos;
        } else if (buf_pos == min_pos) {               // the minimum slid out of the window
            if (l >= w + k - 1 && mn.x != UINT64_MAX) PMX_MV_PUSH(mn);
            uint64_t m = rx;                           // wave-wide minimum of the slots
            for (int o = 32; o > 0; o >>= 1) {
                const uint64_t other = __shfl_xor(m, o);
                m = other < m ? other : m;
            }
            m = sk_rl64(m, 0);                         // (every lane holds it; make it a scalar)
            const unsigned long long eq = __ballot(lane < w && rx == m);   // slots that hold it (never empty)
            // last in scan order (j = buf_pos+1 .. w-1, 0 .. buf_pos, '>=') = the largest slot <= buf_pos if there is
            // one, else the largest slot
            const unsigned long long lo = eq & ((2ULL << buf_pos) - 1ULL);
            const int best_j = 63 - __builtin_clzll(lo ? lo : eq);
            const uint32_t best_y = (uint32_t)__builtin_amdgcn_readlane((int)ry, best_j);
            mn.x = m;
            mn.y = m == UINT64_MAX ? UINT64_MAX : (y_hi | best_y);
            min_pos = best_j;
            if (l >= w + k - 1 && mn.x != UINT64_MAX && (eq & ~(1ULL << min_pos)) != 0ULL) {   // other occurrences of the minimum
                for (int j = buf_pos + 1; j < w; ++j)
                    if ((eq >> j & 1ULL) && j != min_pos) PMX_MV_PUSH(ring_get(j));
                for (int j = 0; j <= buf_pos; ++j)
                    if ((eq >> j & 1ULL) && j != min_pos) PMX_MV_PUSH(ring_get(j));
            }
        }
        if (++buf_pos == w) buf_pos = 0;
    }
    if (mn.x != UINT64_MAX) PMX_MV_PUSH(mn);
#undef PMX_MV_PUSH
    wave_sync();
    W.n_mv = n_mv;
    if (overflow) W.status |= PMX_ST_OVERFLOW;
}
#endif

#if PMX_W == 64 && defined(__HIP_DEVICE_COMPILE__)
// Wave-per-pair kernels, k odd and w <= 20: the 64 lanes sketch 64 consecutive slices of the segment on their own
// (sketch_slice above: w + k + 1 bases of run-in reproduce the sequential state, tests/test_align_host.py) -- ~36 steps
// per lane for a 150-base read instead of 150 wave-uniform steps with the ring across the lanes.  A lane keeps its (few)
// minimizers in registers; a wave prefix sum gives every lane its place in W.mv.
// (the bases of a long read sit in the wave's HBM slab: one byte load per base was one memory round trip per step of a lane's
//  walk; the walk is sequential, so four bases come per load and the next word is requested while these are used)
struct SliceBase {
    const uint8_t* seq;
    int cur = -2;
    uint32_t wd = 0, nxt = 0;
    __device__ int operator()(int i) {
        const int wi = i >> 2;
        if (wi != cur) {
            const uint32_t* p = reinterpret_cast<const uint32_t*>(seq);   // (the sequence arrays are 16-byte aligned and padded by 16)
            wd = wi == cur + 1 ? nxt : p[wi];
            nxt = p[wi + 1];
            cur = wi;
        }
        return (int)(wd >> (8 * (i & 3)) & 0xffu);
    }
};
struct SliceKeep {   // a lane's own minimizers: a handful at most for a slice of a few bases
    uint64_t x0, y0, x1, y1, x2, y2, x3, y3;
    int n;
    __device__ void operator()(uint64_t x, uint64_t y) {
        if (n == 0) { x0 = x; y0 = y; } else if (n == 1) { x1 = x; y1 = y; } else if (n == 2) { x2 = x; y2 = y; } else if (n == 3) { x3 = x; y3 = y; }
        ++n;
    }
};
struct SliceWrite { A128* mv; int at; uint64_t y_hi; __device__ void operator()(uint64_t x, uint64_t y) { A128 v; v.x = x; v.y = y_hi | y; mv[at++] = v; } };
template <int WMAX>
__device__ void sketch_segment_lanes(Work& W, const uint8_t* seq, int len, int w, int k, uint32_t rid) {
    PMX_LDS(&W); PMX_LDS(seq);
    A128* mvp = W.mv; PMX_LDS(mvp);
    const int lane = lane_id();
    const int per = (len + 63) / 64;
    const int begin = lane * per, end = begin + per < len ? begin + per : len;
    SliceBase base;
    base.seq = seq;
    SliceKeep kp;
    kp.n = 0; kp.x0 = kp.y0 = kp.x1 = kp.y1 = kp.x2 = kp.y2 = kp.x3 = kp.y3 = 0;
    if (begin < len) sketch_slice<WMAX>(begin, end, len, w, k, base, kp);
    int incl = kp.n;
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    const int total = __shfl(incl, 63);
    if (W.n_mv + total > W.caps.max_mini) { W.status |= PMX_ST_OVERFLOW; return; }
    SliceWrite wr{mvp, W.n_mv + incl - kp.n, (uint64_t)rid << 32};
    if (__ballot(kp.n > 4) == 0ULL) {   // (uniform) the usual case: what the lanes kept goes out
        if (kp.n > 0) wr(kp.x0, kp.y0);
        if (kp.n > 1) wr(kp.x1, kp.y1);
        if (kp.n > 2) wr(kp.x2, kp.y2);
        if (kp.n > 3) wr(kp.x3, kp.y3);
    } else if (begin < len && kp.n > 0) sketch_slice<WMAX>(begin, end, len, w, k, base, wr);   // a repeat-rich slice: second pass
    W.n_mv += total;
}
#endif

PMX_HDN void sketch_segment(Work& W, Ptr<const uint8_t> seq, int len, int w, int k, uint32_t rid) {
#if PMX_W == 64 && defined(__HIP_DEVICE_COMPILE__)
    if ((k & 1) && w >= 1 && w <= 20 && len >= 64 && !W.sk_no_lane_ring) {   // (sr: w = 11, map-ont: 10, map-hifi: 19)
        if (w <= 12) sketch_segment_lanes<12>(W, seq, len, w, k, rid);
        else sketch_segment_lanes<20>(W, seq, len, w, k, rid);
        wave_sync();
        return;
    }
    if (w >= 1 && w <= 64 && !W.sk_no_lane_ring) {
        sketch_segment_wave(W, seq, len, w, k, rid);
        return;
    }
#endif
#if PMX_W == 1
    if (w >= 1 && w <= 16) {   // the ring is unrolled to the next size class (sr preset: w = 11)
        if (w <= 8) sketch_segment_reg<8>(W, seq, len, w, k, rid);
        else if (w <= 12) sketch_segment_reg<12>(W, seq, len, w, k, rid);
        else sketch_segment_reg<16>(W, seq, len, w, k, rid);
        return;
    }
#endif
#ifdef PMX_INTERLEAVED
    if (W.sk_lds_x) {
        RingLds ring{W.sk_lds_x, W.sk_lds_y, (uint64_t)rid << 32};
        sketch_segment_t(W, ring, seq, len, w, k, rid);
        return;
    }
#endif
    RingMem ring{W.sk_buf};
    PMX_LDS(ring.buf);
    sketch_segment_t(W, ring, seq, len, w, k, rid);
}

// collect_minimizers (map.c:59-73): segment s gets rid = s and its positions shifted by the summed
// lengths of the previous segments.  (sdust masking is off: options.c:23.)
PMX_HD void collect_minimizers(Work& W, const Opt& o) {
    PMX_LDS(&W);
    Ptr<A128> mvp = W.mv; PMX_LDS(mvp);
    W.n_mv = 0;
    int sum = 0;
    for (int s = 0; s < W.n_segs; ++s) {
        const int n0 = W.n_mv;
        sketch_segment(W, W.qseq[s][0], W.qlen[s], o.w, o.k, (uint32_t)s);
        if (sum != 0) {   // (a single segment -- every long read -- has nothing to shift: ~1,800 dependent read-modify-writes in the slab)
            wave_sync();
            for (int j = n0 + lane_id(); j < W.n_mv; j += PMX_W) mvp[j].y += (uint64_t)sum << 1;
            wave_sync();
        }
        sum += W.qlen[s];
    }
}

// mm_seed_mz_flt (seed.c:5-26): a query with more than q_occ_max minimizers drops every minimizer VALUE that occurs in
// it more than q_occ_max times and more than q_occ_frac of all (long reads through repeats).  Which copies go does not
// depend on how the sort orders equal keys (whole groups go, survivors keep their order).  Scratch: W.a2 (idle here).
PMX_HDN void seed_mz_flt(Work& W, int32_t q_occ_max, float q_occ_frac) {
    PMX_LDS(&W);
    const int n = W.n_mv;
    if (n <= q_occ_max || q_occ_frac <= 0.0f || q_occ_max <= 0) return;
    if (n > W.caps.max_anchor) { W.status |= PMX_ST_OVERFLOW; return; }
    Ptr<A128> mv = W.mv; PMX_LDS(mv);
    Ptr<A128> a = W.a2; PMX_LDS(a);
    wave_sync();
#if PMX_W > 1 && !defined(PMX_ALL_LDS)
    if (W.dp_fast && (size_t)9 * (W.caps.dp_fast_tlen + 32) >= 4096) {
        // no minimizer value can occur more than q_occ_max times if no bucket of a 1,024-bucket count of the values holds
        // more than that (the usual case: nothing to filter, and the sort below -- a sequential procedure -- is skipped)
        uint32_t* cnt = reinterpret_cast<uint32_t*>(W.dp_fast);
        const int lane = lane_id();
        for (int b = lane; b < 1024; b += PMX_W) cnt[b] = 0;
        wave_sync();
        uint32_t mx = 0;
        for (int i = lane; i < n; i += PMX_W) {
            const uint32_t c = atomicAdd(&cnt[(uint32_t)mix64(mv[i].x >> 8) & 1023u], 1u) + 1u;
            mx = c > mx ? c : mx;
        }
        const bool over = (int32_t)mx > q_occ_max;
        wave_sync();
        if (__ballot(over) == 0ULL) return;
    }
#endif
    for (int i = lane_id(); i < n; i += PMX_W) { A128 t; t.x = mv[i].x; t.y = (uint64_t)i; a[i] = t; }
    wave_sync();
    radix_sort_128x(a, a + n, &W.status);
    wave_sync();
    bool any = false;
    for (int st = 0, i = 1; i <= n; ++i) {
        if (i == n || a[i].x != a[st].x) {
            const int32_t cnt = i - st;
            if (cnt > q_occ_max && (float)cnt > (float)n * q_occ_frac) {
                any = true;
                for (int j = st; j < i; ++j) mv[(int)a[j].y].x = 0;
            }
            st = i;
        }
    }
    wave_sync();
    if (any) {
        int j = 0;
        for (int i = 0; i < n; ++i) {
            const A128 t = mv[i];
            if (t.x != 0) { mv[j] = t; ++j; }
        }
        W.n_mv = j;
    }
    wave_sync();
}

// mm_idx_get (index.c:81-99)
PMX_HD uint32_t index_lookup(const RefIndex& ri, uint64_t minier, uint32_t* off) {
    uint32_t slot = (uint32_t)mix64(minier) & ri.ht_mask;
    while (true) {
        const HtEnt e = ri.ht[slot];
        if (e.key == minier) { *off = e.off; return e.cnt; }
        if (e.key == UINT64_MAX) return 0;
        slot = (slot + 1) & ri.ht_mask;
    }
}

// binary heaps exactly as ksort.h:43-59 builds them (ties must break the same way)
PMX_HD void heap_down_min_x(Ptr<A128> l, int i, int n) {   // "less" = a.x > b.x  -> min-heap on x (map.c:76)
    PMX_LDS(l);
    int k = i;
    const A128 tmp = l[i];
    while ((k = (k << 1) + 1) < n) {
        if (k != n - 1 && l[k].x > l[k + 1].x) ++k;
        if (l[k].x > tmp.x) break;
        l[i] = l[k];
        i = k;
    }
    l[i] = tmp;
}
// heap_down_min_x(l, 0, n) for a root that is about to be overwritten with tmp: sifts tmp down from the root
// without first storing it, reads both children of a node with independent loads, and returns the new root so the
// caller does not have to load it back (same comparisons in the same order as ks_heapdown)
PMX_HD A128 heap_replace_root_min_x(Ptr<A128> l, int n, const A128 tmp) {
    PMX_LDS(l);
    int i = 0, k;
    A128 root = tmp;
    while ((k = (i << 1) + 1) < n) {
        A128 ck = l[k];
        if (k != n - 1) {
            const A128 ck1 = l[k + 1];
            if (ck.x > ck1.x) { ++k; ck = ck1; }
        }
        if (ck.x > tmp.x) break;
        l[i] = ck;
        if (i == 0) root = ck;
        i = k;
    }
    l[i] = tmp;
    return root;
}
// the same with the root's two children cached in registers (c1 = l[1], c2 = l[2], kept in step with memory): the first
// level of the sift needs no load, i.e. one dependent round trip less per pop
PMX_HD A128 heap_replace_root_min_x_c(Ptr<A128> l, int n, const A128 tmp, A128& c1, A128& c2) {
    PMX_LDS(l);
    if (n <= 1) { l[0] = tmp; return tmp; }
    int k = 1;
    A128 ck = c1;
    if (n > 2 && c1.x > c2.x) { k = 2; ck = c2; }
    if (ck.x > tmp.x) { l[0] = tmp; return tmp; }
    l[0] = ck;
    int i = k, kk;
    A128 at_k = tmp;   // what ends up at position k
    while ((kk = (i << 1) + 1) < n) {
        A128 ch = l[kk];
        if (kk != n - 1) {
            const A128 ch1 = l[kk + 1];
            if (ch.x > ch1.x) { ++kk; ch = ch1; }
        }
        if (ch.x > tmp.x) break;
        l[i] = ch;
        if (i == k) at_k = ch;
        i = kk;
    }
    l[i] = tmp;
    if (k == 1) c1 = at_k; else c2 = at_k;
    return ck;
}
PMX_HD void heap_down_max_u64(uint64_t* l, int i, int n) {   // ks_heapdown_uint64_t: max-heap
    int k = i;
    const uint64_t tmp = l[i];
    while ((k = (k << 1) + 1) < n) {
        if (k != n - 1 && l[k] < l[k + 1]) ++k;
        if (l[k] < tmp) break;
        l[i] = l[k];
        i = k;
    }
    l[i] = tmp;
}

// mm_seed_select (seed.c:56-96): within a streak of high-occurrence minimizers keep the
// max_high_occ least frequent ones.
PMX_HDN void seed_select(int n, Ptr<SeedA> a, int len, int max_occ, int max_max_occ, int dist) {
    PMX_LDS(a);
    if (n == 0 || n == 1) return;
    int m = 0;
    for (int i = 0; i < n; ++i)
        if ((int)a[i].n > max_occ) ++m;
    if (m == 0) return;
    uint64_t b[128];
    int last0 = -1;
    for (int i = 0; i <= n; ++i) {
        if (i == n || (int)a[i].n <= max_occ) {
            if (i - last0 > 1) {
                const int ps = last0 < 0 ? 0 : (int)(a[last0].q_pos >> 1);
                const int pe = i == n ? len : (int)(a[i].q_pos >> 1);
                const int st = last0 + 1, en = i;
                int max_high_occ = (int)((double)(pe - ps) / dist + .499);
                if (max_high_occ > 0) {
                    if (max_high_occ > 128) max_high_occ = 128;
                    int j, k;
                    for (j = st, k = 0; j < en && k < max_high_occ; ++j, ++k) b[k] = (uint64_t)a[j].n << 32 | (uint32_t)j;
                    for (int q = (k >> 1) - 1; q >= 0; --q) heap_down_max_u64(b, q, k);
                    for (; j < en; ++j) {
                        if ((int)a[j].n < (int)(b[0] >> 32)) {
                            b[0] = (uint64_t)a[j].n << 32 | (uint32_t)j;
                            heap_down_max_u64(b, 0, k);
                        }
                    }
                    for (j = 0; j < k; ++j) a[(uint32_t)b[j]].flt = 1;
                }
                for (int j = st; j < en; ++j) a[j].flt ^= 1;
                for (int j = st; j < en; ++j)
                    if ((int)a[j].n > max_max_occ) a[j].flt = 1;
            }
            last0 = i;
        }
    }
}

// mm_collect_matches (seed.c:98-131) + mm_seed_collect_all (:28-52).  The index probes of all
// minimizers are issued lane-parallel first (one L2 round trip instead of n_mv dependent ones).
PMX_HDN void collect_matches(Work& W, const Opt& o, const RefIndex& ri, int qlen, int max_occ) {
    PMX_LDS(&W);
    Ptr<A128> mv = W.mv; PMX_LDS(mv);
    Ptr<A128> hp = W.heap; PMX_LDS(hp);   // (wave models: scratch for the lane-parallel probes)
    (void)hp;
    Ptr<SeedA> seeds = W.seeds; PMX_LDS(seeds);
    Ptr<SeedB> seeds_b = W.seeds_b; PMX_LDS(seeds_b);
    Ptr<uint64_t> mini_pos = W.mini_pos;   // global scratch (only mm_est_err would read it)
    int n_m0 = 0;
#if PMX_W == 1
    // scalar models: four probes in flight at a time (their first loads do not depend on each other), and the seed
    // records are built straight from the probe results (no pass through heap[] scratch)
    {
        const int n_mv = W.n_mv;
        bool any_high = false;          // some seed occurs more than max_occ times
        int64_t sum_n = 0;
        uint64_t prev_key = 0;          // mv[i - 1].x >> 8
        for (int i0 = 0; i0 < n_mv; i0 += 4) {
            A128 m[5];
            HtEnt e[4];
            uint32_t slot[4];
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                m[b].x = m[b].y = 0;
                if (i0 + b < n_mv) m[b] = mv[i0 + b];
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                slot[b] = (uint32_t)mix64(m[b].x >> 8) & ri.ht_mask;
                e[b] = HtEnt{UINT64_MAX, 0u, 0u};
                if (i0 + b < n_mv) e[b] = ri.ht[slot[b]];
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (i0 + b >= n_mv) break;
                const uint64_t key = m[b].x >> 8;
                while (e[b].key != key && e[b].key != UINT64_MAX) {   // collision: keep probing
                    slot[b] = (slot[b] + 1) & ri.ht_mask;
                    e[b] = ri.ht[slot[b]];
                }
                const uint32_t t = e[b].key == key ? e[b].cnt : 0u;
                if (t != 0) {
                    const int i = i0 + b;
                    SeedA q;
                    SeedB qb;
                    q.q_pos = (uint32_t)m[b].y;
                    qb.q_span = (uint32_t)(m[b].x & 0xff);
                    q.off = e[b].off;
                    q.n = t;
                    qb.seg_id = (uint32_t)(m[b].y >> 32);
                    qb.is_tandem = q.flt = 0;
                    qb.pad = 0;
                    if (i > 0 && key == prev_key) qb.is_tandem = 1;
                    if (i < n_mv - 1 && key == m[b + 1].x >> 8) qb.is_tandem = 1;
                    seeds[n_m0] = q;
                    seeds_b[n_m0] = qb;
                    ++n_m0;
                    any_high = any_high || (int)t > max_occ;
                    sum_n += t;
                }
                prev_key = key;
            }
        }
        if (!any_high) {
            // no seed above the occurrence cap (the usual case): mm_seed_select / the flt pass mark nothing and the
            // compaction below is the identity, so its results are known here (mini_pos[] has no reader on this path)
            W.n_mini_pos = n_m0;
            W.n_seeds = n_m0;
            W.n_a = sum_n;
            W.rep_len = 0;
            return;
        }
    }
#else
    const int lane = lane_id();
    {
        // 64 minimizers per step: probe, tandem flags from the neighbours, and the seed records of the minimizers the index
        // knows written through a prefix sum.  When no seed is above the occurrence cap (the usual case) that is all of
        // mm_collect_matches: mm_seed_select / the flt pass mark nothing and the compaction is the identity.
        const int n_mv = W.n_mv;
        bool high = false;
        int64_t sum_n = 0;
        int n_out = 0;
        for (int i0 = 0; i0 < n_mv; i0 += PMX_W) {
            const int i = i0 + lane;
            uint32_t off = 0, t = 0;
            A128 p;
            p.x = p.y = 0;
            bool tandem = false;
            if (i < n_mv) {
                p = mv[i];
                const uint64_t key = p.x >> 8;
                t = index_lookup(ri, key, &off);
                hp[i].x = off;
                hp[i].y = t;
                if (i > 0 && key == mv[i - 1].x >> 8) tandem = true;
                if (i < n_mv - 1 && key == mv[i + 1].x >> 8) tandem = true;
            }
            const unsigned long long keep = __ballot(t != 0);
            const int at = n_out + (int)__builtin_popcountll(keep & ((1ULL << lane) - 1ULL));
            if (t != 0) {
                SeedA q;
                SeedB qb;
                q.q_pos = (uint32_t)p.y;
                q.off = off;
                q.n = t;
                q.flt = 0;
                qb.q_span = (uint32_t)(p.x & 0xff);
                qb.seg_id = (uint32_t)(p.y >> 32);
                qb.is_tandem = tandem ? 1u : 0u;
                qb.pad = 0;
                seeds[at] = q;
                seeds_b[at] = qb;
            }
            n_out += (int)__builtin_popcountll(keep);
            high = high || (int)t > max_occ;
            sum_n += (int64_t)t;
        }
        wave_sync();
        if (__ballot(high) == 0ULL) {
            for (int d = PMX_W / 2; d > 0; d >>= 1) sum_n += __shfl_xor(sum_n, d);
            W.n_mini_pos = n_out;
            W.n_seeds = n_out;
            W.n_a = sum_n;
            W.rep_len = 0;
            return;
        }
    }
    for (int i = 0; i < W.n_mv; ++i) {
        const A128 p = mv[i];
        const uint32_t off = (uint32_t)hp[i].x;
        const uint32_t t = (uint32_t)hp[i].y;
        if (t == 0) continue;
        SeedA q;
        SeedB qb;
        q.q_pos = (uint32_t)p.y;
        qb.q_span = (uint32_t)(p.x & 0xff);
        q.off = off;
        q.n = t;
        qb.seg_id = (uint32_t)(p.y >> 32);
        qb.is_tandem = q.flt = 0;
        qb.pad = 0;
        if (i > 0 && p.x >> 8 == mv[i - 1].x >> 8) qb.is_tandem = 1;
        if (i < W.n_mv - 1 && p.x >> 8 == mv[i + 1].x >> 8) qb.is_tandem = 1;
        seeds[n_m0] = q;
        seeds_b[n_m0] = qb;
        ++n_m0;
    }
#endif
    wave_sync();
    if (o.occ_dist > 0 && o.max_max_occ > max_occ) seed_select(n_m0, seeds, qlen, max_occ, o.max_max_occ, o.occ_dist);
    else
        for (int i = 0; i < n_m0; ++i)
            if ((int)seeds[i].n > max_occ) seeds[i].flt = 1;
    int rep_st = 0, rep_en = 0, n_m = 0, rep_len = 0;
    int64_t n_a = 0;
    W.n_mini_pos = 0;
    for (int i = 0; i < n_m0; ++i) {
        const SeedA q = seeds[i];
        const SeedB qb = seeds_b[i];
        if (q.flt) {
            const int en = (int)(q.q_pos >> 1) + 1, st = en - (int)qb.q_span;
            if (st > rep_en) {
                rep_len += rep_en - rep_st;
                rep_st = st;
                rep_en = en;
            } else rep_en = en;
        } else {
            n_a += q.n;
            mini_pos[W.n_mini_pos++] = (uint64_t)qb.q_span << 32 | q.q_pos >> 1;
            seeds[n_m] = q;
            seeds_b[n_m] = qb;
            ++n_m;
        }
    }
    rep_len += rep_en - rep_st;
    W.n_seeds = n_m;
    W.n_a = n_a;
    W.rep_len = rep_len;
}

// collect_seed_hits_heap (map.c:102-166): k-way merge of the occurrence lists by reference position;
// forward-strand anchors first, then the reverse-strand ones, both ascending.
PMX_HDN void collect_seed_hits_heap(Work& W, const Opt& o, const RefIndex& ri, int qlen, int max_occ) {
    collect_matches(W, o, ri, qlen, max_occ);
    PMX_STAMP(W, 16);
    if (W.n_a > W.caps.max_anchor) {
        W.status |= PMX_ST_OVERFLOW;
        W.n_a = 0;
        return;
    }
    PMX_LDS(&W);
    const int n_m = W.n_seeds;
    const int64_t n_a = W.n_a;
    Ptr<A128> heap = W.heap; PMX_LDS(heap);
    Ptr<A128> a = W.a; PMX_LDS(a);
    Ptr<SeedA> seeds = W.seeds; PMX_LDS(seeds);
    Ptr<SeedB> seeds_b = W.seeds_b; PMX_LDS(seeds_b);
    // stage every occurrence list in idle scratch so the merge never waits on HBM/L2:
    // cache offsets are assigned in seed order, the copies run one seed per lane
    Ptr<uint64_t> pc = ptr_cast<uint64_t>(W.seg_a[0]); PMX_LDS(pc);   // the per-mate anchor block (16*max_anchor bytes) is idle until seg_gen
    int heap_size = 0;
#if PMX_W == 1
    {   // scalar models: one pass, four seeds at a time (their records and first positions are independent loads)
        uint32_t acc = 0;
        for (int i0 = 0; i0 < n_m; i0 += 4) {
            SeedA q[4];
            uint64_t p0[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                q[b].q_pos = q[b].off = q[b].n = q[b].flt = 0;
                if (i0 + b < n_m) q[b] = seeds[i0 + b];
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                p0[b] = 0;
                if (i0 + b < n_m && q[b].n > 0) p0[b] = ri.pos[q[b].off];
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (i0 + b >= n_m) break;
                seeds[i0 + b].flt = acc;   // flt is free now: cache offset
                if (q[b].n > 0) {
                    pc[acc] = p0[b];
                    for (uint32_t j = 1; j < q[b].n; ++j) pc[acc + j] = ri.pos[q[b].off + j];
                    A128 h;
                    h.x = p0[b];
                    h.y = (uint64_t)(i0 + b) << 32;
                    heap[heap_size++] = h;
                }
                acc += q[b].n;
            }
        }
    }
#else
    {
        uint32_t acc = 0;
        for (int i = 0; i < n_m; ++i) { const uint32_t n = seeds[i].n; seeds[i].flt = acc; acc += n; }   // flt is free now: cache offset
        wave_sync();
        for (int i = lane_id(); i < n_m; i += PMX_W) {
            const SeedA q = seeds[i];
            for (uint32_t j = 0; j < q.n; ++j) pc[q.flt + j] = ri.pos[q.off + j];
        }
        wave_sync();
    }
    for (int i = 0; i < n_m; ++i) {
        if (seeds[i].n > 0) {
            heap[heap_size].x = pc[seeds[i].flt];
            heap[heap_size].y = (uint64_t)i << 32;
            ++heap_size;
        }
    }
#endif
    for (int q = (heap_size >> 1) - 1; q >= 0; --q) heap_down_min_x(heap, q, heap_size);
    int64_t n_for = 0, n_rev = 0;
    PMX_STAMP(W, 17);
    A128 top, hc1, hc2;   // heap[0], heap[1], heap[2] kept in registers from pop to pop
    top.x = top.y = hc1.x = hc1.y = hc2.x = hc2.y = 0;
    if (heap_size > 0) top = heap[0];
    if (heap_size > 1) hc1 = heap[1];
    if (heap_size > 2) hc2 = heap[2];
    while (heap_size > 0) {
        const uint32_t si = (uint32_t)(top.y >> 32);
        const SeedA q = seeds[si];
        const SeedB qb = seeds_b[si];
        const uint64_t r = top.x;
        const int32_t rpos = (int32_t)((uint32_t)r >> 1);
        A128 p;
        if ((r & 1) == (q.q_pos & 1)) {   // forward strand
            p.x = (r & 0xffffffff00000000ULL) | (uint32_t)rpos;
            p.y = (uint64_t)qb.q_span << 32 | q.q_pos >> 1;
            p.y |= (uint64_t)qb.seg_id << PMX_SEED_SEG_SHIFT;
            if (qb.is_tandem) p.y |= PMX_SEED_TANDEM;
            a[n_for++] = p;
        } else {                          // reverse strand: query position mirrored
            p.x = 1ULL << 63 | (r & 0xffffffff00000000ULL) | (uint32_t)rpos;
            p.y = (uint64_t)qb.q_span << 32 | (uint32_t)(qlen - ((int)(q.q_pos >> 1) + 1 - (int)qb.q_span) - 1);
            p.y |= (uint64_t)qb.seg_id << PMX_SEED_SEG_SHIFT;
            if (qb.is_tandem) p.y |= PMX_SEED_TANDEM;
            a[n_a - (++n_rev)] = p;
        }
        A128 nt;
        if ((uint32_t)top.y < q.n - 1) {   // next occurrence of the same seed
            nt.y = top.y + 1;
            nt.x = pc[q.flt + (uint32_t)nt.y];
        } else {                            // list exhausted: the last heap element takes the root
            nt = heap[heap_size - 1];
            --heap_size;
        }
        top = heap_replace_root_min_x_c(heap, heap_size, nt, hc1, hc2);
    }
    // the reverse-strand block was filled back to front
    for (int64_t j = 0; j < n_rev >> 1; ++j) {
        const A128 t = a[n_a - 1 - j];
        a[n_a - 1 - j] = a[n_a - (n_rev - j)];
        a[n_a - (n_rev - j)] = t;
    }
    // (no seed is skipped on this path: MM_F_NO_DIAG/NO_DUAL/FOR_ONLY/REV_ONLY are never set, so
    //  n_for + n_rev == n_a; map.c:161-164 is a no-op)
}

// collect_seed_hits (map.c:168-204): the path of the reads that do not carry MM_F_HEAP_SORT -- the long-read presets
// (src/mm_align.c:167-180 leaves the flag clear).  Every occurrence becomes an anchor, seed by seed, and the list is sorted
// by x with radix_sort_128x, an UNSTABLE sort: anchors of equal x (one reference position hit by two minimizers of the
// read that have the same value) end up in an order that only the reference's procedure defines.
//   * host / scalar models: fill + the restated radix_sort_128x (aln_sort.hpp): the reference, step for step;
//   * wave kernels: the fill runs one seed per lane, the sort is a stable LSD radix sort across the wave (8-bit digits of
//     the reference position, then the strand bit; per-chunk ranks from ballots, bucket counters in the idle LDS DP area);
//     without equal keys every sort gives the same list, so the list is checked for adjacent equal x afterwards and only
//     then re-made by the exact sequential procedure.
PMX_HD A128 seed_anchor(const SeedA q, const SeedB qb, uint64_t r, int qlen) {
    const int32_t rpos = (int32_t)((uint32_t)r >> 1);
    A128 p;
    if ((r & 1) == (q.q_pos & 1)) {   // forward strand
        p.x = (r & 0xffffffff00000000ULL) | (uint32_t)rpos;
        p.y = (uint64_t)qb.q_span << 32 | q.q_pos >> 1;
    } else {                          // reverse strand: query position mirrored
        p.x = 1ULL << 63 | (r & 0xffffffff00000000ULL) | (uint32_t)rpos;
        p.y = (uint64_t)qb.q_span << 32 | (uint32_t)(qlen - ((int)(q.q_pos >> 1) + 1 - (int)qb.q_span) - 1);
    }
    p.y |= (uint64_t)qb.seg_id << PMX_SEED_SEG_SHIFT;
    if (qb.is_tandem) p.y |= PMX_SEED_TANDEM;
    return p;
}

PMX_HDN void collect_seed_hits_sorted(Work& W, const Opt& o, const RefIndex& ri, int qlen, int max_occ) {
    collect_matches(W, o, ri, qlen, max_occ);
    PMX_STAMP(W, 16);
    if (W.n_a > W.caps.max_anchor) {
        W.status |= PMX_ST_OVERFLOW;
        W.n_a = 0;
        return;
    }
    PMX_LDS(&W);
    const int n_m = W.n_seeds;
    const int64_t n_a = W.n_a;
    Ptr<A128> a = W.a; PMX_LDS(a);
    Ptr<SeedA> seeds = W.seeds; PMX_LDS(seeds);
    Ptr<SeedB> seeds_b = W.seeds_b; PMX_LDS(seeds_b);
#if PMX_W > 1 && !defined(PMX_ALL_LDS)
    const int lane = lane_id();
    bool exact_path = W.dp_fast == nullptr || n_a >= (1 << 24) || ri.len >= (1 << 30);
    if (!exact_path) {
        // anchor offsets: exclusive prefix sum of the occurrence counts (64 seeds per step)
        uint32_t acc = 0;
        for (int i0 = 0; i0 < n_m; i0 += PMX_W) {
            const int i = i0 + lane;
            const uint32_t n = i < n_m ? seeds[i].n : 0u;
            uint32_t incl = n;
            for (int d = 1; d < PMX_W; d <<= 1) { const uint32_t y = __shfl_up(incl, d); if (lane >= d) incl += y; }
            if (i < n_m) seeds[i].flt = acc + incl - n;   // flt is free now
            acc += __shfl(incl, PMX_W - 1);
        }
        wave_sync();
        for (int i = lane; i < n_m; i += PMX_W) {
            const SeedA q = seeds[i];
            const SeedB qb = seeds_b[i];
            for (uint32_t j = 0; j < q.n; ++j) a[q.flt + j] = seed_anchor(q, qb, ri.pos[q.off + j], qlen);
        }
        wave_sync();
        // stable LSD radix sort on key = strand << 31 | reference position
        uint32_t* cnt = reinterpret_cast<uint32_t*>(W.dp_fast);   // 256 counters
        Ptr<A128> src = a, dst = W.a2;
        int pos_bits = 1;
        while ((1 << pos_bits) < ri.len) ++pos_bits;
        const int n_pass = (pos_bits + 7) / 8 + 1;   // the last pass sorts by the strand bit
        for (int pass = 0; pass < n_pass; ++pass) {
            const bool strand_pass = pass == n_pass - 1;
            const int shift = pass * 8;
            for (int b = lane; b < 256; b += PMX_W) cnt[b] = 0;
            wave_sync();
            for (int64_t i = lane; i < n_a; i += PMX_W) {
                const uint64_t x = src[i].x;
                const uint32_t dgt = strand_pass ? (uint32_t)(x >> 63) : ((uint32_t)x >> shift) & 255u;
                atomicAdd(&cnt[dgt], 1u);
            }
            wave_sync();
            {   // exclusive scan of the 256 counters: four per lane
                uint32_t c0 = cnt[lane * 4], c1 = cnt[lane * 4 + 1], c2 = cnt[lane * 4 + 2], c3 = cnt[lane * 4 + 3];
                const uint32_t tot = c0 + c1 + c2 + c3;
                uint32_t incl = tot;
                for (int d = 1; d < PMX_W; d <<= 1) { const uint32_t y = __shfl_up(incl, d); if (lane >= d) incl += y; }
                uint32_t base = incl - tot;
                wave_sync();
                cnt[lane * 4] = base; base += c0;
                cnt[lane * 4 + 1] = base; base += c1;
                cnt[lane * 4 + 2] = base; base += c2;
                cnt[lane * 4 + 3] = base;
            }
            wave_sync();
            for (int64_t i0 = 0; i0 < n_a; i0 += PMX_W) {
                const int64_t i = i0 + lane;
                const bool act = i < n_a;
                A128 v;
                v.x = v.y = 0;
                if (act) v = src[i];
                const uint32_t dgt = !act ? 256u : strand_pass ? (uint32_t)(v.x >> 63) : ((uint32_t)v.x >> shift) & 255u;
                unsigned long long same = __ballot(act);   // lanes with my digit
                for (int b = 0; b < 8; ++b) {
                    const unsigned long long m = __ballot((dgt >> b) & 1u);
                    same &= ((dgt >> b) & 1u) ? m : ~m;
                }
                const unsigned long long below = same & ((1ULL << lane) - 1ULL);
                uint32_t at = 0;
                if (act) at = cnt[dgt] + (uint32_t)__builtin_popcountll(below);
                wave_sync();
                if (act) {
                    dst[at] = v;
                    if ((same >> lane) >> 1 == 0ULL) cnt[dgt] += (uint32_t)__builtin_popcountll(same);   // the group's last lane
                }
                wave_sync();
            }
            const Ptr<A128> t = src; src = dst; dst = t;
        }
        if (src != a) {
            for (int64_t i = lane; i < n_a; i += PMX_W) a[i] = src[i];
            wave_sync();
        }
        bool tie = false;
        for (int64_t i = lane; i + 1 < n_a; i += PMX_W) tie = tie || a[i].x == a[i + 1].x;
        exact_path = __ballot(tie) != 0ULL;
        wave_sync();
    }
    if (!exact_path) return;
#endif
    {   // the reference's procedure
        int64_t k = 0;
        for (int i = 0; i < n_m; ++i) {
            const SeedA q = seeds[i];
            const SeedB qb = seeds_b[i];
            for (uint32_t j = 0; j < q.n; ++j) a[k++] = seed_anchor(q, qb, ri.pos[q.off + j], qlen);
        }
        wave_sync();
        radix_sort_128x(a, a + n_a, &W.status);
    }
}

}  // namespace aln
}  // namespace pmx
