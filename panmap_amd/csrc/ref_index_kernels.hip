// ALIGN stage, reference side: the minimizer index of the placed genome (mm_idx_str, src/3rdparty/minimap2/index.c:
// 408-451, called once per sample at src/mm_align.c:192) built ON THE DEVICE.  The host build (align/aln_host.hpp
// build_ref_index: sequential sketch, sort, table, eight uploads) kept the GPU idle for ~1 ms of an 11 ms step between
// the placement result and the first align kernel; here the genome goes up as ASCII once and five small launches make
// the same index: encode -> sketch (one thread per 32-base slice, align/aln_seed.hpp sketch_slice) -> radix sort of
// (minimizer << 22 | position word) -> run starts / counts -> hash table.  One host round trip in the middle sizes the
// table by the number of distinct minimizers exactly as the host build does, so both builds describe the same index
// (same keys, same occurrence lists in the same order; slot order within a probe run may differ, lookups cannot).
#define PMX_THREAD_PER_PAIR 1
#include <hip/hip_runtime.h>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include <stdexcept>

#include "align/aln_host.hpp"
#include "device/dev_util.hpp"
#include "ref_index_device.h"

namespace pmx {
namespace aln {

namespace {
constexpr int kSlice = 32;        // bases per sketch thread (+ w + k + 1 of run-in)
constexpr int kPosBits = 22;      // position word (position << 1 | strand) in the sort key: references below 2^21 bases
constexpr uint64_t kNoKey = ~0ULL;

__device__ __forceinline__ uint8_t nt4_dev(unsigned char c) {   // seq_nt4_table (sketch.c:9-26)
    c &= 0xDF;   // upper case
    return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : (c == 'T' || c == 'U') ? 3 : 4;
}

// one thread per 32-base word: nt4 codes (seq, 8 bytes of padding), 2-bit packed words and their ambiguity masks
__global__ void k_ref_encode(const char* __restrict__ ascii, int len, uint8_t* seq, uint64_t* pk, uint64_t* pk_amb, int n_words) {
    const int wd = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (wd >= n_words) return;
    uint64_t p = 0, a = 0;
    for (int b = 0; b < 32; ++b) {
        const int i = wd * 32 + b;
        if (i < len) {
            const uint8_t c = nt4_dev((unsigned char)ascii[i]);
            seq[i] = c;
            if (c < 4) p |= (uint64_t)c << (2 * b);
            else a |= 1ULL << (2 * b);
        } else if (i < len + 8) seq[i] = 0;
    }
    pk[wd] = p;
    pk_amb[wd] = a;
}

struct RefBase {
    const uint8_t* seq;
    __device__ int operator()(int i) const { return (int)seq[i]; }
};
struct RefEmit {
    uint64_t* keys;
    unsigned long long* n_out;
    uint64_t cap;
    __device__ void operator()(uint64_t x, uint64_t y) const {
        const unsigned long long at = atomicAdd(n_out, 1ULL);
        if (at < cap) keys[at] = (x >> 8) << kPosBits | (y & ((1ULL << kPosBits) - 1));
    }
};

// one thread per slice of kSlice bases (the emission order is irrelevant: the list is sorted next)
__global__ void k_ref_sketch(const uint8_t* __restrict__ seq, int len, int w, int k, uint64_t* keys, uint64_t cap, unsigned long long* n_out) {
    const int s = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int64_t begin = (int64_t)s * kSlice;
    if (begin >= len) return;
    const int end = (int)(begin + kSlice < len ? begin + kSlice : len);
    RefBase base{seq};
    RefEmit emit{keys, n_out, cap};
    sketch_slice<12>((int)begin, end, len, w, k, base, emit);
}

// sorted keys -> occurrence list (pos) and, per distinct minimizer, one count; ctr[1] += distinct, ctr[2] += those with
// ten or more occurrences
__global__ void k_ref_runs(const uint64_t* __restrict__ sorted, int64_t n, uint64_t* pos, unsigned long long* ctr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = sorted[i];
    if (key == kNoKey) return;                 // past the last minimizer
    pos[i] = key & ((1ULL << kPosBits) - 1);   // y of a single reference: rid 0
    if (i > 0 && (sorted[i - 1] >> kPosBits) == (key >> kPosBits)) return;
    int64_t j = i + 1;
    while (j < n && (sorted[j] >> kPosBits) == (key >> kPosBits)) ++j;
    atomicAdd(&ctr[1], 1ULL);
    if (j - i >= 10) atomicAdd(&ctr[2], 1ULL);
}

// the run starts enter the open-addressing table (linear probing from mix64(key), index.c:81-99 restated as in
// build_ref_index); ht / ht_pv arrive filled with 0xff bytes
__global__ void k_ref_table(const uint64_t* __restrict__ sorted, int64_t n, HtEnt* ht, uint32_t* ht_pv, uint32_t mask) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t sk = sorted[i];
    const uint64_t key = sk >> kPosBits;
    if (i > 0 && (sorted[i - 1] >> kPosBits) == key) return;
    int64_t j = i + 1;
    while (j < n && (sorted[j] >> kPosBits) == key) ++j;
    uint32_t slot = (uint32_t)mix64(key) & mask;
    for (;;) {
        const unsigned long long prev = atomicCAS((unsigned long long*)&ht[slot].key, (unsigned long long)UINT64_MAX, (unsigned long long)key);
        if (prev == (unsigned long long)UINT64_MAX) break;
        slot = (slot + 1) & mask;
    }
    ht[slot].off = (uint32_t)i;
    ht[slot].cnt = (uint32_t)(j - i);
    if (j - i == 1) ht_pv[slot] = (uint32_t)(sk & ((1ULL << kPosBits) - 1));
}
}  // namespace

bool ref_index_device_supported(const Opt& o, int64_t ref_len) {
    return (o.k & 1) && 2 * o.k <= 64 - kPosBits && o.w >= 1 && o.w <= 12 && ref_len > 0 && ref_len < (1LL << (kPosBits - 1));
}

bool build_ref_index_device(hipStream_t st, const char* reference, int64_t ref_len, Opt& o, RefIndexDevice& d) {
    const int len = (int)ref_len;
    const int n_words = (len + 31) / 32 + 2;
    const uint64_t cap = (uint64_t)len + 16;   // every emitted minimizer is a distinct valid k-mer position
    d.ascii.ensure((size_t)len);
    d.seq.ensure((size_t)len + 8);
    d.pk.ensure((size_t)n_words);
    d.pk_amb.ensure((size_t)n_words);
    d.keys.ensure(cap);
    d.keys2.ensure(cap);
    d.pos.ensure(cap);
    d.ctr.ensure(4);
    PMX_HIP(hipMemcpyAsync(d.ascii.p, reference, (size_t)len, hipMemcpyHostToDevice, st));
    PMX_HIP(hipMemsetAsync(d.ctr.p, 0, 4 * sizeof(unsigned long long), st));
    PMX_HIP(hipMemsetAsync(d.keys.p, 0xff, cap * sizeof(uint64_t), st));
    hipLaunchKernelGGL(k_ref_encode, dim3((unsigned)((n_words + 127) / 128)), dim3(128), 0, st, d.ascii.p, len, d.seq.p, d.pk.p, d.pk_amb.p, n_words);
    const int n_slices = (len + kSlice - 1) / kSlice;
    hipLaunchKernelGGL(k_ref_sketch, dim3((unsigned)((n_slices + 63) / 64)), dim3(64), 0, st, d.seq.p, len, o.w, o.k, d.keys.p, cap, d.ctr.p);
    size_t bytes = 0;
    const unsigned end_bit = (unsigned)(2 * o.k + kPosBits);
    PMX_HIP(rocprim::radix_sort_keys(nullptr, bytes, d.keys.p, d.keys2.p, (size_t)cap, 0, end_bit, st));
    d.tmp.ensure(bytes);
    PMX_HIP(rocprim::radix_sort_keys(d.tmp.p, bytes, d.keys.p, d.keys2.p, (size_t)cap, 0, end_bit, st));
    // the unused tail of the key buffer is all ones: the largest value in the sorted bits, and no minimizer has every
    // position bit set (ref_index_device_supported), so the sentinels sort behind the minimizers and k_ref_runs skips them
    hipLaunchKernelGGL(k_ref_runs, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, st, d.keys2.p, (int64_t)cap, d.pos.p, d.ctr.p);
    unsigned long long h[4] = {0, 0, 0, 0};
    PMX_HIP(hipMemcpyAsync(h, d.ctr.p, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    PMX_HIP(hipStreamSynchronize(st));
    const int64_t n_mv = (int64_t)h[0];
    if ((uint64_t)n_mv > cap) throw std::runtime_error("reference sketch emitted more minimizers than positions");
    const size_t n_keys = (size_t)h[1], n_ge10 = (size_t)h[2];
    if (o.mid_occ <= 0) {
        // the threshold is the kk-th smallest occurrence count + 1, clamped from below at 10 (50: map-hifi): when more
        // than kk distinct minimizers occur fewer than ten times that count is below ten and the clamp decides
        if (n_keys > 0 && n_keys - n_ge10 <= mid_occ_rank(n_keys)) return false;   // repeat-rich reference: the host build computes it exactly
        set_mid_occ(o, n_keys ? 10 : INT32_MAX);
    }
    size_t tcap = 16;
    while (tcap < n_keys * 2 + 2) tcap <<= 1;
    d.ht.ensure(tcap);
    d.ht_pv.ensure(tcap);
    PMX_HIP(hipMemsetAsync(d.ht.p, 0xff, tcap * sizeof(HtEnt), st));
    PMX_HIP(hipMemsetAsync(d.ht_pv.p, 0xff, tcap * sizeof(uint32_t), st));
    if (n_mv > 0) hipLaunchKernelGGL(k_ref_table, dim3((unsigned)((n_mv + 255) / 256)), dim3(256), 0, st, d.keys2.p, n_mv, d.ht.p, d.ht_pv.p, (uint32_t)(tcap - 1));
    PMX_HIP(hipGetLastError());
    d.ht_mask = (uint32_t)(tcap - 1);
    d.n_mv = n_mv;
    d.n_keys = (int64_t)n_keys;
    o.ref_len = len;
    return true;
}

}  // namespace aln
}  // namespace pmx
