// C ABI, device part 4: --meta (haplotype deconvolution of a mixed sample; BASELINE config 5).
//
// What the reference does (src/main.cpp:1192-1313 runDeconvolution; src/mgsr.cpp):
//   1. every read becomes its list of k-min-mers ("seedmers": hash + orientation), identical lists are merged with a
//      multiplicity (initializeQueryData, mgsr.cpp:1774-2237);
//   2. every node gets an overlap coefficient: distinct read seedmer hashes its genome holds / its distinct seedmers
//      (computeOverlapCoefficients, :5685-5790); the nodes of the best `top_oc` distinct values are the candidates
//      (squareEM::squareEM, :8010-8060);
//   3. every (read, candidate) pair gets a parsimony score: with f = the read's seedmers the node's genome holds in the
//      read's orientation and r = those it holds in the other one, the score is max(f, r) (scoreReadsHelper, :7225-7455,
//      a DFS that applies and reverts per-node seed deltas);
//   4. candidates with identical score columns are merged, P(read | node) = err^(n - s) * (1 - err)^s with n = the read's
//      seedmers, and a SQUAREM-accelerated EM estimates the mixture proportions (:8100-8160, :4341-4443); nodes below
//      0.5 % are dropped and the EM is run again, up to five rounds (:4445-4490, main.cpp:1263-1271).
// Here, MI355X-first:
//   * the node side is the ORIENTED seed index (pmx_index_build_ex mode | PMX_INDEX_ORIENTED): a count-change index like
//     the place stage's, keyed hash ^ PMX_ORIENT_XOR for right-to-left k-min-mers.  Nodes are in DFS pre-order, so "seedmer
//     becomes present / absent at node v" holds for the contiguous index range [v, end(v)] of v's subtree: presence at any
//     node is the XOR of the ranges that cover it.  No tree walk, no per-node state: k_meta_mask_events toggles, for every
//     transition of a seedmer the reads carry, a bit range of that seedmer's row in a (seedmer x candidate) bit matrix.
//   * k_meta_scores: one thread per (read, 64 candidates).  The read's seedmers are added as bit planes (a carry-save
//     counter per candidate bit: 64 candidates advance with a handful of 64-bit ops per seedmer), for both orientations;
//     max(f, r) per candidate goes out as a 16-bit score.
//   * step 2 reuses the place stage: the overlap coefficient's two counts are its per-node intersection / genome seed counts.
//   * the EM runs on the score matrix, P(read | node) gathered from a table of the few distinct (n, s) pairs (computed on
//     the host with the libm the oracle uses), FP64 throughout, every reduction in a fixed order (bit-stable runs):
//     k_meta_denoms (a wave per read), k_meta_colsum (a thread per candidate over a chunk of reads; chunk partials added in
//     chunk order).  HBM-bound: 2 B per (read, candidate) per pass, six passes per SQUAREM iteration.
// Read-side seedmer extraction is host C++ here (threads; the k-min-mer definitions of host/seed_host.hpp); moving it into
// the seeding kernel is the next step.  parity: the reference's own MGSR index and EM cannot be built here (panman / TBB /
// Eigen / abseil are absent): scores and EM are checked by the test suite against a direct restatement (per-node seed
// sets by walking the tree, numpy EM), the end result against the reference's e2e expectation on rsv_4K (70 / 30 mixture
// recovered within its ranges, src/test/e2e/run_e2e.sh:182-204).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <numeric>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "device/dev_util.hpp"
#include "host/index_build.hpp"
#include "host/seed_host.hpp"
#include "place_kernels.h"
#include "readset.hpp"

using namespace pmx;

namespace {
int fail(int code, const std::string& msg) {
    set_error(msg);
    return code;
}
#define PMX_TRY try {
#define PMX_CATCH                                                      \
    }                                                                  \
    catch (const HipError& e) { return fail(PMX_ERR_DEVICE, e.msg); }  \
    catch (const std::exception& e) { return fail(PMX_ERR_DEVICE, e.what()); }

// index of `key` in the ascending array keys[0..n), or -1
__device__ __forceinline__ int64_t find_sorted(const uint64_t* __restrict__ keys, int64_t n, uint64_t key) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo < n && keys[lo] == key ? lo : -1;
}

// first position in the ascending array a[0..n) whose value is >= v
__device__ __forceinline__ int lower_bound_u32(const uint32_t* __restrict__ a, int n, uint32_t v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// Every count change of the oriented index whose seedmer becomes present (parent count 0) or absent (child count 0) and
// whose hash the reads carry toggles the candidates inside the node's subtree in that seedmer's row: bits [lo, hi) of
// mask[orientation][uid].  Candidates are sorted by DFS index, so the subtree is one bit range.
__global__ void k_meta_mask_events(const uint64_t* __restrict__ ch_key, const int16_t* __restrict__ ch_pc, const int16_t* __restrict__ ch_cc,
                                   const uint32_t* __restrict__ ch_node, int64_t n_changes, const uint32_t* __restrict__ subtree_end,
                                   const uint64_t* __restrict__ uniq, int64_t n_uniq, const uint32_t* __restrict__ cand_dfs, int n_cand, int words,
                                   unsigned long long* mask_fwd, unsigned long long* mask_rev) {
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n_changes; c += (int64_t)gridDim.x * blockDim.x) {
        const bool was = ch_pc[c] > 0, is = ch_cc[c] > 0;
        if (was == is) continue;
        const uint64_t key = ch_key[c];
        int64_t uid = find_sorted(uniq, n_uniq, key);
        unsigned long long* row = mask_fwd;
        if (uid < 0) { uid = find_sorted(uniq, n_uniq, key ^ PMX_ORIENT_XOR); row = mask_rev; }
        if (uid < 0) continue;
        const uint32_t v = ch_node[c];
        const int lo = lower_bound_u32(cand_dfs, n_cand, v), hi = lower_bound_u32(cand_dfs, n_cand, subtree_end[v] + 1u);
        if (hi <= lo) continue;
        row += (size_t)uid * (size_t)words;
        for (int w = lo >> 6; w <= (hi - 1) >> 6; ++w) {
            const int b0 = w == (lo >> 6) ? (lo & 63) : 0, b1 = w == ((hi - 1) >> 6) ? ((hi - 1) & 63) : 63;
            const unsigned long long bits = (b1 == 63 ? ~0ULL : ((1ULL << (b1 + 1)) - 1ULL)) & ~((1ULL << b0) - 1ULL);
            atomicXor(&row[w], bits);
        }
    }
}

// bit-sliced counters: plane[p] holds bit p of 64 independent counts; add one 64-bit row of 0/1 increments
template <int PLANES>
__device__ __forceinline__ void planes_add(unsigned long long (&plane)[PLANES], unsigned long long inc) {
#pragma unroll
    for (int p = 0; p < PLANES; ++p) {
        const unsigned long long carry = plane[p] & inc;
        plane[p] ^= inc;
        inc = carry;
    }
}

// One thread per (read, word of 64 candidates).  seed_uid / seed_rev: the read's seedmers as row indices + orientation.
// score[read][cand] = max(#seedmers present in the read's orientation, #present in the other one).
template <int PLANES>
__global__ void k_meta_scores(const int64_t* __restrict__ read_off, const uint32_t* __restrict__ seed_uid, const uint8_t* __restrict__ seed_rev,
                              int64_t n_reads, const unsigned long long* __restrict__ mask_fwd, const unsigned long long* __restrict__ mask_rev,
                              int words, int n_cand, uint16_t* __restrict__ score) {
    const int64_t total = n_reads * (int64_t)words;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / words;
        const int w = (int)(t - r * words);
        unsigned long long same[PLANES], other[PLANES];
#pragma unroll
        for (int p = 0; p < PLANES; ++p) { same[p] = 0; other[p] = 0; }
        for (int64_t i = read_off[r]; i < read_off[r + 1]; ++i) {
            const size_t row = (size_t)seed_uid[i] * (size_t)words + (size_t)w;
            const unsigned long long f = mask_fwd[row], b = mask_rev[row];
            const bool rev = seed_rev[i] != 0;
            planes_add<PLANES>(same, rev ? b : f);    // the genome holds it the way the read does
            planes_add<PLANES>(other, rev ? f : b);
        }
        const int c0 = w * 64;
        for (int b = 0; b < 64 && c0 + b < n_cand; ++b) {
            uint32_t s = 0, o = 0;
#pragma unroll
            for (int p = 0; p < PLANES; ++p) { s |= (uint32_t)((same[p] >> b) & 1ULL) << p; o |= (uint32_t)((other[p] >> b) & 1ULL) << p; }
            score[(size_t)r * (size_t)n_cand + (size_t)(c0 + b)] = (uint16_t)(s > o ? s : o);
        }
    }
}

// 128-bit digest of every candidate's score column (candidates with equal columns are merged before the EM)
__global__ void k_meta_column_digest(const uint16_t* __restrict__ score, int64_t n_reads, int n_cand, uint64_t* digest) {
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n_cand; c += gridDim.x * blockDim.x) {
        uint64_t a = 0x9e3779b97f4a7c15ULL, b = 0xc2b2ae3d27d4eb4fULL;
        for (int64_t r = 0; r < n_reads; ++r) {
            const uint64_t v = (uint64_t)score[(size_t)r * (size_t)n_cand + (size_t)c] + 1ULL;
            a = (a ^ v) * 0xff51afd7ed558ccdULL; a ^= a >> 29;
            b = (b + v * 0x9e3779b97f4a7c15ULL) * 0xc4ceb9fe1a85ec53ULL; b ^= b >> 31;
        }
        digest[2 * (size_t)c] = a;
        digest[2 * (size_t)c + 1] = b;
    }
}

// EM pass 1: denom[j] = 1 / sum_i P(j, i) * props[i] over the kept columns, in column order (lanes stride the columns, a fixed
// butterfly adds the lanes); llh[j] = weight[j] * log(denom[j]).  A wave per read.
__global__ void k_meta_denoms(const uint16_t* __restrict__ score, int n_cand, const int* __restrict__ cols, int n_cols, const double* __restrict__ props,
                              const int64_t* __restrict__ rows, int64_t n_rows, const uint32_t* __restrict__ tab_off, const double* __restrict__ tab,
                              const double* __restrict__ weight, double* denom, double* llh, const int* done) {
    if (done && *done) return;   // (the SQUAREM loop runs ahead of the host: launches queued past convergence do nothing)
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t j = wave; j < n_rows; j += n_waves) {
        const int64_t r = rows[j];
        const uint16_t* srow = score + (size_t)r * (size_t)n_cand;
        const double* t = tab + tab_off[j];
        double acc = 0.0;
        // (same order of additions; four columns' loads are in flight at a time -- each entry is a chain of three dependent loads,
        //  and one at a time the pass ran at an eighth of the HBM rate)
        int i = lane;
        for (; i + 192 < n_cols; i += 256) {
            const int c0 = cols[i], c1 = cols[i + 64], c2 = cols[i + 128], c3 = cols[i + 192];
            const uint16_t s0 = srow[c0], s1 = srow[c1], s2 = srow[c2], s3 = srow[c3];
            const double p0 = props[i], p1 = props[i + 64], p2 = props[i + 128], p3 = props[i + 192];
            const double t0 = t[s0], t1 = t[s1], t2 = t[s2], t3 = t[s3];
            acc += t0 * p0;
            acc += t1 * p1;
            acc += t2 * p2;
            acc += t3 * p3;
        }
        for (; i < n_cols; i += 64) acc += t[srow[cols[i]]] * props[i];
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) { denom[j] = 1.0 / acc; llh[j] = weight[j] * log(acc); }   // (the reciprocal: pass 2 multiplies every entry of the row by it -- a division per entry before)
    }
}

// EM pass 2: part[chunk][i] = sum over the chunk's reads j (in order) of weight[j] * (P(j, i) * props[i] * (1 / denom[j])).
// A thread per column; the chunks are added in chunk order by k_meta_fold.
__global__ void k_meta_colsum(const uint16_t* __restrict__ score, int n_cand, const int* __restrict__ cols, int n_cols, const double* __restrict__ props,
                              const int64_t* __restrict__ rows, int64_t n_rows, int64_t chunk, const uint32_t* __restrict__ tab_off,
                              const double* __restrict__ tab, const double* __restrict__ weight, const double* __restrict__ denom, double* part,
                              const int* done) {
    if (done && *done) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cols) return;
    const int64_t j0 = (int64_t)blockIdx.y * chunk, j1 = j0 + chunk < n_rows ? j0 + chunk : n_rows;
    const int c = cols[i];
    const double pi = props[i];
    double acc = 0.0;
    int64_t j = j0;
    for (; j + 3 < j1; j += 4) {   // (same order of additions, four reads' loads in flight)
        const int64_t r0 = rows[j], r1 = rows[j + 1], r2 = rows[j + 2], r3 = rows[j + 3];
        const uint32_t o0 = tab_off[j], o1 = tab_off[j + 1], o2 = tab_off[j + 2], o3 = tab_off[j + 3];
        const uint16_t s0 = score[(size_t)r0 * (size_t)n_cand + (size_t)c], s1 = score[(size_t)r1 * (size_t)n_cand + (size_t)c],
                       s2 = score[(size_t)r2 * (size_t)n_cand + (size_t)c], s3 = score[(size_t)r3 * (size_t)n_cand + (size_t)c];
        const double t0 = tab[o0 + s0], t1 = tab[o1 + s1], t2 = tab[o2 + s2], t3 = tab[o3 + s3];
        acc += weight[j] * (t0 * pi * denom[j]);
        acc += weight[j + 1] * (t1 * pi * denom[j + 1]);
        acc += weight[j + 2] * (t2 * pi * denom[j + 2]);
        acc += weight[j + 3] * (t3 * pi * denom[j + 3]);
    }
    for (; j < j1; ++j) acc += weight[j] * (tab[tab_off[j] + score[(size_t)rows[j] * (size_t)n_cand + (size_t)c]] * pi * denom[j]);   // denom = 1 / denominator
    part[(size_t)blockIdx.y * (size_t)n_cols + (size_t)i] = acc;
}

// the seedmer lists of a chunk of reads from their slot stretches (k_seed_histogram, list mode: entry e of read r at
// woff[r] * 32 + e) to contiguous arrays; eight lanes per read
__global__ void __launch_bounds__(256) k_meta_gather_lists(const uint64_t* __restrict__ list_hash, const uint8_t* __restrict__ list_rev,
                                                          const int64_t* __restrict__ woff, const int64_t* __restrict__ out_off, int64_t n_reads,
                                                          uint64_t* __restrict__ out_hash, uint8_t* __restrict__ out_rev) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, n_thr = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = gid >> 3; r < n_reads; r += n_thr >> 3) {
        const int64_t src = woff[r] * 32, dst = out_off[r], n = out_off[r + 1] - dst;
        for (int64_t e = gid & 7; e < n; e += 8) { out_hash[dst + e] = list_hash[src + e]; out_rev[dst + e] = list_rev[src + e]; }
    }
}

__global__ void k_meta_fold(const double* __restrict__ part, int n_chunks, int n_cols, double scale, double* out, const int* done) {
    if (done && *done) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cols) return;
    double acc = 0.0;
    int c = 0;
    for (; c + 8 <= n_chunks; c += 8) {   // (same order of additions, eight loads in flight)
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(c + u) * (size_t)n_cols + (size_t)i];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; c < n_chunks; ++c) acc += part[(size_t)c * (size_t)n_cols + (size_t)i];
    out[i] = acc * scale;
}

// sums of 1,024 contiguous values each, a wave per block in a FIXED order (lane l adds v[l], v[l + 64], ... in order, a fixed
// butterfly adds the lanes); the host adds the block sums in order.  (One thread per block took 104 us per likelihood.)
__global__ void k_meta_sum_blocks(const double* __restrict__ v, int64_t n, double* block_sums, const int* done) {
    if (done && *done) return;
    const int lane = threadIdx.x & 63;
    const int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t lo = b * 1024, hi = lo + 1024 < n ? lo + 1024 : n;
    if (lo >= n) return;
    double acc = 0.0;
    for (int64_t i = lo + lane; i < hi; i += 64) acc += v[i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) block_sums[b] = acc;
}

// ---- the SQUAREM iteration's vector arithmetic on the device (runSquareEM, src/mgsr.cpp:4394-4443).  The vectors have a few
// thousand entries; every sum runs in index order on one thread, exactly as the host loop they replace did, so the results are
// the same numbers -- what is gone is four host round trips per iteration.  One block of 256 threads each.
struct EmCtl {
    double llh, llh2, llh_sq, difference;
    int done, iterations;
};
// v <- v with non-positive entries raised to 1e-12, divided by its sum (normalizeProps, :4374-4383); src may equal dst
// (the in-order sums run on one thread over a copy of the vector in LDS: straight from global memory a dependent load per
//  element made each of them ~130 us)
__device__ __forceinline__ double em_sum_in_order(const double* lds_v, int n) {
    double s = 0.0;
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        const double a0 = lds_v[i], a1 = lds_v[i + 1], a2 = lds_v[i + 2], a3 = lds_v[i + 3], a4 = lds_v[i + 4], a5 = lds_v[i + 5], a6 = lds_v[i + 6], a7 = lds_v[i + 7];
        s += a0; s += a1; s += a2; s += a3; s += a4; s += a5; s += a6; s += a7;
    }
    for (; i < n; ++i) s += lds_v[i];
    return s;
}
__global__ void k_em_normalize(const double* src, double* dst, int n, const EmCtl* ctl, double* work) {
    if (ctl->done) return;
    extern __shared__ double em_lds_[];
    double* em_lds = work ? work : em_lds_;   // (more columns than LDS holds: a scratch vector in global memory)
    __shared__ double s_sum;
    for (int i = threadIdx.x; i < n; i += blockDim.x) { const double x = src[i]; em_lds[i] = x <= 0 ? 1e-12 : x; }
    __syncthreads();
    if (threadIdx.x == 0) s_sum = em_sum_in_order(em_lds, n);
    __syncthreads();
    const double s = s_sum;
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = em_lds[i] / s;
}
__global__ void k_em_copy(const double* src, double* dst, int n, const EmCtl* ctl) {
    if (ctl->done) return;
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}
// the squared-extrapolation point: alpha = -|r| / |v|, sq = p0 - 2 alpha r + alpha^2 v, normalised
__global__ void k_em_extrapolate(const double* p0, const double* p1, const double* p2, double* sq, int n, const EmCtl* ctl, double* work) {
    if (ctl->done) return;
    extern __shared__ double em_lds_[];   // r*r, then v*v, then the extrapolated point: n entries each
    double* em_lds = work ? work : em_lds_;
    __shared__ double s_alpha, s_sum;
    double* rr = em_lds;
    double* vv = em_lds + n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double r = p1[i] - p0[i], v = (p2[i] - p1[i]) - r;
        rr[i] = r * r;
        vv[i] = v * v;
    }
    __syncthreads();
    if (threadIdx.x == 0) s_alpha = -sqrt(em_sum_in_order(rr, n)) / sqrt(em_sum_in_order(vv, n));
    __syncthreads();
    const double alpha = s_alpha;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double r = p1[i] - p0[i], v = (p2[i] - p1[i]) - r;
        const double x = p0[i] - 2.0 * alpha * r + alpha * alpha * v;
        rr[i] = x <= 0 ? 1e-12 : x;
    }
    __syncthreads();
    if (threadIdx.x == 0) s_sum = em_sum_in_order(rr, n);
    __syncthreads();
    const double s = s_sum;
    for (int i = threadIdx.x; i < n; i += blockDim.x) sq[i] = rr[i] / s;
}
// the block sums of a likelihood pass, added in order (getExp, :4385-4388): which = 0 -> llh2, 1 -> llh_sq
__global__ void k_em_store_llh(const double* bsum, int64_t n_bsum, EmCtl* ctl, int which) {
    if (ctl->done || threadIdx.x != 0) return;
    double s = 0.0;
    for (int64_t i = 0; i < n_bsum; ++i) s += bsum[i];
    if (which == 0) ctl->llh2 = s; else ctl->llh_sq = s;
}
// keep the extrapolated point unless it lost likelihood; count the iteration; test convergence (:4424-4441)
__global__ void k_em_choose(const double* p0, const double* p2, const double* sq, double* props, int n, EmCtl* ctl, double convergence, double delta_threshold) {
    if (ctl->done) return;
    __shared__ int s_take_sq;
    if (threadIdx.x == 0) s_take_sq = ctl->llh_sq > ctl->llh2 - convergence ? 1 : 0;
    __syncthreads();
    const double* from = s_take_sq ? sq : p2;
    for (int i = threadIdx.x; i < n; i += blockDim.x) props[i] = from[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        const double now = s_take_sq ? ctl->llh_sq : ctl->llh2;
        const double difference = now - ctl->llh;
        ctl->difference = difference;
        ctl->llh = now;
        ++ctl->iterations;
        if (delta_threshold == 0) {
            if (fabs(difference) < convergence) ctl->done = 1;
        } else {
            double mc = 0.0;
            for (int i = 0; i < n; ++i) { const double d = fabs(from[i] - p0[i]); mc = mc > d ? mc : d; }   // (a maximum: any order)
            if (mc < delta_threshold) ctl->done = 1;
        }
    }
}
}  // namespace

struct pmx_meta_group {
    uint32_t node;                   // representative (lowest DFS index of the group)
    std::vector<uint32_t> members;   // the other candidates with the same score column
    double prop = 0.0;
};

struct pmx_meta {
    pmx_ctx* ctx = nullptr;
    const pmx_index* idx_std = nullptr;
    int64_t n_nodes = 0, n_changes = 0;
    SyncmerParams params;
    // oriented index on the device
    DevBuf<uint64_t> ch_key;
    DevBuf<int16_t> ch_pc, ch_cc;
    DevBuf<uint32_t> ch_node, subtree_end;
    pmx_place* placer = nullptr;
    // reads (merged by seedmer list)
    int64_t n_raw_reads = 0, n_reads = 0, n_seedmers = 0;
    std::vector<int64_t> h_read_off;
    std::vector<uint64_t> h_seed_hash;
    std::vector<uint8_t> h_seed_rev;
    std::vector<int64_t> h_mult;
    std::vector<uint64_t> h_uniq;
    DevBuf<int64_t> d_read_off;
    DevBuf<uint32_t> d_seed_uid;
    DevBuf<uint8_t> d_seed_rev;
    DevBuf<uint64_t> d_uniq;
    std::vector<double> oc;                 // per node
    // candidates
    std::vector<uint32_t> cand;             // DFS indices, ascending
    DevBuf<uint32_t> d_cand;
    DevBuf<unsigned long long> mask_fwd, mask_rev;
    DevBuf<uint16_t> score;                 // [n_reads][n_cand]
    std::vector<int32_t> h_max_score;       // per read, over the candidates
    // result
    std::vector<pmx_meta_group> groups;     // sorted by proportion, descending
    int em_rounds = 0, em_iterations = 0;
    double llh = 0.0;
    double dust_threshold = 100.0;          // --dust: 100 = no filter
    int64_t n_dust_dropped = 0;
};

extern "C" {

int pmx_meta_create(pmx_ctx* ctx, const pmx_index* idx_std, const pmx_index* idx_oriented, pmx_meta** out) {
    if (!ctx || !idx_std || !idx_oriented || !out) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    const LiteIndex* S = pmx_index_internal(idx_std);
    const LiteIndex* O = pmx_index_internal(idx_oriented);
    if (!O->params.oriented) return fail(PMX_ERR_ARG, "the second index must be built with PMX_INDEX_ORIENTED");
    if (S->params.oriented || S->parent != O->parent || S->params.k != O->params.k || S->params.s != O->params.s || S->params.l != O->params.l ||
        S->params.t != O->params.t || S->params.open != O->params.open)
        return fail(PMX_ERR_ARG, "the two indexes must cover the same tree with the same seeding parameters");
    std::unique_ptr<pmx_meta> m(new pmx_meta());
    m->ctx = ctx;
    m->idx_std = idx_std;
    m->params = O->params;
    const int64_t n = (int64_t)O->parent.size(), c = (int64_t)O->hash.size();
    m->n_nodes = n;
    m->n_changes = c;
    // nodes are in DFS pre-order: the subtree of v is the index range [v, end(v)]
    std::vector<uint32_t> end((size_t)n);
    for (int64_t v = 0; v < n; ++v) end[v] = (uint32_t)v;
    for (int64_t v = n - 1; v > 0; --v) end[O->parent[v]] = std::max(end[O->parent[v]], end[v]);
    std::vector<uint32_t> node((size_t)c);
    for (int64_t v = 0; v < n; ++v)
        for (uint64_t q = O->offsets[v]; q < O->offsets[v + 1]; ++q) node[q] = (uint32_t)v;
    m->ch_key.alloc((size_t)c); m->ch_pc.alloc((size_t)c); m->ch_cc.alloc((size_t)c); m->ch_node.alloc((size_t)c); m->subtree_end.alloc((size_t)n);
    if (c > 0) {
        PMX_HIP(hipMemcpy(m->ch_key.p, O->hash.data(), sizeof(uint64_t) * (size_t)c, hipMemcpyHostToDevice));
        PMX_HIP(hipMemcpy(m->ch_pc.p, O->parent_count.data(), sizeof(int16_t) * (size_t)c, hipMemcpyHostToDevice));
        PMX_HIP(hipMemcpy(m->ch_cc.p, O->child_count.data(), sizeof(int16_t) * (size_t)c, hipMemcpyHostToDevice));
        PMX_HIP(hipMemcpy(m->ch_node.p, node.data(), sizeof(uint32_t) * (size_t)c, hipMemcpyHostToDevice));
    }
    PMX_HIP(hipMemcpy(m->subtree_end.p, end.data(), sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice));
    const int rc = pmx_place_create(ctx, idx_std, &m->placer);
    if (rc != PMX_OK) return rc;
    *out = m.release();
    return PMX_OK;
    PMX_CATCH
}

void pmx_meta_free(pmx_ctx* ctx, pmx_meta* m) {
    if (!m) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    if (m->placer) pmx_place_free(ctx, m->placer);
    delete m;
}

// Step 1 + 2: the reads' seedmer lists (host threads), merged by list; the overlap coefficient of every node (place stage).
int pmx_meta_set_reads(pmx_ctx* ctx, pmx_meta* m, const char* concat, const int64_t* offsets, int64_t n_reads) {
    if (!ctx || !m || !offsets || n_reads < 0 || (!concat && n_reads > 0)) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    const SyncmerParams p = m->params;
    const int l = p.l;
    // ---- the reads' seedmer lists ON THE DEVICE (round 4; src/mgsr.cpp:1774-2237): the reads are packed 2 bit/base like any
    // read set and the generic seeding kernel runs in its list mode (k_seed_histogram: same syncmers, same k-min-mers as
    // the place stage, orientation = R < F); a gather makes the lists contiguous.  Chunks of reads bound the list arrays
    // (one slot per base).  The host only filters by DUST (integer state, a few operations per base) and merges equal lists.
    const double dust_thr = m->dust_threshold;
    std::vector<uint8_t> dusty(dust_thr < 100.0 ? (size_t)n_reads : 0, 0);
    if (dust_thr < 100.0) {   // src/mgsr.cpp:1593-1594: a read with a non-zero score above the threshold is left out
        unsigned n_thr = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        if (n_reads < 4096) n_thr = 1;
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < n_thr; ++t)
            pool.emplace_back([&, t]() {
                for (int64_t r = (int64_t)t; r < n_reads; r += n_thr) {
                    const double d = pmx_read_dust(concat + offsets[r], offsets[r + 1] - offsets[r], 64);
                    if (d != 0 && d > dust_thr) dusty[(size_t)r] = 1;
                }
            });
        for (auto& th : pool) th.join();
    }
    // the dusty reads leave the sample altogether (src/mgsr.cpp:1590-1597: they never enter seqToIndexVec), the overlap
    // coefficients included: everything below sees the kept reads only
    const int64_t n_raw = n_reads;
    int64_t n_dusty = 0;
    for (uint8_t d : dusty) n_dusty += d;
    std::string kept_concat;
    std::vector<int64_t> kept_off;
    if (n_dusty > 0) {
        kept_off.push_back(0);
        for (int64_t r = 0; r < n_reads; ++r) {
            if (dusty[(size_t)r]) continue;
            kept_concat.append(concat + offsets[r], (size_t)(offsets[r + 1] - offsets[r]));
            kept_off.push_back((int64_t)kept_concat.size());
        }
        concat = kept_concat.data();
        offsets = kept_off.data();
        n_reads = (int64_t)kept_off.size() - 1;
        dusty.clear();
    }
    std::vector<int64_t> r_off((size_t)n_reads + 1, 0);   // flat seedmer lists of the raw reads
    std::vector<uint64_t> r_hash;
    std::vector<uint8_t> r_rev;
    {
        SeedParams sp;
        sp.k = p.k; sp.s = p.s; sp.t = p.t; sp.l = l; sp.open = p.open ? 1 : 0; sp.trim_start = 0; sp.trim_end = 0;
        const int w = sp.k - sp.s + 1;
        const size_t lds = (size_t)(2 * w + l) * PMX_SEED_BLOCK * sizeof(uint64_t) + (size_t)(PMX_SEED_BLOCK / 64) * (PMX_SEED_QCAP * sizeof(uint64_t) + 8);
        if (lds > 160 * 1024) return fail(PMX_ERR_UNSUPPORTED, "k-s+1 too large for the LDS ring");
        if (lds > 64 * 1024) PMX_HIP(hipFuncSetAttribute((const void*)k_seed_histogram, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        DevBuf<uint64_t> d_lh, d_oh;
        DevBuf<uint8_t> d_lr, d_or, d_keep;
        DevBuf<uint32_t> d_ln;
        DevBuf<int64_t> d_ooff;
        DevBuf<unsigned long long> d_ctr;
        d_ctr.alloc(PMX_CTR_N);
        PMX_HIP(hipMemsetAsync(d_ctr.p, 0, sizeof(unsigned long long) * PMX_CTR_N, ctx->stream));
        const int64_t chunk = 2000000;
        for (int64_t c0 = 0; c0 < n_reads; c0 += chunk) {
            const int64_t c1 = std::min(n_reads, c0 + chunk), nc = c1 - c0;
            pmx_readset* rs = nullptr;
            int rc = pmx_readset_upload(ctx, concat, offsets + c0, nc, &rs);
            if (rc != PMX_OK) return rc;
            std::unique_ptr<pmx_readset, void (*)(pmx_readset*)> rs_guard(rs, [](pmx_readset* x) { delete x; });
            rc = pmx_readset_pack(ctx, rs);
            if (rc != PMX_OK) return rc;
            const size_t slots = (size_t)std::max<int64_t>(rs->n_words, 1) * 32;
            d_lh.ensure(slots); d_lr.ensure(slots); d_ln.ensure((size_t)nc);
            PMX_HIP(hipMemsetAsync(d_ln.p, 0, sizeof(uint32_t) * (size_t)nc, ctx->stream));
            const uint8_t* keep = nullptr;
            if (!dusty.empty()) {
                std::vector<uint8_t> k8((size_t)nc);
                for (int64_t i = 0; i < nc; ++i) k8[(size_t)i] = dusty[(size_t)(c0 + i)] ? 0 : 1;
                d_keep.ensure((size_t)nc);
                PMX_HIP(hipMemcpyAsync(d_keep.p, k8.data(), (size_t)nc, hipMemcpyHostToDevice, ctx->stream));
                PMX_HIP(hipStreamSynchronize(ctx->stream));   // (k8 goes out of scope)
                keep = d_keep.p;
            }
            hipLaunchKernelGGL(k_seed_histogram, dim3(grid_for(nc, PMX_SEED_BLOCK, ctx->n_cu * 16)), dim3(PMX_SEED_BLOCK), lds, ctx->stream, rs->words.p, rs->amb.p,
                               rs->woff.p, rs->off.p, (int64_t)0, nc, sp, (uint64_t*)nullptr, (unsigned long long*)nullptr, (uint64_t)0, d_ctr.p, keep,
                               (const uint8_t*)nullptr, 0, d_lh.p, d_lr.p, d_ln.p);
            PMX_HIP(hipGetLastError());
            std::vector<uint32_t> h_n((size_t)nc);
            PMX_HIP(hipMemcpyAsync(h_n.data(), d_ln.p, sizeof(uint32_t) * (size_t)nc, hipMemcpyDeviceToHost, ctx->stream));
            PMX_HIP(hipStreamSynchronize(ctx->stream));
            std::vector<int64_t> o_off((size_t)nc + 1, 0);
            for (int64_t i = 0; i < nc; ++i) o_off[(size_t)i + 1] = o_off[(size_t)i] + (int64_t)h_n[(size_t)i];
            const int64_t tot = o_off[(size_t)nc];
            const size_t base = r_hash.size();
            r_hash.resize(base + (size_t)tot);
            r_rev.resize(base + (size_t)tot);
            if (tot > 0) {
                d_ooff.ensure((size_t)nc + 1); d_oh.ensure((size_t)tot); d_or.ensure((size_t)tot);
                PMX_HIP(hipMemcpyAsync(d_ooff.p, o_off.data(), sizeof(int64_t) * ((size_t)nc + 1), hipMemcpyHostToDevice, ctx->stream));
                hipLaunchKernelGGL(k_meta_gather_lists, dim3(grid_for(nc * 8, 256, ctx->n_cu * 16)), dim3(256), 0, ctx->stream, d_lh.p, d_lr.p, rs->woff.p, d_ooff.p, nc,
                                   d_oh.p, d_or.p);
                PMX_HIP(hipGetLastError());
                PMX_HIP(hipMemcpyAsync(r_hash.data() + base, d_oh.p, sizeof(uint64_t) * (size_t)tot, hipMemcpyDeviceToHost, ctx->stream));
                PMX_HIP(hipMemcpyAsync(r_rev.data() + base, d_or.p, (size_t)tot, hipMemcpyDeviceToHost, ctx->stream));
                PMX_HIP(hipStreamSynchronize(ctx->stream));
            }
            for (int64_t i = 0; i < nc; ++i) r_off[(size_t)(c0 + i) + 1] = (int64_t)base + o_off[(size_t)i + 1];
        }
    }
    // reads with the same seedmer list (hash and orientation, in order) are one read with a multiplicity; a read without
    // seedmers scores 0 everywhere and carries no weight in the EM (src/mgsr.cpp:8170-8173): dropped here
    struct View { const uint64_t* h; const uint8_t* v; int64_t n; };
    auto view = [&](int64_t r) { return View{r_hash.data() + r_off[(size_t)r], r_rev.data() + r_off[(size_t)r], r_off[(size_t)r + 1] - r_off[(size_t)r]}; };
    std::vector<int64_t> order;
    for (int64_t r = 0; r < n_reads; ++r)
        if (r_off[(size_t)r + 1] > r_off[(size_t)r]) order.push_back(r);
    // (lexicographic on the hash list, then on the orientation list: the order the vector comparisons gave)
    auto cmp3 = [&](int64_t a, int64_t b) {
        const View x = view(a), y = view(b);
        const int64_t nmin = std::min(x.n, y.n);
        for (int64_t i = 0; i < nmin; ++i)
            if (x.h[i] != y.h[i]) return x.h[i] < y.h[i] ? -1 : 1;
        if (x.n != y.n) return x.n < y.n ? -1 : 1;
        const int c = memcmp(x.v, y.v, (size_t)x.n);
        return c < 0 ? -1 : c > 0 ? 1 : 0;
    };
    auto less = [&](int64_t a, int64_t b) { return cmp3(a, b) < 0; };
    std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { const int c = cmp3(a, b); return c < 0 || (c == 0 && a < b); });
    m->n_raw_reads = n_raw;
    m->n_dust_dropped = n_dusty;
    m->h_read_off.assign(1, 0);
    m->h_seed_hash.clear(); m->h_seed_rev.clear(); m->h_mult.clear();
    for (size_t i = 0; i < order.size(); ++i) {
        const View o = view(order[i]);
        if (i > 0 && cmp3(order[i - 1], order[i]) == 0) { ++m->h_mult.back(); continue; }
        m->h_seed_hash.insert(m->h_seed_hash.end(), o.h, o.h + o.n);
        m->h_seed_rev.insert(m->h_seed_rev.end(), o.v, o.v + o.n);
        m->h_read_off.push_back((int64_t)m->h_seed_hash.size());
        m->h_mult.push_back(1);
    }
    m->n_reads = (int64_t)m->h_mult.size();
    m->n_seedmers = (int64_t)m->h_seed_hash.size();
    m->h_uniq = m->h_seed_hash;
    std::sort(m->h_uniq.begin(), m->h_uniq.end());
    m->h_uniq.erase(std::unique(m->h_uniq.begin(), m->h_uniq.end()), m->h_uniq.end());
    std::vector<uint32_t> uid((size_t)m->n_seedmers);
    for (int64_t i = 0; i < m->n_seedmers; ++i)
        uid[(size_t)i] = (uint32_t)(std::lower_bound(m->h_uniq.begin(), m->h_uniq.end(), m->h_seed_hash[(size_t)i]) - m->h_uniq.begin());
    m->d_read_off.ensure((size_t)m->n_reads + 1);
    m->d_seed_uid.ensure((size_t)std::max<int64_t>(m->n_seedmers, 1));
    m->d_seed_rev.ensure((size_t)std::max<int64_t>(m->n_seedmers, 1));
    m->d_uniq.ensure(std::max<size_t>(m->h_uniq.size(), 1));
    PMX_HIP(hipMemcpy(m->d_read_off.p, m->h_read_off.data(), sizeof(int64_t) * ((size_t)m->n_reads + 1), hipMemcpyHostToDevice));
    if (m->n_seedmers > 0) {
        PMX_HIP(hipMemcpy(m->d_seed_uid.p, uid.data(), sizeof(uint32_t) * (size_t)m->n_seedmers, hipMemcpyHostToDevice));
        PMX_HIP(hipMemcpy(m->d_seed_rev.p, m->h_seed_rev.data(), (size_t)m->n_seedmers, hipMemcpyHostToDevice));
        PMX_HIP(hipMemcpy(m->d_uniq.p, m->h_uniq.data(), sizeof(uint64_t) * m->h_uniq.size(), hipMemcpyHostToDevice));
    }
    // overlap coefficients through the place stage: seed the reads on the device, score the tree with every read seed kept
    m->oc.assign((size_t)m->n_nodes, 0.0);
    if (n_reads > 0) {
        pmx_readset* rs = nullptr;
        int rc = pmx_readset_upload(ctx, concat, offsets, n_reads, &rs);
        if (rc != PMX_OK) return rc;
        pmx_place_params pp;
        memset(&pp, 0, sizeof(pp));
        pp.min_read_support = 1;
        pmx_place_result res;
        rc = pmx_readset_pack(ctx, rs);
        if (rc == PMX_OK) rc = pmx_place_reset(ctx, m->placer);
        if (rc == PMX_OK) rc = pmx_place_add_reads(ctx, m->placer, rs, &pp);
        if (rc == PMX_OK) rc = pmx_place_score(ctx, m->placer, &pp, n_reads, &res);
        pmx_readset_free(ctx, rs);
        if (rc != PMX_OK) return rc;
        std::vector<int64_t> counts2((size_t)m->n_nodes * 2);
        rc = pmx_place_node_outputs(ctx, m->placer, nullptr, nullptr, counts2.data());
        if (rc != PMX_OK) return rc;
        for (int64_t v = 0; v < m->n_nodes; ++v)
            m->oc[(size_t)v] = counts2[2 * (size_t)v + 1] > 0 ? (double)counts2[2 * (size_t)v] / (double)counts2[2 * (size_t)v + 1] : 0.0;
    }
    m->cand.clear();
    m->groups.clear();
    return PMX_OK;
    PMX_CATCH
}

// Step 2 (selection) + 3: the nodes of the `top_oc` best distinct overlap coefficients become the candidates
// (cand_override / n_override > 0: exactly these nodes instead -- tests); every (read, candidate) score.
int pmx_meta_score(pmx_ctx* ctx, pmx_meta* m, int64_t top_oc, const uint32_t* cand_override, int64_t n_override) {
    if (!ctx || !m || (n_override > 0 && !cand_override)) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    m->cand.clear();
    if (n_override > 0) {
        m->cand.assign(cand_override, cand_override + n_override);
        for (uint32_t v : m->cand)
            if ((int64_t)v >= m->n_nodes) return fail(PMX_ERR_ARG, "candidate node out of range");
    } else {
        std::vector<uint32_t> by_oc((size_t)m->n_nodes);
        std::iota(by_oc.begin(), by_oc.end(), 0u);
        std::stable_sort(by_oc.begin(), by_oc.end(), [&](uint32_t a, uint32_t b) { return m->oc[a] > m->oc[b]; });
        int64_t ranks = 0;
        double cur = -1.0;
        for (uint32_t v : by_oc) {
            if (m->oc[v] != cur) {
                cur = m->oc[v];
                if (++ranks > top_oc) break;
            }
            m->cand.push_back(v);
        }
    }
    std::sort(m->cand.begin(), m->cand.end());
    m->cand.erase(std::unique(m->cand.begin(), m->cand.end()), m->cand.end());
    const int n_cand = (int)m->cand.size(), words = (n_cand + 63) / 64;
    m->groups.clear();
    m->h_max_score.assign((size_t)m->n_reads, 0);
    if (n_cand == 0 || m->n_reads == 0) return PMX_OK;
    m->d_cand.ensure((size_t)n_cand);
    PMX_HIP(hipMemcpyAsync(m->d_cand.p, m->cand.data(), sizeof(uint32_t) * (size_t)n_cand, hipMemcpyHostToDevice, ctx->stream));
    const size_t mask_words = m->h_uniq.size() * (size_t)words;
    m->mask_fwd.ensure(std::max<size_t>(mask_words, 1));
    m->mask_rev.ensure(std::max<size_t>(mask_words, 1));
    PMX_HIP(hipMemsetAsync(m->mask_fwd.p, 0, sizeof(unsigned long long) * std::max<size_t>(mask_words, 1), ctx->stream));
    PMX_HIP(hipMemsetAsync(m->mask_rev.p, 0, sizeof(unsigned long long) * std::max<size_t>(mask_words, 1), ctx->stream));
    if (m->n_changes > 0)
        hipLaunchKernelGGL(k_meta_mask_events, dim3(grid_for(m->n_changes, 256, ctx->n_cu * 8)), dim3(256), 0, ctx->stream, m->ch_key.p, m->ch_pc.p,
                           m->ch_cc.p, m->ch_node.p, m->n_changes, m->subtree_end.p, m->d_uniq.p, (int64_t)m->h_uniq.size(), m->d_cand.p, n_cand, words,
                           m->mask_fwd.p, m->mask_rev.p);
    m->score.ensure((size_t)m->n_reads * (size_t)n_cand);
    int64_t longest = 0;
    for (int64_t r = 0; r < m->n_reads; ++r) longest = std::max(longest, m->h_read_off[(size_t)r + 1] - m->h_read_off[(size_t)r]);
    if (longest >= 65535) return fail(PMX_ERR_UNSUPPORTED, "a read with 65,535 seedmers or more (16-bit scores)");
    const int64_t threads = m->n_reads * (int64_t)words;
    const dim3 grid(grid_for(threads, 256, ctx->n_cu * 16)), block(256);
    if (longest < 128)
        hipLaunchKernelGGL(k_meta_scores<7>, grid, block, 0, ctx->stream, m->d_read_off.p, m->d_seed_uid.p, m->d_seed_rev.p, m->n_reads, m->mask_fwd.p,
                           m->mask_rev.p, words, n_cand, m->score.p);
    else
        hipLaunchKernelGGL(k_meta_scores<16>, grid, block, 0, ctx->stream, m->d_read_off.p, m->d_seed_uid.p, m->d_seed_rev.p, m->n_reads, m->mask_fwd.p,
                           m->mask_rev.p, words, n_cand, m->score.p);
    PMX_HIP(hipGetLastError());
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
    PMX_CATCH
}

int64_t pmx_meta_num_reads(const pmx_meta* m) { return m ? m->n_reads : 0; }
int64_t pmx_meta_num_candidates(const pmx_meta* m) { return m ? (int64_t)m->cand.size() : 0; }
int pmx_meta_candidates(const pmx_meta* m, uint32_t* out, int64_t cap) {
    if (!m || cap < (int64_t)m->cand.size()) return PMX_ERR_ARG;
    std::copy(m->cand.begin(), m->cand.end(), out);
    return PMX_OK;
}
int pmx_meta_overlap_coefficients(const pmx_meta* m, double* out, int64_t cap) {
    if (!m || cap < (int64_t)m->oc.size()) return PMX_ERR_ARG;
    std::copy(m->oc.begin(), m->oc.end(), out);
    return PMX_OK;
}
// the merged reads: seedmers per read (n) and multiplicities; either pointer may be NULL
int pmx_meta_read_info(const pmx_meta* m, int64_t* n_seedmers, int64_t* multiplicity, int64_t cap) {
    if (!m || cap < m->n_reads) return PMX_ERR_ARG;
    for (int64_t r = 0; r < m->n_reads; ++r) {
        if (n_seedmers) n_seedmers[r] = m->h_read_off[(size_t)r + 1] - m->h_read_off[(size_t)r];
        if (multiplicity) multiplicity[r] = m->h_mult[(size_t)r];
    }
    return PMX_OK;
}
// the merged reads' seedmers (hash, orientation) with n_reads + 1 offsets -- what the checker recomputes scores from
int pmx_meta_read_seedmers(const pmx_meta* m, int64_t* offsets, uint64_t* hash, uint8_t* rev, int64_t cap_seedmers) {
    if (!m || cap_seedmers < m->n_seedmers || !offsets) return PMX_ERR_ARG;
    std::copy(m->h_read_off.begin(), m->h_read_off.end(), offsets);
    if (hash) std::copy(m->h_seed_hash.begin(), m->h_seed_hash.end(), hash);
    if (rev) std::copy(m->h_seed_rev.begin(), m->h_seed_rev.end(), rev);
    return PMX_OK;
}
int pmx_meta_scores(pmx_ctx* ctx, pmx_meta* m, uint16_t* out, int64_t cap) {
    if (!ctx || !m || !out) return PMX_ERR_ARG;
    const int64_t n = m->n_reads * (int64_t)m->cand.size();
    if (cap < n) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    if (n > 0) PMX_HIP(hipMemcpy(out, m->score.p, sizeof(uint16_t) * (size_t)n, hipMemcpyDeviceToHost));
    return PMX_OK;
    PMX_CATCH
}

// Step 4: merge candidates with equal score columns, SQUAREM EM, drop nodes below prop_threshold, again (<= em_max_rounds).
int pmx_meta_em(pmx_ctx* ctx, pmx_meta* m, const pmx_meta_params* mp) {
    if (!ctx || !m || !mp) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int n_cand = (int)m->cand.size();
    const int64_t n_reads = m->n_reads;
    m->groups.clear();
    m->em_rounds = m->em_iterations = 0;
    m->llh = 0.0;
    if (n_cand == 0 || n_reads == 0) return PMX_OK;
    // ---- columns: one per distinct score column (digest; members = the other candidates of that column)
    DevBuf<uint64_t> d_dig;
    d_dig.alloc(2 * (size_t)n_cand);
    hipLaunchKernelGGL(k_meta_column_digest, dim3((n_cand + 63) / 64), dim3(64), 0, st, m->score.p, n_reads, n_cand, d_dig.p);
    std::vector<uint64_t> dig(2 * (size_t)n_cand);
    PMX_HIP(hipMemcpyAsync(dig.data(), d_dig.p, sizeof(uint64_t) * dig.size(), hipMemcpyDeviceToHost, st));
    PMX_HIP(hipStreamSynchronize(st));
    std::map<std::pair<uint64_t, uint64_t>, int> first_of;
    std::vector<int> col_cand;                       // column -> candidate position of its representative
    std::vector<std::vector<uint32_t>> col_members;
    for (int c = 0; c < n_cand; ++c) {
        const auto key = std::make_pair(dig[2 * (size_t)c], dig[2 * (size_t)c + 1]);
        auto it = first_of.find(key);
        if (it == first_of.end()) { first_of.emplace(key, (int)col_cand.size()); col_cand.push_back(c); col_members.emplace_back(); }
        else col_members[(size_t)it->second].push_back(m->cand[(size_t)c]);
    }
    // ---- rows: the reads that score somewhere (the others carry no weight, src/mgsr.cpp:8170-8173)
    std::vector<uint16_t> h_score((size_t)n_reads * (size_t)n_cand);
    PMX_HIP(hipMemcpy(h_score.data(), m->score.p, sizeof(uint16_t) * h_score.size(), hipMemcpyDeviceToHost));
    std::vector<int64_t> rows;
    for (int64_t r = 0; r < n_reads; ++r) {
        int mx = 0;
        for (int c = 0; c < n_cand; ++c) mx = std::max<int>(mx, h_score[(size_t)r * (size_t)n_cand + (size_t)c]);
        m->h_max_score[(size_t)r] = mx;
        const int64_t n_seed = m->h_read_off[(size_t)r + 1] - m->h_read_off[(size_t)r];
        // --discard (src/main.cpp:1229-1240): the threshold is TRUNCATED to an integer there,
        // `maxScore < static_cast<int>(seedmers * discard)`, so a read with int(n * d) <= max < n * d stays in the EM
        if (mx == 0 || mx < (int)((double)n_seed * mp->discard)) continue;
        rows.push_back(r);
    }
    const int64_t n_rows = (int64_t)rows.size();
    if (n_rows == 0) return PMX_OK;
    // ---- P(read | node) = err^(n - s) * (1 - err)^s for the distinct n of the reads: tables computed with the host libm
    std::map<int64_t, uint32_t> off_of_n;
    std::vector<double> tab;
    std::vector<uint32_t> tab_off((size_t)n_rows);
    std::vector<double> weight((size_t)n_rows);
    double total_weight = 0.0;
    for (int64_t j = 0; j < n_rows; ++j) {
        const int64_t r = rows[(size_t)j], n_seed = m->h_read_off[(size_t)r + 1] - m->h_read_off[(size_t)r];
        auto it = off_of_n.find(n_seed);
        if (it == off_of_n.end()) {
            it = off_of_n.emplace(n_seed, (uint32_t)tab.size()).first;
            for (int64_t s = 0; s <= n_seed; ++s) tab.push_back(std::pow(mp->error_rate, (double)(n_seed - s)) * std::pow(1.0 - mp->error_rate, (double)s));
        }
        tab_off[(size_t)j] = it->second;
        weight[(size_t)j] = (double)m->h_mult[(size_t)r];
        total_weight += weight[(size_t)j];
    }
    const double inv_total = 1.0 / total_weight;
    DevBuf<int64_t> d_rows;
    DevBuf<uint32_t> d_tab_off;
    DevBuf<double> d_tab, d_weight, d_denom, d_llh, d_props, d_out, d_part, d_bsum;
    DevBuf<int> d_cols;
    d_rows.alloc((size_t)n_rows); d_tab_off.alloc((size_t)n_rows); d_tab.alloc(tab.size()); d_weight.alloc((size_t)n_rows);
    d_denom.alloc((size_t)n_rows); d_llh.alloc((size_t)n_rows);
    PMX_HIP(hipMemcpy(d_rows.p, rows.data(), sizeof(int64_t) * (size_t)n_rows, hipMemcpyHostToDevice));
    PMX_HIP(hipMemcpy(d_tab_off.p, tab_off.data(), sizeof(uint32_t) * (size_t)n_rows, hipMemcpyHostToDevice));
    PMX_HIP(hipMemcpy(d_tab.p, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice));
    PMX_HIP(hipMemcpy(d_weight.p, weight.data(), sizeof(double) * (size_t)n_rows, hipMemcpyHostToDevice));
    const int64_t chunk = 256;   // reads per partial column sum (2,048: 2,200 waves for 3,000 columns x 90k reads, each a serial walk: 373 us per pass)
    const int n_chunks = (int)((n_rows + chunk - 1) / chunk);
    const int64_t n_bsum = (n_rows + 1023) / 1024;
    d_bsum.alloc((size_t)n_bsum);
    std::vector<double> h_bsum((size_t)n_bsum);

    std::vector<int> cols = col_cand;       // current columns (candidate positions)
    std::vector<std::vector<uint32_t>> members = col_members;
    std::vector<double> props;
    for (int round = 0; round < std::max(1, mp->em_max_rounds); ++round) {
        const int n_cols = (int)cols.size();
        d_cols.ensure((size_t)n_cols); d_props.ensure((size_t)n_cols); d_out.ensure((size_t)n_cols); d_part.ensure((size_t)n_chunks * (size_t)n_cols);
        PMX_HIP(hipMemcpy(d_cols.p, cols.data(), sizeof(int) * (size_t)n_cols, hipMemcpyHostToDevice));
        // The whole SQUAREM loop stays on the device: the proportion vectors never leave it, the small vector arithmetic runs in
        // single-block kernels with the host loop's own order of operations, and the host only looks at the convergence flag
        // every 16 iterations (launches queued past convergence return at once).  Per iteration: 6 passes over the score
        // matrix (4 x k_meta_denoms, 2 x k_meta_colsum) and 13 small launches; before, 4 host round trips.
        DevBuf<double> d_p0, d_p1, d_p2, d_sq;
        DevBuf<EmCtl> d_ctl;
        d_p0.alloc((size_t)n_cols); d_p1.alloc((size_t)n_cols); d_p2.alloc((size_t)n_cols); d_sq.alloc((size_t)n_cols);
        d_ctl.alloc(1);
        EmCtl h_ctl;
        memset(&h_ctl, 0, sizeof(h_ctl));
        PMX_HIP(hipMemcpyAsync(d_ctl.p, &h_ctl, sizeof(h_ctl), hipMemcpyHostToDevice, st));
        props.assign((size_t)n_cols, 1.0 / (double)n_cols);
        PMX_HIP(hipMemcpyAsync(d_props.p, props.data(), sizeof(double) * (size_t)n_cols, hipMemcpyHostToDevice, st));
        const int* d_done = &d_ctl.p->done;
        // the in-order sums of the small kernels read a copy of the vector in LDS (2 x n_cols doubles at most); beyond 160 KB a
        // scratch vector in global memory stands in
        DevBuf<double> d_em_work;
        double* em_work = nullptr;
        const size_t em_lds = 2 * sizeof(double) * (size_t)n_cols;
        if (em_lds > (size_t)160 * 1024) { d_em_work.alloc(2 * (size_t)n_cols); em_work = d_em_work.p; }
        else if (em_lds > (size_t)64 * 1024) {
            PMX_HIP(hipFuncSetAttribute((const void*)k_em_normalize, hipFuncAttributeMaxDynamicSharedMemorySize, (int)em_lds));
            PMX_HIP(hipFuncSetAttribute((const void*)k_em_extrapolate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)em_lds));
        }
        auto denoms = [&](const double* pr) {
            hipLaunchKernelGGL(k_meta_denoms, dim3(grid_for(n_rows * 64, 256, ctx->n_cu * 8)), dim3(256), 0, st, m->score.p, n_cand, d_cols.p, n_cols, pr,
                               d_rows.p, n_rows, d_tab_off.p, d_tab.p, d_weight.p, d_denom.p, d_llh.p, d_done);
        };
        auto em_step = [&](const double* from, double* to) {   // updateProps (src/mgsr.cpp:4341-4372) + normalizeProps
            denoms(from);
            hipLaunchKernelGGL(k_meta_colsum, dim3((n_cols + 63) / 64, n_chunks), dim3(64), 0, st, m->score.p, n_cand, d_cols.p, n_cols, from, d_rows.p,
                               n_rows, chunk, d_tab_off.p, d_tab.p, d_weight.p, d_denom.p, d_part.p, d_done);
            hipLaunchKernelGGL(k_meta_fold, dim3((n_cols + 63) / 64), dim3(64), 0, st, d_part.p, n_chunks, n_cols, inv_total, d_out.p, d_done);
            hipLaunchKernelGGL(k_em_normalize, dim3(1), dim3(256), em_work ? 0 : sizeof(double) * (size_t)n_cols, st, d_out.p, to, n_cols, d_ctl.p, em_work);
        };
        auto log_likelihood = [&](const double* pr, int which) {                          // getExp (:4385-4388)
            denoms(pr);
            hipLaunchKernelGGL(k_meta_sum_blocks, dim3((unsigned)((n_bsum + 3) / 4)), dim3(256), 0, st, d_llh.p, n_rows, d_bsum.p, d_done);
            hipLaunchKernelGGL(k_em_store_llh, dim3(1), dim3(64), 0, st, d_bsum.p, n_bsum, d_ctl.p, which);
        };
        const int look_every = 16;
        for (int iter = 0; iter < mp->em_max_iterations;) {                               // runSquareEM (:4394-4443)
            const int batch = std::min(look_every, mp->em_max_iterations - iter);
            for (int b = 0; b < batch; ++b) {
                hipLaunchKernelGGL(k_em_copy, dim3(1), dim3(256), 0, st, d_props.p, d_p0.p, n_cols, d_ctl.p);
                em_step(d_p0.p, d_p1.p);
                em_step(d_p1.p, d_p2.p);
                hipLaunchKernelGGL(k_em_extrapolate, dim3(1), dim3(256), em_work ? 0 : 2 * sizeof(double) * (size_t)n_cols, st, d_p0.p, d_p1.p, d_p2.p, d_sq.p, n_cols, d_ctl.p, em_work);
                log_likelihood(d_p2.p, 0);
                log_likelihood(d_sq.p, 1);
                hipLaunchKernelGGL(k_em_choose, dim3(1), dim3(256), 0, st, d_p0.p, d_p2.p, d_sq.p, d_props.p, n_cols, d_ctl.p, mp->em_convergence,
                                   mp->em_delta_threshold);
            }
            PMX_HIP(hipGetLastError());
            PMX_HIP(hipMemcpyAsync(&h_ctl, d_ctl.p, sizeof(h_ctl), hipMemcpyDeviceToHost, st));
            PMX_HIP(hipStreamSynchronize(st));
            iter += batch;
            if (h_ctl.done) break;
        }
        PMX_HIP(hipMemcpyAsync(props.data(), d_props.p, sizeof(double) * (size_t)n_cols, hipMemcpyDeviceToHost, st));
        PMX_HIP(hipStreamSynchronize(st));
        m->em_iterations += h_ctl.iterations;
        const double llh = h_ctl.llh;
        m->llh = llh;
        ++m->em_rounds;
        // removeLowPropNodes (:4445-4490), called after EVERY round including the last allowed one (src/main.cpp:1263-1271):
        // when it removes anything the surviving nodes' proportions are reset to uniform, and if that was the last round
        // the uniform vector is what the abundance file reports -- mirrored, not "fixed"
        std::vector<int> keep;
        for (int i = 0; i < n_cols; ++i)
            if (props[(size_t)i] >= mp->prop_threshold) keep.push_back(i);
        if ((int)keep.size() == n_cols) break;
        std::vector<int> cols2;
        std::vector<std::vector<uint32_t>> members2;
        for (int i : keep) { cols2.push_back(cols[(size_t)i]); members2.push_back(members[(size_t)i]); }
        cols.swap(cols2);
        members.swap(members2);
        props.assign(cols.size(), cols.empty() ? 0.0 : 1.0 / (double)cols.size());
        if (cols.empty()) break;
    }
    for (size_t i = 0; i < cols.size(); ++i) {
        pmx_meta_group g;
        g.node = m->cand[(size_t)cols[i]];
        g.members = members[i];
        g.prop = props[i];
        m->groups.push_back(std::move(g));
    }
    std::stable_sort(m->groups.begin(), m->groups.end(), [](const pmx_meta_group& a, const pmx_meta_group& b) { return a.prop > b.prop; });
    return PMX_OK;
    PMX_CATCH
}

int pmx_meta_set_dust(pmx_meta* m, double threshold) {
    if (!m || !(threshold <= 100.0)) return PMX_ERR_ARG;      // src/main.cpp:1353-1356: --dust must be <= 100
    m->dust_threshold = threshold;
    return PMX_OK;
}

int64_t pmx_meta_num_haplotypes(const pmx_meta* m) { return m ? (int64_t)m->groups.size() : 0; }
int pmx_meta_haplotype(const pmx_meta* m, int64_t i, uint32_t* node, double* prop, int64_t* n_members, uint32_t* members, int64_t cap) {
    if (!m || i < 0 || i >= (int64_t)m->groups.size()) return PMX_ERR_ARG;
    const pmx_meta_group& g = m->groups[(size_t)i];
    if (node) *node = g.node;
    if (prop) *prop = g.prop;
    if (n_members) *n_members = (int64_t)g.members.size();
    if (members && cap >= (int64_t)g.members.size()) std::copy(g.members.begin(), g.members.end(), members);
    return PMX_OK;
}
int pmx_meta_em_info(const pmx_meta* m, int32_t* rounds, int32_t* iterations, double* log_likelihood) {
    if (!m) return PMX_ERR_ARG;
    if (rounds) *rounds = m->em_rounds;
    if (iterations) *iterations = m->em_iterations;
    if (log_likelihood) *log_likelihood = m->llh;
    return PMX_OK;
}

}  // extern "C"
