// PLACE stage kernels for gfx950 (wave64): read packing, syncmer / k-min-mer seeding fused with
// the seed-histogram insert, histogram finalisation, and per-node delta scoring down the tree.
//
// Everything here is integer hashing / FP64 accumulation on HBM- and latency-bound access
// patterns: no MFMA.  Reference behaviour: src/seeding.cpp:47-229 (syncmers),
// src/placement.cpp:1611-1686 (k-min-mers + histogram), :931-984 (read magnitudes),
// :242-345 (computeChildMetrics), src/placement.hpp:120-149 (score getters).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device/pmx_math.h"
#include "place_kernels.h"

namespace pmx {

// ------------------------------------------------------------------------------------- pack
// One thread per output word (32 bases): 2-bit code + ambiguity bit.  Under an ambiguity bit the
// code field is 3 for 'U'/'u' (minimap2's nt4 table maps U to T, the seeding hash does not) else 0.
__global__ void k_read_word_counts(const int64_t* __restrict__ off, int64_t n_reads, int64_t total_bytes, int64_t* __restrict__ nwords,
                                   unsigned long long* stats) {
    unsigned long long mx = 0, bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n_reads; i += (int64_t)gridDim.x * blockDim.x) {
        if (i == n_reads) { nwords[i] = 0; continue; }   // (the scan's last output = the total)
        const int64_t a = off[i], b = off[i + 1];
        const int64_t len = b - a;
        if (len < 0 || a < 0 || b > total_bytes) { bad = 1; nwords[i] = 0; continue; }
        nwords[i] = (len + 31) / 32;
        mx = (unsigned long long)len > mx ? (unsigned long long)len : mx;
    }
    // one atomic per wave
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long m2 = __shfl_xor(mx, o), b2 = __shfl_xor(bad, o);
        mx = m2 > mx ? m2 : mx;
        bad |= b2;
    }
    if ((threadIdx.x & 63) == 0) {
        if (mx) atomicMax(&stats[0], mx);
        if (bad) atomicOr(&stats[1], 1ULL);
    }
}

// word j of read r into the read's 64-byte record (reads of up to 160 bases: five words): 2-bit words at bytes 0..39,
// ambiguity words at 40..59, the length at 60; the thread of word 0 also clears the slots the read does not use (records are
// compared dword for dword by k_collapse_reads)
__device__ __forceinline__ void pack_record(uint8_t* recs, int64_t r, int j, int len, uint64_t v, uint32_t a) {
    if (len > 160 || j >= 5) return;
    uint8_t* rec = recs + (size_t)r * 64;
    reinterpret_cast<uint64_t*>(rec)[j] = v;
    reinterpret_cast<uint32_t*>(rec + 40)[j] = a;
    if (j == 0) {
        reinterpret_cast<uint32_t*>(rec + 60)[0] = (uint32_t)len;
        for (int q = (len + 31) >> 5; q < 5; ++q) { reinterpret_cast<uint64_t*>(rec)[q] = 0; reinterpret_cast<uint32_t*>(rec + 40)[q] = 0; }
    }
}

// reads [r0, r1) only (r1 < 0: every read): the words woff[r0] .. woff[r1] - 1
__global__ void k_pack_reads(const uint8_t* __restrict__ ascii, const int64_t* __restrict__ off,
                             const int64_t* __restrict__ woff, int64_t n_reads, int64_t n_words,
                             uint64_t* __restrict__ words, uint32_t* __restrict__ amb, int64_t r0, int64_t r1, uint8_t* __restrict__ recs) {
    const int64_t w_lo = r1 < 0 ? 0 : woff[r0], w_hi = r1 < 0 ? n_words : woff[r1];
    for (int64_t w = w_lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < w_hi; w += (int64_t)gridDim.x * blockDim.x) {
        // read r with woff[r] <= w < woff[r+1]: start from the proportional guess (exact for reads of one length: two
        // loads instead of log2(n) dependent ones), gallop to a bracket, then bisect
        int64_t lo, hi;
        {
            int64_t g = (int64_t)((double)w * (double)n_reads / (double)n_words);
            g = g < 0 ? 0 : (g > n_reads - 1 ? n_reads - 1 : g);
            if (woff[g] <= w) {
                lo = g;
                int64_t step = 1;
                hi = g + 1;
                while (hi < n_reads && woff[hi] <= w) { lo = hi; step <<= 1; hi = hi + step < n_reads ? hi + step : n_reads; }
            } else {
                hi = g;
                int64_t step = 1;
                lo = g - 1;
                while (lo > 0 && woff[lo] > w) { hi = lo; step <<= 1; lo = lo - step > 0 ? lo - step : 0; }
            }
            while (hi - lo > 1) {
                const int64_t mid = (lo + hi) >> 1;
                if (woff[mid] <= w) lo = mid; else hi = mid;
            }
        }
        const int64_t r = lo;
        const int64_t base0 = (w - woff[r]) * 32;
        const int64_t len = off[r + 1] - off[r];
        const uint8_t* p = ascii + off[r] + base0;
        int nb = (int)(len - base0 < 32 ? len - base0 : 32);
        uint64_t v = 0;
        uint32_t a = 0;
        // the (up to) 32 bytes as aligned 32-bit words: nine loads instead of thirty-two byte loads (the word that holds a
        // needed byte lies inside the buffer's last page whatever the buffer's end; no word without a needed byte is touched)
        const uintptr_t addr = (uintptr_t)p;
        const uint32_t* q = reinterpret_cast<const uint32_t*>(addr & ~(uintptr_t)3);
        const int sh = (int)(addr & 3u);
        const int n_w = nb > 0 ? (sh + nb + 3) >> 2 : 0;
        uint32_t wd[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wd[t] = t < n_w ? q[t] : 0u;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            // A C G T (either case) -> 0 1 2 3 without a branch: with c = ch & 0xDF, ((c >> 1) ^ (c >> 2)) & 3 maps
            // 0x41 0x43 0x47 0x54 to 0 1 2 3 (and U = 0x55 to 3); anything else is ambiguous with code 0
            // (static register indices: byte j sits in the 64-bit pair wd[j/4 + 1] : wd[j/4], shifted by the thread's offset)
            const uint64_t pair = (uint64_t)wd[(j >> 2) + 1] << 32 | (uint64_t)wd[j >> 2];
            const uint32_t ch = (uint32_t)(pair >> (8 * ((j & 3) + sh))) & 0xffu;
            const uint32_t c = ch & 0xDFu;
            const uint32_t tr = ((c >> 1) ^ (c >> 2)) & 3u;
            const bool acgt = c == 0x41u || c == 0x43u || c == 0x47u || c == 0x54u;
            const bool is_u = c == 0x55u;
            const bool in = j < nb;
            const uint32_t code = (in && (acgt || is_u)) ? tr : 0u, am = (in && !acgt) ? 1u : 0u;
            v |= (uint64_t)code << (2 * j);
            a |= am << j;
        }
        words[w] = v;
        amb[w] = a;
        if (recs) pack_record(recs, r, (int)(w - woff[r]), (int)len, v, a);
    }
}

// The same for a read set whose reads all have `len` bases (the host knows: total = n x max): read and word follow from
// the word index by a division, and the word's 32 bytes arrive as three aligned 16-byte loads -- no offset look-ups, a
// third of the load instructions: the kernel is a stream and sits near the device-to-device copy rate.
__global__ void k_pack_reads_fixed(const uint8_t* __restrict__ ascii, int64_t off0, int len, int64_t w_lo, int64_t w_hi,
                                   uint64_t* __restrict__ words, uint32_t* __restrict__ amb, uint8_t* __restrict__ recs) {
    const int nw = (len + 31) >> 5;
    for (int64_t w = w_lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < w_hi; w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = w / nw;
        const int base0 = (int)(w - r * nw) * 32;
        const int nb = len - base0 < 32 ? len - base0 : 32;
        const uintptr_t addr = (uintptr_t)(ascii + off0 + r * (int64_t)len + base0);
        const uint4* q = reinterpret_cast<const uint4*>(addr & ~(uintptr_t)15);
        const int sh = (int)(addr & 15u);
        // (a 16-byte block that holds a needed byte lies inside the buffer's allocation granule; none without one is touched)
        const int n_q = (sh + nb + 15) >> 4;
        uint32_t wd[12];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (t < n_q) v = q[t];
            wd[4 * t] = v.x; wd[4 * t + 1] = v.y; wd[4 * t + 2] = v.z; wd[4 * t + 3] = v.w;
        }
        // shift the 48 bytes down by sh bytes: byte j of the word's bases = byte sh + j of wd
        uint32_t bs[8];
        const int sw = sh >> 2, sb = (sh & 3) * 8;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {   // (sw is 0..3: selects instead of a dynamic register index)
                lo = sw == c ? wd[t + c] : lo;
                hi = sw == c ? wd[t + c + 1] : hi;
            }
            bs[t] = sb ? (lo >> sb) | (hi << (32 - sb)) : lo;
        }
        uint64_t v = 0;
        uint32_t a = 0;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const uint32_t ch = (bs[j >> 2] >> (8 * (j & 3))) & 0xffu;
            const uint32_t c = ch & 0xDFu;
            const uint32_t tr = ((c >> 1) ^ (c >> 2)) & 3u;
            const bool acgt = c == 0x41u || c == 0x43u || c == 0x47u || c == 0x54u;
            const bool is_u = c == 0x55u;
            const bool in = j < nb;
            const uint32_t code = (in && (acgt || is_u)) ? tr : 0u, am = (in && !acgt) ? 1u : 0u;
            v |= (uint64_t)code << (2 * j);
            a |= am << j;
        }
        words[w] = v;
        amb[w] = a;
        if (recs) pack_record(recs, r, (int)(w - r * nw), len, v, a);
    }
}

// --------------------------------------------------------------------------------- read dedup (--dedup)
// The reference sorts the read strings and counts every distinct sequence once (src/placement.cpp:1550-1620).
// Here: two independent 64-bit hashes of the raw ASCII of each read, a stable radix sort on the 128-bit key,
// and an exact byte comparison with the predecessor in sorted order; the first read of every run is kept.
__global__ void k_read_hashes(const uint8_t* __restrict__ ascii, const int64_t* __restrict__ off, int64_t n_reads, uint64_t* h1,
                              uint64_t* h2, uint32_t* idx) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = off[r], e = off[r + 1];
        uint64_t a = 0xcbf29ce484222325ULL, c = 0x9e3779b97f4a7c15ULL ^ (uint64_t)(e - b);
        for (int64_t i = b; i < e; ++i) {
            const uint64_t ch = ascii[i];
            a = (a ^ ch) * 0x100000001b3ULL;                   // FNV-1a
            c = mix64(c + ch + 0x632be59bd9b4e019ULL);         // chained avalanche
        }
        h1[r] = a;
        h2[r] = c;
        idx[r] = (uint32_t)r;
    }
}

__global__ void k_gather_u64(const uint64_t* __restrict__ src, const uint32_t* __restrict__ idx, int64_t n, uint64_t* dst) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[idx[i]];
}

// sorted by (h1, h2): keep[read] = 1 for the first read of every run of byte-identical reads
__global__ void k_mark_first_of_run(const uint8_t* __restrict__ ascii, const int64_t* __restrict__ off, const uint64_t* __restrict__ h1s,
                                    const uint64_t* __restrict__ h2, const uint32_t* __restrict__ perm, int64_t n, uint8_t* keep) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t r = perm[j];
        bool first = true;
        if (j > 0) {
            const uint32_t q = perm[j - 1];
            const int64_t lr = off[r + 1] - off[r], lq = off[q + 1] - off[q];
            if (h1s[j] == h1s[j - 1] && h2[r] == h2[q] && lr == lq) {
                bool same = true;
                for (int64_t i = 0; i < lr && same; ++i) same = ascii[off[r] + i] == ascii[off[q] + i];
                first = !same;
            }
        }
        keep[r] = first ? 1 : 0;
    }
}

// --dedup over ranks: the (h1, h2) of the reads a rank keeps, appended in any order
__global__ void k_kept_read_hashes(const uint64_t* __restrict__ h1, const uint64_t* __restrict__ h2, const uint8_t* __restrict__ keep, int64_t n,
                                   uint64_t* out_h1, uint64_t* out_h2, unsigned long long* n_out) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        if (!keep[r]) continue;
        const unsigned long long at = atomicAdd(n_out, 1ULL);
        out_h1[at] = h1[r];
        out_h2[at] = h2[r];
    }
}
// ... and a kept read whose hash pair occurs among `seen` (sorted by h1; the kept reads of the lower ranks) is dropped
__global__ void k_drop_seen_reads(const uint64_t* __restrict__ h1, const uint64_t* __restrict__ h2, int64_t n, const uint64_t* __restrict__ seen_h1,
                                  const uint64_t* __restrict__ seen_h2, int64_t n_seen, uint8_t* keep) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        if (!keep[r]) continue;
        const uint64_t a = h1[r], b = h2[r];
        int64_t lo = 0, hi = n_seen;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (seen_h1[mid] < a) lo = mid + 1;
            else hi = mid;
        }
        for (; lo < n_seen && seen_h1[lo] == a; ++lo)
            if (seen_h2[lo] == b) { keep[r] = 0; break; }
    }
}

// --------------------------------------------------------------------------------- seeding
__device__ __forceinline__ void table_insert(uint64_t* keys, unsigned long long* vals, uint64_t mask, uint64_t key,
                                             unsigned long long mult, unsigned long long* counters) {
    // A slot's key goes EMPTY -> key once and never changes again, so the common case (the seed is already in
    // the table: the ~10^4 true seeds of a sample are hit millions of times) needs no compare-and-swap: a plain
    // device-scope load finds the slot and the count is bumped with a fire-and-forget atomic add.  A returning
    // CAS on those hot lines cost microseconds each and the 37 inserts of a read are serial.
    // (probe sequences are capped: at the load factors the host keeps, a run of 4096 occupied slots means the table was
    //  sized too small; the insert is counted as an overflow and the host starts over with a larger table)
    uint64_t slot = mix64(key) & mask;
    const uint64_t max_probes = mask < 4096 ? mask : 4096;
    for (uint64_t probes = 0; probes <= max_probes; ++probes) {
        // a long probe sequence: if some insert has already failed the host will redo the call with a larger table, so
        // stop walking a full one.  (Checked only here: every wave polling one word at every drain costs +190 us per
        // launch -- reads of a single address are served one after the other.)
        if ((probes & 63) == 63 && __hip_atomic_load(&counters[PMX_CTR_OVERFLOW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ULL) return;
        const unsigned long long cur = __hip_atomic_load((unsigned long long*)&keys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == key) {
            atomicAdd(&vals[slot], mult);
            return;
        }
        if (cur == PMX_EMPTY_KEY) {
            const unsigned long long prev = atomicCAS((unsigned long long*)&keys[slot], (unsigned long long)PMX_EMPTY_KEY, (unsigned long long)key);
            if (prev == PMX_EMPTY_KEY) atomicAdd(&counters[PMX_CTR_SHARD0 + (slot & (PMX_CTR_NSHARD - 1))], 1ULL);   // sharded: one hot word would serialise
            if (prev == PMX_EMPTY_KEY || prev == key) {
                atomicAdd(&vals[slot], mult);
                return;
            }
        }
        slot = (slot + 1) & mask;
    }
    atomicAdd(&counters[PMX_CTR_OVERFLOW], 1ULL);
}

// Insert a wave's queued seeds: every active lane takes up to four queue entries at a time and issues their first
// probes together (independent loads in flight), so a drain costs about one table round trip per four entries per
// lane instead of one per entry.  A first probe that neither hits nor finds an empty slot continues in table_insert.
// ABSORBED: the caller (k_seed_histogram_ks) overwrites the entries its block cache has taken with PMX_EMPTY_KEY and they
// are skipped here; the generic kernel has no cache and hands every entry on.  (A seed whose hash IS the all-ones sentinel
// -- 2^-64 per seed -- cannot be a key of the table in either kernel: the table's empty mark is that value.)
template <bool ABSORBED>
__device__ __forceinline__ void drain_seed_queue(const uint64_t* queue, int n_q, int rank, int n_act, uint64_t* keys, unsigned long long* vals,
                                                 uint64_t mask, unsigned long long* counters, const uint8_t* qlane = nullptr,
                                                 const uint32_t* wmult = nullptr) {
    // qlane / wmult (k_seed_histogram_ks over distinct-read tiles): entry q was pushed by lane qlane[q], whose read stands for
    // wmult[lane] identical reads -- its seeds count that many times
    for (int q0 = rank; q0 < n_q; q0 += 4 * n_act) {
        uint64_t h[4], slot[4];
        unsigned long long cur[4], m[4];
        bool ok[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int q = q0 + b * n_act;
            ok[b] = q < n_q;
            h[b] = ok[b] ? queue[q] : 0;
            m[b] = (ok[b] && qlane) ? (unsigned long long)wmult[qlane[q]] : 1ULL;
            if (ABSORBED) ok[b] = ok[b] && h[b] != PMX_EMPTY_KEY;   // (entries the block cache has absorbed)
            slot[b] = mix64(h[b]) & mask;
            cur[b] = 0;
            if (ok[b]) cur[b] = __hip_atomic_load((unsigned long long*)&keys[slot[b]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (!ok[b]) continue;
            if (cur[b] == h[b]) atomicAdd(&vals[slot[b]], m[b]);
            else table_insert(keys, vals, mask, h[b], m[b], counters);   // empty slot or collision: the full probe sequence
        }
    }
}

// One thread per read; rolling k-mer / s-mer hashes in registers, the (k-s+1)-deep s-mer ring and the
// l-deep syncmer ring of every thread in LDS, laid out [slot][thread] (conflict-free ds_read_b64).
__global__ void __launch_bounds__(PMX_SEED_BLOCK)
k_seed_histogram(const uint64_t* __restrict__ words, const uint32_t* __restrict__ amb, const int64_t* __restrict__ woff,
                 const int64_t* __restrict__ off, int64_t r_begin, int64_t n_reads, SeedParams sp, uint64_t* keys,
                 unsigned long long* vals, uint64_t mask, unsigned long long* counters, const uint8_t* __restrict__ keep,
                 const uint8_t* __restrict__ qual, int min_q, uint64_t* __restrict__ list_hash, uint8_t* __restrict__ list_rev,
                 uint32_t* __restrict__ list_n) {
    // list_hash != NULL (--meta, src/mgsr.cpp:1774-2237): no histogram -- every read's seedmers go, in read order and with
    // their orientation (the k-min-mer read right to left is the smaller one: R < F), to the read's own stretch of the list
    // arrays: entry e of read r at woff[r] * 32 + e (a read has fewer seedmers than bases), their number to list_n[r]
    // (cleared by the host: a read too short for a k-mer writes nothing).  l >= 2, no quality filter.
    extern __shared__ uint64_t lds[];
    const int w = sp.k - sp.s + 1;
    const int l = sp.l < 1 ? 1 : sp.l;
    uint64_t* ringF = lds;                                   // [w][B]
    uint64_t* ringR = lds + (size_t)w * PMX_SEED_BLOCK;      // [w][B]
    uint64_t* ringS = lds + (size_t)2 * w * PMX_SEED_BLOCK;  // [l][B]
    const int tid = threadIdx.x;
    // Seeds are not inserted where they are found: a thread finds one at ~1 base in 6, so with an insert in the base
    // loop a wave would wait for a table probe (an L2 round trip) at nearly every base with a handful of lanes
    // active.  Each wave queues its seeds in LDS and, when the queue is nearly full (and once at the end), all of
    // its lanes insert one queued seed each.  Counts are sums, so the insertion order does not matter.
    uint64_t* queue = lds + (size_t)(2 * w + l) * PMX_SEED_BLOCK + (size_t)(tid >> 6) * PMX_SEED_QCAP;
    uint32_t* qcnt = reinterpret_cast<uint32_t*>(lds + (size_t)(2 * w + l) * PMX_SEED_BLOCK + (size_t)(PMX_SEED_BLOCK / 64) * PMX_SEED_QCAP) + (tid >> 6);
    if ((tid & 63) == 0) *qcnt = 0;
    auto enqueue = [&](uint64_t h) { queue[atomicAdd(qcnt, 1u)] = h; };
    auto drain = [&]() {   // called with the wave's in-loop lanes converged; strides over the ACTIVE lanes
        const unsigned long long act = __ballot(1);
        const int rank = (int)__popcll(act & ((1ULL << (tid & 63)) - 1ULL)), n_act = (int)__popcll(act);
        const int n_q = (int)__hip_atomic_load(qcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        drain_seed_queue<false>(queue, n_q, rank, n_act, keys, vals, mask, counters);
        if (rank == 0) *qcnt = 0;
    };
    // base hashes A, C, G, T (src/seeding.hpp:100-112) picked with selects, not a table in memory
    auto HB = [](uint32_t c) -> uint64_t {
        const uint64_t lo = (c & 1u) ? 0x3193c18562a02b4cULL : 0x3c8bfbb395c60474ULL;
        const uint64_t hi = (c & 1u) ? 0x295549f54be24456ULL : 0x20323ed082572324ULL;
        return (c & 2u) ? hi : lo;
    };
    unsigned long long n_seeds = 0;
    // ring slots advance by one per base: kept as wrapping counters (a 64-bit '%' per base costs hundreds of instructions)
    const int first0 = sp.t % w, last0 = (sp.k - sp.s - sp.t) % w;

    for (int64_t r = r_begin + (int64_t)blockIdx.x * PMX_SEED_BLOCK + tid; r < n_reads; r += (int64_t)gridDim.x * PMX_SEED_BLOCK) {
        const int64_t len = off[r + 1] - off[r];
        if (len < sp.k) continue;
        if (keep && !keep[r]) continue;   // --dedup: a later copy of an identical read
        const uint64_t* rw = words + woff[r];
        const uint32_t* ra = amb + woff[r];
        const int64_t valid_start = sp.trim_start, valid_end = len - sp.trim_end - sp.k;
        uint64_t fS = 0, rS = 0, fK = 0, rK = 0;
        uint64_t hist2 = 0;   // last 32 base codes, newest in bits 1:0
        uint32_t hista = 0;   // last 32 ambiguity bits, newest in bit 0
        int64_t last_amb = -1;
        uint64_t cw = 0;
        uint32_t ca = 0;
        uint64_t F = 0, R = 0;  // k-min-mer rolling hashes
        int64_t n_sync = 0;
        int slot_w = 0;                        // (i - s + 1) mod w
        int s_first = first0, s_last = last0;  // (ks + t) mod w, (ks + k - s - t) mod w
        int slot_l = 0;                        // (n_sync - 1) mod l
        uint32_t n_list = 0;                   // list mode: seedmers of this read so far
        // --min-seed-quality (src/placement.cpp:1386-1527): rolling sum of (qual - 33) over the k-mer, and the number
        // of consecutive syncmers (of the read's FULL syncmer list) that passed the trim + quality test
        const uint8_t* rq = qual ? qual + off[r] : nullptr;
        int qsum = 0, pass_run = 0;
        const int ilen = (int)len;
        for (int i = 0; i < ilen; ++i) {
            if (__hip_atomic_load(qcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > (uint32_t)(PMX_SEED_QCAP - 64)) drain();   // uniform: every in-loop lane reads the same word
            if ((i & 31) == 0) { cw = rw[i >> 5]; ca = ra[i >> 5]; }
            const uint32_t code = (uint32_t)(cw & 3u);
            const uint32_t am = ca & 1u;
            cw >>= 2; ca >>= 1;
            const uint64_t hb = am ? 0 : HB(code), hc = am ? 0 : HB(3 - code);
            if (am) last_amb = i;
            // outgoing bases (distance s and k behind)
            if (i < sp.s) { fS ^= rotl64(hb, (unsigned)(sp.s - 1 - i)); rS ^= rotl64(hc, (unsigned)i); }
            else {
                const uint32_t oc = (uint32_t)(hist2 >> (2 * (sp.s - 1))) & 3u, oa = (hista >> (sp.s - 1)) & 1u;
                const uint64_t ob = oa ? 0 : HB(oc), ocm = oa ? 0 : HB(3 - oc);
                fS = rotl64(fS, 1) ^ rotl64(ob, (unsigned)sp.s) ^ hb;
                rS = rotr64(rS, 1) ^ rotr64(ocm, 1) ^ rotl64(hc, (unsigned)(sp.s - 1));
            }
            if (i < sp.k) { fK ^= rotl64(hb, (unsigned)(sp.k - 1 - i)); rK ^= rotl64(hc, (unsigned)i); }
            else {
                const uint32_t oc = (uint32_t)(hist2 >> (2 * (sp.k - 1))) & 3u, oa = (hista >> (sp.k - 1)) & 1u;
                const uint64_t ob = oa ? 0 : HB(oc), ocm = oa ? 0 : HB(3 - oc);
                fK = rotl64(fK, 1) ^ rotl64(ob, (unsigned)sp.k) ^ hb;
                rK = rotr64(rK, 1) ^ rotr64(ocm, 1) ^ rotl64(hc, (unsigned)(sp.k - 1));
            }
            hist2 = (hist2 << 2) | code;
            hista = (hista << 1) | am;
            if (i >= sp.s - 1) {
                ringF[(size_t)slot_w * PMX_SEED_BLOCK + tid] = fS;
                ringR[(size_t)slot_w * PMX_SEED_BLOCK + tid] = rS;
                slot_w = slot_w + 1 == w ? 0 : slot_w + 1;
            }
            if (rq) {
                qsum += (int)rq[i] - 33;
                if (i >= sp.k) qsum -= (int)rq[i - sp.k] - 33;
            }
            if (i < sp.k - 1) continue;
            const int64_t ks = i - sp.k + 1;
            const int cur_first = s_first, cur_last = s_last;   // this k-mer's slots; advance for the next one
            s_first = s_first + 1 == w ? 0 : s_first + 1;
            s_last = s_last + 1 == w ? 0 : s_last + 1;
            if (last_amb >= ks || fK == rK) continue;
            // window minima (all w slots: a data-dependent rescan diverges across the wave and was measured slower)
            uint64_t fmin = UINT64_MAX, rmin = UINT64_MAX;
#pragma unroll 4
            for (int j = 0; j < w; ++j) {
                const uint64_t a = ringF[(size_t)j * PMX_SEED_BLOCK + tid], b = ringR[(size_t)j * PMX_SEED_BLOCK + tid];
                fmin = a < fmin ? a : fmin;
                rmin = b < rmin ? b : rmin;
            }
            bool fs, rs;
            if (sp.open) {
                fs = ringF[(size_t)cur_first * PMX_SEED_BLOCK + tid] == fmin;
                rs = ringR[(size_t)cur_last * PMX_SEED_BLOCK + tid] == rmin;
            } else {
                fs = ringF[(size_t)cur_first * PMX_SEED_BLOCK + tid] == fmin || ringF[(size_t)cur_last * PMX_SEED_BLOCK + tid] == fmin;
                rs = ringR[(size_t)cur_last * PMX_SEED_BLOCK + tid] == rmin || ringR[(size_t)cur_first * PMX_SEED_BLOCK + tid] == rmin;
            }
            if (!(fs || rs)) continue;
            const uint64_t h = fK < rK ? fK : rK;
            if (rq) {   // quality-filtered branch: windows over all syncmers, valid only if every member passes
                const bool pass = ks >= valid_start && ks <= valid_end && qsum >= min_q * sp.k;   // avg >= min_q, exactly
                ++n_sync;
                if (sp.l <= 1) {
                    if (pass) { enqueue(h); ++n_seeds; }
                    continue;
                }
                ringS[(size_t)slot_l * PMX_SEED_BLOCK + tid] = h;
                slot_l = slot_l + 1 == l ? 0 : slot_l + 1;   // now the slot of the OLDEST of the last l syncmers
                pass_run = pass ? pass_run + 1 : 0;
                if (n_sync >= l && pass_run >= l) {
                    uint64_t Fq = 0, Rq = 0;
                    int sl = slot_l;
                    for (int q2 = 0; q2 < l; ++q2) {   // oldest -> newest
                        const uint64_t hq = ringS[(size_t)sl * PMX_SEED_BLOCK + tid];
                        Fq = rotl64(Fq, (unsigned)sp.k) ^ hq;
                        Rq ^= rotl64(hq, (unsigned)(sp.k * q2));
                        sl = sl + 1 == l ? 0 : sl + 1;
                    }
                    if (Fq != Rq) { enqueue(Fq < Rq ? Fq : Rq); ++n_seeds; }
                }
                continue;
            }
            if (ks < valid_start || ks > valid_end) continue;   // primer trim (src/placement.cpp:1629-1648)
            ++n_sync;
            if (sp.l <= 1) {
                enqueue(h);
                ++n_seeds;
                continue;
            }
            bool have = false;
            if (n_sync <= l) {
                F = rotl64(F, (unsigned)sp.k) ^ h;
                R ^= rotl64(h, (unsigned)(sp.k * (int)(n_sync - 1)));
                have = n_sync == l;
            } else {
                const uint64_t prev = ringS[(size_t)slot_l * PMX_SEED_BLOCK + tid];
                F = rotl64(F, (unsigned)sp.k) ^ rotl64(prev, (unsigned)(sp.k * l)) ^ h;
                R = rotr64(R, (unsigned)sp.k) ^ rotr64(prev, (unsigned)sp.k) ^ rotl64(h, (unsigned)(sp.k * (l - 1)));
                have = true;
            }
            ringS[(size_t)slot_l * PMX_SEED_BLOCK + tid] = h;
            slot_l = slot_l + 1 == l ? 0 : slot_l + 1;
            if (have && F != R) {
                if (list_hash) {
                    const int64_t at = woff[r] * 32 + (int64_t)n_list;
                    list_hash[at] = F < R ? F : R;
                    list_rev[at] = R < F ? 1 : 0;
                    ++n_list;
                } else enqueue(F < R ? F : R);
                ++n_seeds;
            }
        }
        if (list_n) list_n[r] = n_list;
    }
    drain();   // every lane of the wave is here
    // one atomic per wave: a per-thread atomic on this single word was the whole cost of the kernel
    // (same-address atomics retire at a few hundred per microsecond)
    for (int o = 32; o > 0; o >>= 1) n_seeds += __shfl_xor(n_seeds, o);
    if ((tid & 63) == 0 && n_seeds) atomicAdd(&counters[PMX_CTR_SEEDS], n_seeds);
}

// ---------------------------------------------------------------------------------------------------------
// k_seed_histogram for compile-time (K, S), t == 0 and no quality filter (the default parameters k=19 s=8): the
// generic kernel is bound by instruction issue (~770 instructions per base, half of them branch bookkeeping).
// Here
//  * every rotation amount is a constant, and "rotate the hash of the outgoing base" is a select among four
//    pre-rotated constants;
//  * the minimum of the last W = K-S+1 s-mer hashes comes from the block prefix/suffix scheme (van Herk /
//    Gil-Werman): s-mers are grouped in blocks of W; a running prefix minimum of the current block lives in a
//    register, and when a block completes one backward pass turns its W ring slots into suffix minima IN PLACE.
//    The window ending at s-mer p is the tail [p-W+1, block end] of the previous block plus the head of the current
//    one, so min = min(suffix[(p+1) mod W], prefix): ~3 ring operations per base instead of 2W reads.  With t == 0
//    the closed-syncmer test needs "oldest s-mer == min" and "newest s-mer == min": the newest is in a register,
//    and the oldest equals the minimum iff it was its own suffix minimum (one bit per slot, kept in a register mask
//    by the backward pass) and that suffix minimum is <= the prefix;
//  * the syncmer / k-min-mer bookkeeping is predicated instead of branched wherever that is cheap.
// Results are identical to k_seed_histogram (same hashes, same syncmers, same counts).
__device__ __forceinline__ constexpr uint64_t c_rotl64(uint64_t x, unsigned r) { return (r & 63u) ? (x << (r & 63u)) | (x >> (64u - (r & 63u))) : x; }
// the same rotation at run time, by a compile-time amount, as two v_alignbit_b32 on the halves (the shift-and-or form compiles
// to a 64-bit shift -- a slow-rate instruction on this chip -- plus a 32-bit shift and an or: four rotations per base in the
// seeding loop)
template <unsigned R>
__device__ __forceinline__ uint64_t d_rotl64(uint64_t x) {
    constexpr unsigned r = R & 63u;
    if (r == 0) return x;
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    if (r >= 32) { const uint32_t t_ = lo; lo = hi; hi = t_; }   // rotate by 32: swap the halves
    constexpr unsigned q = r & 31u;
    if (q == 0) return (uint64_t)hi << 32 | lo;
    const uint32_t nlo = __builtin_amdgcn_alignbit(lo, hi, 32u - q);   // (lo << q) | (hi >> (32 - q))
    const uint32_t nhi = __builtin_amdgcn_alignbit(hi, lo, 32u - q);
    return (uint64_t)nhi << 32 | nlo;
}
__device__ __forceinline__ constexpr uint64_t c_hb(uint32_t c) {
    return c == 0 ? 0x3c8bfbb395c60474ULL : c == 1 ? 0x3193c18562a02b4cULL : c == 2 ? 0x20323ed082572324ULL : 0x295549f54be24456ULL;
}
// hash of base c rotated left by R, or of its complement (3 - c); 0 when the position is ambiguous
template <unsigned R, bool COMP>
__device__ __forceinline__ uint64_t hb_rot(uint32_t c, uint32_t am) {
    constexpr uint64_t v0 = c_rotl64(c_hb(COMP ? 3 : 0), R), v1 = c_rotl64(c_hb(COMP ? 2 : 1), R), v2 = c_rotl64(c_hb(COMP ? 1 : 2), R),
                       v3 = c_rotl64(c_hb(COMP ? 0 : 3), R);
    const uint64_t lo = (c & 1u) ? v1 : v0, hi = (c & 1u) ? v3 : v2;
    const uint64_t v = (c & 2u) ? hi : lo;
    return am ? 0ULL : v;
}

// k_seed_histogram_ks: a wave's queued seeds go to the block's (seed, count) cache, or -- first sightings, slots taken by
// another seed -- to the table.  Called by all 64 lanes.
__device__ __forceinline__ void seed_queue_to_cache_and_table(uint64_t* queue, int n_q, int lane, unsigned long long* ckey, uint32_t* ccnt, uint16_t* ctag,
                                                           uint64_t* keys, unsigned long long* vals, uint64_t mask, unsigned long long* counters,
                                                           const uint8_t* qlane, const uint32_t* wmult) {
    for (int q = lane; q < n_q; q += 64) {
        const uint64_t h = queue[q];
        const uint32_t m = wmult[qlane[q]];
        if (h == PMX_EMPTY_KEY) { table_insert(keys, vals, mask, h, (unsigned long long)m, counters); continue; }   // (the sentinel value itself)
        const uint64_t hm = mix64(h);
        const uint32_t cs = (uint32_t)hm & (PMX_SEED_CACHE - 1);
        const uint16_t tag = (uint16_t)((hm >> 40) | 1u);
        unsigned long long cur = ckey[cs];
        if (cur == PMX_EMPTY_KEY) {
            if (ctag[cs] != tag) { ctag[cs] = tag; continue; }   // first sighting: to the table (stays queued)
            cur = atomicCAS(&ckey[cs], (unsigned long long)PMX_EMPTY_KEY, (unsigned long long)h);
            if (cur == PMX_EMPTY_KEY) cur = h;
        }
        if (cur == h) {
            atomicAdd(&ccnt[cs], m);
            queue[q] = PMX_EMPTY_KEY;
        }
    }
    drain_seed_queue<true>(queue, n_q, lane, 64, keys, vals, mask, counters, qlane, wmult);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Read collapse (src/placement.cpp:1550-1593: the reference sorts the reads and seeds every DISTINCT sequence once with its
// multiplicity -- counts are additive, so the histogram is the same).  Here, ahead of the seeding kernel: a block takes 1,024
// consecutive reads of the locality order (reads that start within a few bases of each other: at depth, most of them are
// copies of one another), keeps every read's packed form (five 2-bit words + five ambiguity words + its length = 64 bytes) in
// LDS, and a 2,048-slot LDS hash table elects one owner per distinct record -- a candidate that meets an owner with the same
// hash is compared with it DWORD FOR DWORD (exact: no read is merged on a hash alone) and adds one to the owner's count.  The
// owners leave the block as TILES of 64 distinct reads laid out [tile][word][lane], which the seeding kernel then reads with
// coalesced loads (the gather from the reads' own, scattered, places happens once, here, a whole read at a time: the seeding
// kernel used to re-fetch a read's lines for every 32 bases).  Copies that fall into different blocks are simply seeded twice.
// Reads shorter than min_len (no k-mer) and reads a --dedup mask drops are not emitted.  Reads of up to 160 bases.
__global__ void __launch_bounds__(PMX_DEDUP_BLOCK)
k_collapse_reads(const uint64_t* __restrict__ words, const uint32_t* __restrict__ amb, const int64_t* __restrict__ woff, const int64_t* __restrict__ off,
                 int64_t r_begin, int64_t r_end, const uint8_t* __restrict__ keep, const uint32_t* __restrict__ perm, int min_len, int fixed_len,
                 const uint8_t* __restrict__ recs, uint64_t* __restrict__ t_words, uint32_t* __restrict__ t_amb, uint32_t* __restrict__ t_len, uint32_t* __restrict__ t_mult,
                 unsigned long long* n_out) {
    // dynamic LDS, PMX_DEDUP_LDS_BYTES (80 KB for 1,024 reads: two blocks per CU)
    extern __shared__ uint32_t lds32[];
    uint32_t (*rec)[PMX_DEDUP_BLOCK] = reinterpret_cast<uint32_t (*)[PMX_DEDUP_BLOCK]>(lds32);   // [dword][thread]: words 0..4 (lo, hi), ambiguity words 0..4, length
    uint32_t* rhash = lds32 + 16 * PMX_DEDUP_BLOCK;
    uint32_t* mult = rhash + PMX_DEDUP_BLOCK;
    uint32_t* tab = mult + PMX_DEDUP_BLOCK;              // 2 * PMX_DEDUP_BLOCK slots
    __shared__ uint32_t wave_tot[PMX_DEDUP_BLOCK / 64];
    __shared__ unsigned long long base_sh;
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t j = r_begin + (int64_t)blockIdx.x * PMX_DEDUP_BLOCK + tid;
    uint64_t w[5] = {0, 0, 0, 0, 0};
    uint32_t a[5] = {0, 0, 0, 0, 0};
    uint32_t len = 0;
    if (j < r_end) {
        const int64_t r = perm ? (int64_t)perm[j] : j;
        // fixed_len > 0: every read of the set has that many bases (the host knows: total = n x max), so the read's place in
        // the packed arrays follows from its index -- two scattered loads per read less
        if (recs) {
            // the read's record: one aligned 64-byte line, its length included (ragged sets: the host clears the records
            // before packing, so a read without a single word -- length 0 -- reads as length 0)
            const uint4* rp = reinterpret_cast<const uint4*>(recs + (size_t)r * 64);
            const uint4 q0 = rp[0], q1 = rp[1], q2 = rp[2], q3 = rp[3];
            const int64_t l64 = fixed_len > 0 ? (int64_t)fixed_len : (int64_t)q3.w;
            if (l64 >= min_len && l64 <= 160 && !(keep && !keep[r])) {
                w[0] = (uint64_t)q0.y << 32 | q0.x; w[1] = (uint64_t)q0.w << 32 | q0.z;
                w[2] = (uint64_t)q1.y << 32 | q1.x; w[3] = (uint64_t)q1.w << 32 | q1.z;
                w[4] = (uint64_t)q2.y << 32 | q2.x;
                a[0] = q2.z; a[1] = q2.w; a[2] = q3.x; a[3] = q3.y; a[4] = q3.z;
                len = (uint32_t)l64;
            }
        } else {
        const int64_t l64 = fixed_len > 0 ? (int64_t)fixed_len : off[r + 1] - off[r];
        if (l64 >= min_len && l64 <= 160 && !(keep && !keep[r])) {
            len = (uint32_t)l64;
            const int nw = (int)((l64 + 31) >> 5);
            const int64_t w0 = fixed_len > 0 ? r * (int64_t)nw : woff[r];
            const uint64_t* rw = words + w0;
            const uint32_t* ra = amb + w0;
#pragma unroll
            for (int q = 0; q < 5; ++q)
                if (q < nw) { w[q] = rw[q]; a[q] = ra[q]; }
        }
        }
    }
    uint32_t d[16];
#pragma unroll
    for (int q = 0; q < 5; ++q) { d[2 * q] = (uint32_t)w[q]; d[2 * q + 1] = (uint32_t)(w[q] >> 32); d[10 + q] = a[q]; }
    d[15] = len;
    uint32_t h = 0x9747b28cu;
#pragma unroll
    for (int q = 0; q < 16; ++q) {   // murmur3-style mixing of the sixteen dwords
        uint32_t k1 = d[q] * 0xcc9e2d51u;
        k1 = (k1 << 15) | (k1 >> 17);
        h ^= k1 * 0x1b873593u;
        h = ((h << 13) | (h >> 19)) * 5u + 0xe6546b64u;
        rec[q][tid] = d[q];
    }
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    rhash[tid] = h;
    mult[tid] = 1u;
    tab[tid] = 0xffffffffu;
    tab[tid + PMX_DEDUP_BLOCK] = 0xffffffffu;
    __syncthreads();
    bool owner = false;
    if (len) {
        uint32_t slot = h & (2 * PMX_DEDUP_BLOCK - 1);
        for (;;) {
            const uint32_t cur = atomicCAS(&tab[slot], 0xffffffffu, (uint32_t)tid);
            if (cur == 0xffffffffu) { owner = true; break; }
            if (rhash[cur] == h) {
                bool same = true;
#pragma unroll
                for (int q = 0; q < 16; ++q) same = same && rec[q][cur] == d[q];
                if (same) { atomicAdd(&mult[cur], 1u); break; }
            }
            slot = (slot + 1) & (2 * PMX_DEDUP_BLOCK - 1);   // (at most PMX_DEDUP_BLOCK owners in twice as many slots: ends)
        }
    }
    // owners -> consecutive places of the tile arrays (one cursor bump per block)
    const unsigned long long ob = __ballot(owner);
    const uint32_t in_wave = (uint32_t)__popcll(ob & ((1ULL << lane) - 1ULL));
    if (lane == 0) wave_tot[tid >> 6] = (uint32_t)__popcll(ob);
    __syncthreads();   // (also: every duplicate has added itself to its owner's count)
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int q = 0; q < PMX_DEDUP_BLOCK / 64; ++q) {
        const uint32_t t = wave_tot[q];
        before += q < (tid >> 6) ? t : 0u;
        total += t;
    }
    if (tid == 0) base_sh = total ? atomicAdd(n_out, (unsigned long long)total) : 0ULL;
    __syncthreads();
    if (owner) {
        const uint64_t p = base_sh + before + in_wave;
        const uint64_t tile = p >> 6, ln = p & 63;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            t_words[(tile * 5 + q) * 64 + ln] = w[q];
            t_amb[(tile * 5 + q) * 64 + ln] = a[q];
        }
        t_len[p] = len;
        // (--dedup: the mask has already dropped the later copies of every read STRING, src/placement.cpp:1619-1620; kept reads
        //  that pack to the same record -- they differ in letter case only -- are different strings and each counts)
        t_mult[p] = mult[tid];
    }
}

template <int K, int S, int L>
__global__ void __launch_bounds__(PMX_SEED_BLOCK) __attribute__((amdgpu_waves_per_eu(3, 3)))   // 170 VGPRs: the LDS footprint admits 12 waves per CU
k_seed_histogram_ks(const uint64_t* __restrict__ words, const uint32_t* __restrict__ amb, const int64_t* __restrict__ woff,
                    const int64_t* __restrict__ off, int64_t r_begin, int64_t n_reads, SeedParams sp, uint64_t* keys, unsigned long long* vals,
                    uint64_t mask, unsigned long long* counters, const uint8_t* __restrict__ keep, const uint32_t* __restrict__ perm,
                    const uint32_t* __restrict__ t_len, const uint32_t* __restrict__ t_mult, const unsigned long long* __restrict__ t_count) {
    // Two input forms.  t_len == nullptr: the reads of a read set in place (words / amb / woff / off, visited through perm).
    // t_len != nullptr: TILES of distinct reads from k_collapse_reads -- words / amb are the tile arrays [tile][word][lane]
    // (a wave = one tile: every load is one coalesced line per word), t_len / t_mult the reads' lengths and multiplicities,
    // *t_count how many there are (r_begin = 0; n_reads = the launch's capacity: blocks past the count leave at once).
    constexpr int W = K - S + 1;
    static_assert(K <= 32 && S >= 2 && W >= 2 && W <= 32, "window of 2..32 s-mers");
    static_assert(PMX_SEED_QCAP_KS >= 64 * W + 64, "the queue must take a block of W bases from every lane");
    // Per-base LDS round trips were this kernel's time (round-2 counters: 10 LDS instructions per base, each a full stall
    // with two waves per SIMD; half of every wave's cycles waiting): the s-mer ring now lives in REGISTERS.  The base loop
    // runs over s-mer indices in blocks of W with the block fully unrolled, so every ring slot is a compile-time register;
    // the last l syncmer hashes are a four-deep shift register (l <= 4, host-checked); the queue cursor is a wave-uniform
    // register advanced by a ballot (all 64 lanes run every iteration: reads that are too short, filtered or past their end
    // are predicated off, never branched around).  LDS holds only the wave's seed queue and the block's (seed, count) cache.
    extern __shared__ uint64_t lds[];
    constexpr int l = L;   // sp.l, host-checked
    static_assert(L >= 1 && L <= 4, "the syncmer shift register holds four hashes");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    uint64_t* queue = lds + (size_t)(tid >> 6) * PMX_SEED_QCAP_KS;
    // Block cache: every table increment is a memory-side atomic.  The host hands the reads over sorted by their first 16
    // bases (perm[]), so the 128 reads of a block are a few stacks of reads that start at the same place and carry the same
    // seeds: a small direct-mapped (seed, count) cache in LDS absorbs the repeats and the block adds each cached seed to the
    // table ONCE, with its count, when it ends.  A slot taken by another seed sends the seed to the table directly.  Sums
    // commute: same histogram.  A seed is admitted on its SECOND sighting (the first leaves a 16-bit tag in the slot's side
    // word and goes to the table itself), so the sequencing-error singletons -- two thirds of a block's distinct seeds --
    // never hold a slot.
    unsigned long long* ckey = reinterpret_cast<unsigned long long*>(lds + (size_t)(PMX_SEED_BLOCK / 64) * PMX_SEED_QCAP_KS);
    uint32_t* ccnt = reinterpret_cast<uint32_t*>(ckey + PMX_SEED_CACHE);
    uint16_t* ctag = reinterpret_cast<uint16_t*>(ccnt + PMX_SEED_CACHE);
    for (int i = tid; i < PMX_SEED_CACHE; i += PMX_SEED_BLOCK) { ckey[i] = PMX_EMPTY_KEY; ccnt[i] = 0; ctag[i] = 0; }
    // base-hash tables: seven rotations x (A, C, G, T, ambiguous = 0), read with one ds_read_b64 each instead of six selects
    uint64_t* hb_tab = reinterpret_cast<uint64_t*>(ctag + PMX_SEED_CACHE);
    // who pushed a queue entry, and how many identical reads that lane's read stands for (1 without tiles)
    uint32_t* wmult = reinterpret_cast<uint32_t*>(hb_tab + 35) + (size_t)(tid >> 6) * 64;
    uint8_t* qlane = reinterpret_cast<uint8_t*>(reinterpret_cast<uint32_t*>(hb_tab + 35) + (PMX_SEED_BLOCK / 64) * 64) + (size_t)(tid >> 6) * PMX_SEED_QCAP_KS;
    const bool tiles = t_len != nullptr;
    const int wstride = tiles ? 64 : 1;
    if (tiles) n_reads = (int64_t)*t_count < n_reads ? (int64_t)*t_count : n_reads;
    if (tid < 35) {
        const int t = tid / 5, c = tid % 5;
        const bool comp = t == 1 || t == 2 || t == 4 || t == 6;
        const unsigned rot = t == 0 ? 0u : t == 1 ? (unsigned)(S - 1) : t == 2 ? (unsigned)(K - 1) : t == 3 ? (unsigned)S : t == 5 ? (unsigned)K : 63u;
        hb_tab[tid] = c == 4 ? 0ULL : rotl64(c_hb((uint32_t)(comp ? 3 - c : c)), rot);
    }
    __syncthreads();
    int n_q = 0;   // wave-uniform
    auto drain = [&]() {   // all 64 lanes; called once per block of W bases (inlined at every unrolled base it was most of the kernel's code)
        seed_queue_to_cache_and_table(queue, n_q, lane, ckey, ccnt, ctag, keys, vals, mask, counters, qlane, wmult);
        n_q = 0;
    };
    unsigned long long n_seeds = 0;
    constexpr unsigned rot_k = (unsigned)K & 63u, rot_kl = (unsigned)(K * l) & 63u, rot_kl1 = (unsigned)(K * (l - 1)) & 63u;

    // a block takes a CONTIGUOUS range of the (locality-sorted) reads, several batches of PMX_SEED_BLOCK one after the other:
    // neighbours in that order carry the same seeds, so the longer a block's range, the fewer entries its cache flushes
    // per read (every flushed entry and every first sighting is one memory-side atomic)
    const int64_t per_block = (((n_reads - r_begin) + gridDim.x - 1) / gridDim.x + PMX_SEED_BLOCK - 1) / PMX_SEED_BLOCK * PMX_SEED_BLOCK;
    const int64_t blk_lo = r_begin + (int64_t)blockIdx.x * per_block, blk_hi = blk_lo + per_block < n_reads ? blk_lo + per_block : n_reads;
    for (int64_t rb = blk_lo + (tid & ~63); rb < blk_hi; rb += PMX_SEED_BLOCK) {   // wave-uniform
        const int64_t rp = rb + lane;
        int ilen = 0;
        uint32_t my_mult = 1u;
        const uint64_t* rw = words;
        const uint32_t* ra = amb;
        if (n_q > 0) drain();   // (the lanes' multiplicities are about to change: nothing of the last batch may stay queued)
        if (rp < blk_hi) {
            if (tiles) {
                ilen = (int)t_len[rp];
                my_mult = t_mult[rp];
                rw = words + (size_t)(rb >> 6) * 5 * 64 + lane;   // (rb is a multiple of 64: the wave's tile)
                ra = amb + (size_t)(rb >> 6) * 5 * 64 + lane;
            } else {
                const int64_t r = perm ? (int64_t)perm[rp] : rp;
                const int64_t len = off[r + 1] - off[r];
                if (len >= K && !(keep && !keep[r])) {   // (--dedup: a later copy of an identical read is skipped)
                    ilen = (int)len;
                    rw = words + woff[r];
                    ra = amb + woff[r];
                }
            }
        }
        wmult[lane] = my_mult;
        const int valid_start = sp.trim_start, valid_end = ilen - sp.trim_end - K;
        const int n_words = (ilen + 31) >> 5;
        uint64_t fS = 0, rS = 0, fK = 0, rK = 0;
        uint32_t hist_lo = 0, hist_hi = 0;   // last 32 base codes, newest in bits 1:0 of hist_lo (two 32-bit halves: 64-bit shifts are slow-rate)
        uint32_t hista = 0xffffffffu;  // last 32 ambiguity bits, newest in bit 0; all set: no base leaves the first windows
        int last_amb = -1;
        uint32_t cw_lo = 0, cw_hi = 0;                       // the 32 bases being processed, next one in bits 1:0 of cw_lo
        uint64_t cw_next = ilen > 0 ? rw[0] : 0;             // the next 32 bases are requested while these are processed
        uint32_t ca = 0, ca_next = ilen > 0 ? ra[0] : 0;
        uint64_t F = 0, R = 0;  // k-min-mer rolling hashes
        int n_sync = 0;
        uint64_t sy0 = 0, sy1 = 0, sy2 = 0, sy3 = 0;   // the last four syncmer hashes, newest first
        uint64_t rgF[W], rgR[W];                          // open block: the s-mer hashes; closed block: its suffix minima
#pragma unroll
        for (int j = 0; j < W; ++j) { rgF[j] = 0; rgR[j] = 0; }
        uint64_t pfF = 0, pfR = 0, pf0F = 0, pf0R = 0;   // prefix minima of the current block, and the block's first s-mers
        uint32_t selfF = 0, selfR = 0;                   // previous block: bit j = s-mer j was its own suffix minimum

        // one base: the rolling ntHash values of the s-mer and the k-mer that end at it (src/seeding.hpp).  The rolling form
        // serves from base 0 on: while nothing leaves the window the outgoing term is zero (hista starts all ones), and with
        // zero start values the first complete k-mer / s-mer hashes equal the direct sums.
        // The table values of base i + 1 are requested while base i is processed (the outgoing codes are known from the
        // history words, the incoming one from the read word): no LDS round trip in the base's dependency chain.
        uint32_t n_code = 0, n_am = 0;
        uint64_t n_hb = 0, n_cS = 0, n_cK = 0, n_oS = 0, n_oSr = 0, n_oK = 0, n_oKr = 0;
        auto fetch_base = [&](int i) {
            if ((i & 31) == 0) {
                cw_lo = (uint32_t)cw_next; cw_hi = (uint32_t)(cw_next >> 32); ca = ca_next;
                if ((i >> 5) + 1 < n_words) { cw_next = rw[((i >> 5) + 1) * wstride]; ca_next = ra[((i >> 5) + 1) * wstride]; }
            }
            n_code = cw_lo & 3u;
            n_am = ca & 1u;
            cw_lo = __builtin_amdgcn_alignbit(cw_hi, cw_lo, 2); cw_hi >>= 2; ca >>= 1;
            constexpr unsigned bS = 2u * (unsigned)(S - 1), bK = 2u * (unsigned)(K - 1);   // where the base that leaves the s-mer / k-mer sits
            const uint32_t ocS = ((bS < 32u ? hist_lo : hist_hi) >> (bS & 31u)) & 3u, oaS = (hista >> (S - 1)) & 1u;
            const uint32_t ocK = ((bK < 32u ? hist_lo : hist_hi) >> (bK & 31u)) & 3u, oaK = (hista >> (K - 1)) & 1u;
            const uint64_t* tc = hb_tab + (n_am ? 4u : n_code);   // tables: 0 hb, 1 comp << S-1, 2 comp << K-1, 3 << S, 4 comp << 63, 5 << K
            const uint64_t* tS = hb_tab + (oaS ? 4u : ocS);
            const uint64_t* tK = hb_tab + (oaK ? 4u : ocK);
            n_hb = tc[0]; n_cS = tc[5]; n_cK = tc[10];
            n_oS = tS[15]; n_oSr = tS[20];
            n_oK = tK[25]; n_oKr = tK[20];
        };
        auto step_base = [&](int i) {
            last_amb = n_am ? i : last_amb;
            fS = d_rotl64<1>(fS) ^ n_oS ^ n_hb;
            rS = d_rotl64<63>(rS) ^ n_oSr ^ n_cS;
            fK = d_rotl64<1>(fK) ^ n_oK ^ n_hb;
            rK = d_rotl64<63>(rK) ^ n_oKr ^ n_cK;
            hist_hi = __builtin_amdgcn_alignbit(hist_hi, hist_lo, 30);   // (hist_hi << 2) | (hist_lo >> 30)
            hist_lo = (hist_lo << 2) | n_code;
            hista = (hista << 1) | n_am;
            fetch_base(i + 1);
        };
        fetch_base(0);
        for (int i = 0; i < S - 1 && __any(i < ilen); ++i)
            if (i < ilen) step_base(i);
        for (int p0 = 0; __any(p0 + S - 1 < ilen); p0 += W) {
            if (n_q > PMX_SEED_QCAP_KS - 64 * W) drain();   // a block adds at most 64 * W seeds
#pragma unroll
            for (int r = 0; r < W; ++r) {
                const int i = p0 + r + S - 1;
                bool have = false;
                uint64_t out = 0;
                if (i < ilen) {
                    step_base(i);
                    // ---- window minimum of the last W s-mers (block prefix / suffix minima, see the header comment)
                    uint64_t fmin, rmin;
                    bool f_old_min, r_old_min;   // the OLDEST s-mer of the window equals the minimum
                    pfF = r == 0 ? fS : (fS < pfF ? fS : pfF);
                    pfR = r == 0 ? rS : (rS < pfR ? rS : pfR);
                    pf0F = r == 0 ? fS : pf0F;
                    pf0R = r == 0 ? rS : pf0R;
                    if (r == W - 1) {   // the window is exactly the current block; then turn the block into suffix minima
                        fmin = pfF; rmin = pfR;
                        f_old_min = pf0F == pfF;
                        r_old_min = pf0R == pfR;
                        uint64_t sF = fS, sR = rS;
                        selfF = selfR = 1u << (W - 1);
                        rgF[W - 1] = sF;
                        rgR[W - 1] = sR;
#pragma unroll
                        for (int j = W - 2; j >= 0; --j) {
                            const uint64_t xF = rgF[j], xR = rgR[j];
                            selfF |= xF <= sF ? 1u << j : 0u;
                            selfR |= xR <= sR ? 1u << j : 0u;
                            sF = xF < sF ? xF : sF;
                            sR = xR < sR ? xR : sR;
                            rgF[j] = sF;
                            rgR[j] = sR;
                        }
                    } else {
                        const uint64_t sufF = rgF[r + 1 < W ? r + 1 : 0], sufR = rgR[r + 1 < W ? r + 1 : 0];
                        fmin = sufF < pfF ? sufF : pfF;
                        rmin = sufR < pfR ? sufR : pfR;
                        f_old_min = (selfF >> (r + 1) & 1u) != 0u && sufF <= pfF;
                        r_old_min = (selfR >> (r + 1) & 1u) != 0u && sufR <= pfR;
                        rgF[r] = fS;   // (the block is still open: raw values)
                        rgR[r] = rS;
                    }
                    if (p0 > 0 || r == W - 1) {   // i >= K - 1: a whole k-mer
                        const int ks = i - K + 1;
                        // closed syncmer: the minimum sits at the first or the last s-mer; open: at the first (forward strand) /
                        // the last (reverse strand).  t == 0: first = oldest, last = newest of the window.
                        const bool f_new_min = fS == fmin, r_new_min = rS == rmin;
                        const bool fs = sp.open ? f_old_min : (f_old_min || f_new_min);
                        const bool rs = sp.open ? r_new_min : (r_new_min || r_old_min);
                        const bool sync = !(last_amb >= ks || fK == rK) && (fs || rs) && ks >= valid_start && ks <= valid_end;
                        if (sync) {
                            const uint64_t h = fK < rK ? fK : rK;
                            ++n_sync;
                            out = h;
                            have = true;
                            if (l > 1) {
                                if (n_sync <= l) {
                                    F = d_rotl64<rot_k>(F) ^ h;
                                    R ^= rotl64(h, (unsigned)(K * (n_sync - 1)) & 63u);
                                    have = n_sync == l;
                                } else {
                                    const uint64_t prev = l == 2 ? sy1 : l == 3 ? sy2 : sy3;   // the syncmer that leaves the k-min-mer
                                    F = d_rotl64<rot_k>(F) ^ d_rotl64<rot_kl>(prev) ^ h;
                                    R = d_rotl64<64u - rot_k>(R) ^ d_rotl64<64u - rot_k>(prev) ^ d_rotl64<rot_kl1>(h);
                                }
                                sy3 = sy2; sy2 = sy1; sy1 = sy0; sy0 = h;
                                have = have && F != R;
                                out = F < R ? F : R;
                            }
                        }
                    }
                }
                const unsigned long long pushers = __ballot(have);
                if (have) {
                    const int at = n_q + (int)__popcll(pushers & ((1ULL << lane) - 1ULL));
                    queue[at] = out;
                    qlane[at] = (uint8_t)lane;
                    n_seeds += my_mult;
                }
                n_q += (int)__popcll(pushers);
            }
        }
    }
    drain();
    __syncthreads();
    for (int i = tid; i < PMX_SEED_CACHE; i += PMX_SEED_BLOCK) {   // the cached seeds go to the table once, with their counts
        const uint32_t c = ccnt[i];
        if (c) table_insert(keys, vals, mask, (uint64_t)ckey[i], (unsigned long long)c, counters);
    }
    for (int o = 32; o > 0; o >>= 1) n_seeds += __shfl_xor(n_seeds, o);
    if (lane == 0 && n_seeds) atomicAdd(&counters[PMX_CTR_SEEDS], n_seeds);
}
template __global__ void k_seed_histogram_ks<19, 8, 3>(const uint64_t*, const uint32_t*, const int64_t*, const int64_t*, int64_t, int64_t, SeedParams,
                                                       uint64_t*, unsigned long long*, uint64_t, unsigned long long*, const uint8_t*, const uint32_t*,
                                                       const uint32_t*, const uint32_t*, const unsigned long long*);
template __global__ void k_seed_histogram_ks<19, 8, 1>(const uint64_t*, const uint32_t*, const int64_t*, const int64_t*, int64_t, int64_t, SeedParams,
                                                       uint64_t*, unsigned long long*, uint64_t, unsigned long long*, const uint8_t*, const uint32_t*,
                                                       const uint32_t*, const uint32_t*, const unsigned long long*);

// locality key of a read for the seeding order (read_locality_key: reads that start within a few bases of each other)
__global__ void k_read_prefix_keys(const uint64_t* __restrict__ words, const int64_t* __restrict__ woff, int64_t r_begin, int64_t n_reads, uint32_t* key, uint32_t* idx) {
    for (int64_t r = r_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += (int64_t)gridDim.x * blockDim.x) {
        key[r] = woff[r + 1] > woff[r] ? read_locality_key(words[woff[r]]) : 0u;
        idx[r] = (uint32_t)r;
    }
}

// merge externally supplied (hash,count) pairs into the table (multi-GPU histogram exchange)
__global__ void k_table_merge(const uint64_t* __restrict__ hash, const int64_t* __restrict__ count, int64_t n, uint64_t* keys,
                              unsigned long long* vals, uint64_t mask, unsigned long long* counters) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        table_insert(keys, vals, mask, hash[i], (unsigned long long)count[i], counters);
}

__global__ void k_table_rehash(const uint64_t* __restrict__ okeys, const unsigned long long* __restrict__ ovals, uint64_t ocap,
                               uint64_t* keys, unsigned long long* vals, uint64_t mask, unsigned long long* counters) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ocap; i += (uint64_t)gridDim.x * blockDim.x)
        if (okeys[i] != PMX_EMPTY_KEY) table_insert(keys, vals, mask, okeys[i], ovals[i], counters);
}

__global__ void k_table_compact(const uint64_t* __restrict__ keys, const unsigned long long* __restrict__ vals, uint64_t cap,
                                uint64_t* out_hash, int64_t* out_count, unsigned long long* n_out) {
    // One wave compacts a tile of 64 x 16 slots per step: coalesced loads, a wave prefix sum of the per-lane
    // occupancy and ONE atomic per tile on the output cursor (a single-address atomic per 64 slots was the
    // whole cost of this kernel: ~88 atomics/us on one address).  Output order is arbitrary (sorted next).
    constexpr int K = 16;
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t n_tiles = (cap + 64 * K - 1) / (64 * K);
    for (uint64_t tile = wave; tile < n_tiles; tile += n_waves) {
        uint64_t kk[K];
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const uint64_t i = tile * (64 * K) + (uint64_t)j * 64 + lane;
            kk[j] = i < cap ? keys[i] : PMX_EMPTY_KEY;
            cnt += kk[j] != PMX_EMPTY_KEY ? 1 : 0;
        }
        int incl = cnt;
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        const int total = __shfl(incl, 63);
        if (total == 0) continue;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(n_out, (unsigned long long)total);
        base = __shfl(base, 0);
        unsigned long long pos = base + (unsigned long long)(incl - cnt);
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (kk[j] != PMX_EMPTY_KEY) {
                const uint64_t i = tile * (64 * K) + (uint64_t)j * 64 + lane;
                out_hash[pos] = kk[j];
                out_count[pos] = (int64_t)vals[i];
                ++pos;
            }
        }
    }
}

// ------------------------------------------------------------------------------- finalise
// dead[i] = 1 for the 4 homopolymer k-mer hashes (src/placement.cpp:1708-1722)
__global__ void k_mark_homopolymer(const uint64_t* __restrict__ hash, int64_t n, uint64_t h0, uint64_t h1, uint64_t h2, uint64_t h3,
                                   uint8_t* dead) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t h = hash[i];
        dead[i] = (h == h0 || h == h1 || h == h2 || h == h3) ? 1 : 0;
    }
}

// keys for the top-fraction mask: ascending sort of ~count keeps ascending hash among equal counts
__global__ void k_mask_keys(const int64_t* __restrict__ count, const uint8_t* __restrict__ dead, int64_t n, uint64_t* key, uint32_t* idx) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        key[i] = dead[i] ? UINT64_MAX : ~(uint64_t)count[i];
        idx[i] = (uint32_t)i;
    }
}
__global__ void k_mask_apply(const uint32_t* __restrict__ idx, int64_t n_mask, uint8_t* dead) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_mask; i += (int64_t)gridDim.x * blockDim.x) dead[idx[i]] = 1;
}

// integer statistics for resolveMinReadSupport (src/placement.cpp:931-955):
// stats[0]=sum of counts>=2, [1]=number of counts>=2, [2]=sum of all alive counts, [3]=alive entries
__global__ void k_hist_stats(const int64_t* __restrict__ count, const uint8_t* __restrict__ dead, int64_t n, unsigned long long* stats) {
    unsigned long long s2 = 0, c2 = 0, tot = 0, alive = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (dead[i]) continue;
        const int64_t c = count[i];
        ++alive; tot += (unsigned long long)c;
        if (c >= 2) { s2 += (unsigned long long)c; ++c2; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        s2 += __shfl_down(s2, o); c2 += __shfl_down(c2, o); tot += __shfl_down(tot, o); alive += __shfl_down(alive, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&stats[0], s2); atomicAdd(&stats[1], c2); atomicAdd(&stats[2], tot); atomicAdd(&stats[3], alive);
    }
}

__global__ void k_keep_flags(const int64_t* __restrict__ count, const uint8_t* __restrict__ dead, int64_t n, int64_t min_support,
                             uint32_t* flag) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        flag[i] = (!dead[i] && count[i] >= min_support) ? 1u : 0u;
}

// stable compaction (pos = exclusive scan of flag) + log1p of the read count (src/placement.cpp:970)
__global__ void k_keep_scatter(const uint64_t* __restrict__ hash, const int64_t* __restrict__ count, const uint32_t* __restrict__ flag,
                               const uint32_t* __restrict__ pos, int64_t n, uint64_t* kept_hash, double* kept_log) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (!flag[i]) continue;
        kept_hash[pos[i]] = hash[i];
        kept_log[pos[i]] = log1p_count(count[i]);
    }
}

// Canonical-order FP sums (same order as oracle_place.c): kept seeds in ascending
// hash order, sequential inside consecutive blocks of PMX_SUM_BLOCK (one thread per block), then the
// block sums sequentially (one thread).  partial[2*b] = sum L^2, partial[2*b+1] = sum L.
// (one wave per block: the lanes fetch 64 values at a time, coalesced, into LDS; lane 0 walks them for sum L^2, lane 1
//  for sum L -- the same sequential additions in the same order, without 1024 strided dependent loads per thread)
__global__ void __launch_bounds__(64) k_block_sums(const double* __restrict__ kept_log, int64_t n, double* partial) {
    __shared__ double sh[2][64];
    const int lane = threadIdx.x;
    const int64_t nb = (n + PMX_SUM_BLOCK - 1) / PMX_SUM_BLOCK;
    for (int64_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const int64_t beg = b * PMX_SUM_BLOCK, end = beg + PMX_SUM_BLOCK < n ? beg + PMX_SUM_BLOCK : n;
        double acc = 0.0;   // lane 0: sum L^2, lane 1: sum L
        double nxt = beg + lane < end ? kept_log[beg + lane] : 0.0;
        int buf = 0;
        for (int64_t base = beg; base < end; base += 64, buf ^= 1) {
            const double cur = nxt;
            if (base + 64 + lane < end) nxt = kept_log[base + 64 + lane];
            sh[buf][lane] = cur;
            __builtin_amdgcn_wave_barrier();
            const int cnt = (int)(end - base < 64 ? end - base : 64);
            if (lane < 2) {   // eight LDS reads in flight, then the eight dependent additions in order
                const double* v = sh[buf];
                int j = 0;
                for (; j + 8 <= cnt; j += 8) {
                    double a0 = v[j], a1 = v[j + 1], a2 = v[j + 2], a3 = v[j + 3], a4 = v[j + 4], a5 = v[j + 5], a6 = v[j + 6], a7 = v[j + 7];
                    if (lane == 0) { a0 *= a0; a1 *= a1; a2 *= a2; a3 *= a3; a4 *= a4; a5 *= a5; a6 *= a6; a7 *= a7; }
                    acc += a0; acc += a1; acc += a2; acc += a3; acc += a4; acc += a5; acc += a6; acc += a7;
                }
                for (; j < cnt; ++j) { const double L = v[j]; acc += lane == 0 ? L * L : L; }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (lane < 2) partial[2 * b + lane] = acc;
    }
}
__global__ void __launch_bounds__(64) k_sequential_sums(const double* __restrict__ partial, int64_t nb, double* out) {
    __shared__ double sh[2][128];
    if (blockIdx.x != 0) return;
    const int lane = threadIdx.x;
    double acc = 0.0;   // lane 0: sum of partial[2b], lane 1: sum of partial[2b+1], b ascending
    int buf = 0;
    for (int64_t base = 0; base < 2 * nb; base += 128, buf ^= 1) {
        if (base + lane < 2 * nb) sh[buf][lane] = partial[base + lane];
        if (base + 64 + lane < 2 * nb) sh[buf][64 + lane] = partial[base + 64 + lane];
        __builtin_amdgcn_wave_barrier();
        const int cnt = (int)(2 * nb - base < 128 ? 2 * nb - base : 128);
        if (lane < 2) {
            const double* v = sh[buf];
            int j = lane;
            for (; j + 14 < cnt; j += 16) {
                const double a0 = v[j], a1 = v[j + 2], a2 = v[j + 4], a3 = v[j + 6], a4 = v[j + 8], a5 = v[j + 10], a6 = v[j + 12], a7 = v[j + 14];
                acc += a0; acc += a1; acc += a2; acc += a3; acc += a4; acc += a5; acc += a6; acc += a7;
            }
            for (; j < cnt; j += 2) acc += v[j];
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane < 2) out[lane] = acc;
}

__global__ void k_kept_table_build(const uint64_t* __restrict__ kept_hash, const double* __restrict__ kept_log, int64_t n,
                                   uint64_t* tkeys, double* tvals, uint64_t mask) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t key = kept_hash[i];
        uint64_t slot = mix64(key) & mask;
        while (true) {
            unsigned long long prev = atomicCAS((unsigned long long*)&tkeys[slot], (unsigned long long)PMX_EMPTY_KEY, (unsigned long long)key);
            if (prev == PMX_EMPTY_KEY) { tvals[slot] = kept_log[i]; break; }
            slot = (slot + 1) & mask;
        }
    }
}

__device__ __forceinline__ bool kept_lookup(const uint64_t* __restrict__ tkeys, const double* __restrict__ tvals, uint64_t mask,
                                            uint64_t key, double* L) {
    uint64_t slot = mix64(key) & mask;
    while (true) {
        const uint64_t k = tkeys[slot];
        if (k == key) { *L = tvals[slot]; return true; }
        if (k == PMX_EMPTY_KEY) return false;
        slot = (slot + 1) & mask;
    }
}

// weighted-containment denominator: root changes in stored order (src/placement.cpp:1863-1876).  One wave:
// lanes probe 64 changes at a time, then the additions run in stored order.
__global__ void __launch_bounds__(1024)
k_wc_denominator(const uint64_t* __restrict__ ch_hash, const int16_t* __restrict__ ch_child, uint64_t beg, uint64_t end,
                 const uint64_t* __restrict__ tkeys, const double* __restrict__ tvals, uint64_t mask, int has_kept, double* out) {
    // one block: every thread probes its changes in parallel into an LDS tile, then thread 0 adds the tile in
    // stored order (a no-hit change contributes +0.0, which is exact; see k_score_level)
    __shared__ double tile[1024];
    if (blockIdx.x != 0) return;
    double acc = 0.0;
    for (uint64_t base = beg; base < end; base += blockDim.x) {
        const uint64_t i = base + threadIdx.x;
        double inv = 0.0;
        if (i < end) {
            const int16_t cc = ch_child[i];
            double L;
            if (cc > 0 && has_kept && kept_lookup(tkeys, tvals, mask, ch_hash[i], &L)) inv = 1.0 / (double)cc;
        }
        tile[threadIdx.x] = inv;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int cnt = (int)((end - base) < blockDim.x ? (end - base) : blockDim.x);
            int j = 0;
            for (; j + 8 <= cnt; j += 8) {
                const double v0 = tile[j], v1 = tile[j + 1], v2 = tile[j + 2], v3 = tile[j + 3], v4 = tile[j + 4], v5 = tile[j + 5],
                             v6 = tile[j + 6], v7 = tile[j + 7];
                acc += v0; acc += v1; acc += v2; acc += v3; acc += v4; acc += v5; acc += v6; acc += v7;
            }
            for (; j < cnt; ++j) acc += tile[j];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = acc;
}

// --------------------------------------------------------------------------- node scoring
// Child state = parent state + own deltas applied in stored order (src/placement.cpp:242-345, :772-774).
// The floating-point additions are replayed strictly in stored order so every accumulator sees the
// reference's operation sequence (the integer counters are order-free and are wave-reduced).  A BFS level
// lasts as long as its largest node, so the per-change critical path is what matters: everything that does
// not depend on the order (log1p, table probe, the terms) is computed for all changes at once in pass 1;
// pass 2 streams the terms (two 64-change chunks in flight), broadcasts them with v_readlane (no LDS round
// trip) and visits only the changes that hit the kept-seed table for the four hit-only accumulators
// (skipping a no-hit change is what the reference's control flow does).
__device__ __forceinline__ double readlane_f64(double v, int j) {
    const long long b = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, j);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)b >> 32), j);
    return __longlong_as_double((long long)((unsigned long long)hi << 32 | lo));
}

// Pass 1 (one launch, every seed change of the index in parallel): the order-independent part -- log1p,
// kept-seed table probe, the five per-change terms -- written as SoA so pass 2 streams them coalesced.
// meta: bit 0 = hit, bits 1-2 = d_pres + 1, bits 3-4 = d_uniq + 1.
__global__ void __launch_bounds__(256)
k_score_terms(const uint64_t* __restrict__ ch_hash, const int16_t* __restrict__ ch_par, const int16_t* __restrict__ ch_child,
              int64_t n_changes, const uint64_t* __restrict__ tkeys, const double* __restrict__ tvals, uint64_t mask, int has_kept,
              double* __restrict__ t_mag, double* __restrict__ t_raw, double* __restrict__ t_cos, double* __restrict__ t_wc,
              double* __restrict__ t_lc, uint8_t* __restrict__ t_meta) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_changes; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pc = ch_par[i], cc = ch_child[i];
        const double logC = cc > 0 ? log1p_count(cc) : 0.0;
        const double logP = pc > 0 ? log1p_count(pc) : 0.0;
        t_mag[i] = logC * logC - logP * logP;
        const int d_uniq = (cc > 0) - (pc > 0);
        int d_pres = 0, hit = 0;
        double L, d_raw = 0.0, d_cos = 0.0, d_wc = 0.0, d_lc = 0.0;
        if (cc != pc && has_kept && kept_lookup(tkeys, tvals, mask, ch_hash[i], &L)) {
            hit = 1;
            d_pres = (int)((pc == 0) & (cc != 0)) - (int)((cc == 0) & (pc != 0));
            const double o1 = pc > 0 ? L / (double)pc : 0.0, n1 = cc > 0 ? L / (double)cc : 0.0;
            d_raw = n1 - o1;
            d_cos = L * (logC - logP);
            const double o2 = pc > 0 ? 1.0 / (double)pc : 0.0, n2 = cc > 0 ? 1.0 / (double)cc : 0.0;
            d_wc = n2 - o2;
            d_lc = (double)d_pres * L;
        }
        t_raw[i] = d_raw; t_cos[i] = d_cos; t_wc[i] = d_wc; t_lc[i] = d_lc;
        t_meta[i] = (uint8_t)(hit | (d_pres + 1) << 1 | (d_uniq + 1) << 3);
    }
}

struct ScoreTerms {
    double d_raw, d_cos, d_wc, d_lc, d_mag;
    int d_pres, d_uniq, hit;
};

__device__ __forceinline__ ScoreTerms load_terms(uint64_t i, uint64_t end, const double* __restrict__ t_mag, const double* __restrict__ t_raw,
                                                 const double* __restrict__ t_cos, const double* __restrict__ t_wc,
                                                 const double* __restrict__ t_lc, const uint8_t* __restrict__ t_meta) {
    ScoreTerms t;
    t.d_raw = t.d_cos = t.d_wc = t.d_lc = t.d_mag = 0.0;
    t.d_pres = t.d_uniq = t.hit = 0;
    if (i < end) {   // all six loads are independent: nothing waits until the terms are used two chunks later
        const int m = t_meta[i];
        t.d_mag = t_mag[i];
        t.d_raw = t_raw[i]; t.d_cos = t_cos[i]; t.d_wc = t_wc[i]; t.d_lc = t_lc[i];
        t.hit = m & 1;
        t.d_pres = (m & 1) ? ((m >> 1) & 3) - 1 : 0;
        t.d_uniq = ((m >> 3) & 3) - 1;
    }
    return t;
}

// wave-wide integer sum through DPP (quad swaps, then the two row mirrors: every lane of a 16-lane row holds the row's
// total; four scalar reads add the rows): ~10 instructions instead of six ds_bpermute round trips through the LDS
// crossbar, each followed by a wait -- the sums sit in front of every chunk of the scoring kernels' serial add chains
__device__ __forceinline__ int wave_sum_int(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);   // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

// Pass 2, one launch per BFS level, one wave per node: parent state + the node's terms added in stored order.
// The five accumulators are five independent serial chains, so they run in five LANES: each 64-change chunk
// is staged in LDS as [change][accumulator] and lanes 0..4 walk it with one ds_read + one v_add_f64 per
// change (a no-hit change carries +0.0 for the four hit-only terms: adding +0.0 is exact, and no accumulator
// can be -0.0).  The critical path per change is one FP64 add instead of a dozen broadcast instructions.
__global__ void __launch_bounds__(256)
k_score_level(const uint32_t* __restrict__ level_nodes, int64_t n_level, const uint32_t* __restrict__ parent,
              const uint64_t* __restrict__ offsets, const double* __restrict__ t_mag, const double* __restrict__ t_raw,
              const double* __restrict__ t_cos, const double* __restrict__ t_wc, const double* __restrict__ t_lc,
              const uint8_t* __restrict__ t_meta, double* metrics5, int64_t* counts2) {
    __shared__ double stage[4][64 * 5];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int64_t wv = (int64_t)blockIdx.x * (blockDim.x >> 6) + wib;
    if (wv >= n_level) return;
    const uint32_t nd = level_nodes[wv];
    double* sh = stage[wib];
    double acc = 0.0;              // lane k < 5: accumulator k (raw, cos, wc, lc, mag)
    int64_t c0 = 0, c1 = 0;
    const uint64_t beg = offsets[nd], end = offsets[nd + 1];
    ScoreTerms cur = load_terms(beg + lane, end, t_mag, t_raw, t_cos, t_wc, t_lc, t_meta);
    ScoreTerms nx1 = load_terms(beg + 64 + lane, end, t_mag, t_raw, t_cos, t_wc, t_lc, t_meta);
    if (nd != 0) {
        const uint32_t pa = parent[nd];
        if (lane < 5) acc = metrics5[5 * (size_t)pa + lane];
        c0 = counts2[2 * (size_t)pa + 0]; c1 = counts2[2 * (size_t)pa + 1];
    }
    for (uint64_t base = beg; base < end; base += 64) {
        // two chunks of terms are in flight while this one is added up
        const ScoreTerms nx2 = load_terms(base + 128 + lane, end, t_mag, t_raw, t_cos, t_wc, t_lc, t_meta);
        const int cnt = (int)((end - base) < 64 ? (end - base) : 64);
        c1 += wave_sum_int(cur.d_uniq);
        c0 += wave_sum_int(cur.d_pres);          // d_pres is 0 unless hit
        sh[lane * 5 + 0] = cur.d_raw; sh[lane * 5 + 1] = cur.d_cos; sh[lane * 5 + 2] = cur.d_wc; sh[lane * 5 + 3] = cur.d_lc;
        sh[lane * 5 + 4] = cur.d_mag;
        __builtin_amdgcn_wave_barrier();          // same-wave LDS traffic is ordered; this only pins the compiler
        if (lane < 5) {
            int j = 0;
            for (; j + 8 <= cnt; j += 8) {
                const double v0 = sh[(j + 0) * 5 + lane], v1 = sh[(j + 1) * 5 + lane], v2 = sh[(j + 2) * 5 + lane],
                             v3 = sh[(j + 3) * 5 + lane], v4 = sh[(j + 4) * 5 + lane], v5 = sh[(j + 5) * 5 + lane],
                             v6 = sh[(j + 6) * 5 + lane], v7 = sh[(j + 7) * 5 + lane];
                acc += v0; acc += v1; acc += v2; acc += v3; acc += v4; acc += v5; acc += v6; acc += v7;
            }
            for (; j < cnt; ++j) acc += sh[j * 5 + lane];
        }
        __builtin_amdgcn_wave_barrier();
        cur = nx1;
        nx1 = nx2;
    }
    if (lane < 5) metrics5[5 * (size_t)nd + lane] = acc;
    if (lane == 0) { counts2[2 * (size_t)nd + 0] = c0; counts2[2 * (size_t)nd + 1] = c1; }
}

// Pass 2 as ONE persistent launch: wave w takes the BFS positions w, w + W, w + 2W, ... (W = resident waves) and
// waits for its node's parent through a per-node flag instead of a kernel boundary per level.  A parent always
// sits at an earlier BFS position, and every wave works through its positions in ascending order, so the wave that
// owns the earliest unfinished position never waits on anything unfinished: no deadlock as long as all W waves are
// resident (the host launches at most one workgroup per CU).  The arithmetic per node is k_score_level's, bit for bit.
// done[] holds the epoch of the call that last finished the node (no clearing between calls); a wave that polls
// longer than any sane run sets *status and every wave leaves.
__global__ void __launch_bounds__(256)
k_score_tree(const uint32_t* __restrict__ order, int64_t n_nodes, const uint32_t* __restrict__ parent, const uint64_t* __restrict__ offsets,
             const double* __restrict__ t_mag, const double* __restrict__ t_raw, const double* __restrict__ t_cos,
             const double* __restrict__ t_wc, const double* __restrict__ t_lc, const uint8_t* __restrict__ t_meta, double* metrics5,
             int64_t* counts2, uint32_t* done, uint32_t epoch, uint32_t* status) {
    __shared__ double stage[4][64 * 5];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    double* sh = stage[wib];
    for (int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + wib; p < n_nodes; p += n_waves) {
        const uint32_t nd = order[p];
        double acc = 0.0;              // lane k < 5: accumulator k (raw, cos, wc, lc, mag)
        int64_t c0 = 0, c1 = 0;
        const uint64_t beg = offsets[nd], end = offsets[nd + 1];
        // the node's own terms do not depend on the parent: in flight while the wave waits for it
        ScoreTerms cur = load_terms(beg + lane, end, t_mag, t_raw, t_cos, t_wc, t_lc, t_meta);
        ScoreTerms nx1 = load_terms(beg + 64 + lane, end, t_mag, t_raw, t_cos, t_wc, t_lc, t_meta);
        if (nd != 0) {
            const uint32_t pa = parent[nd];
            uint32_t polls = 0;
            bool ok = true;
            while (__hip_atomic_load(&done[pa], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                __builtin_amdgcn_s_sleep(2);
                // (the shared status word is looked at every 1024 polls only: one word read by every waiting wave is a hot spot)
                if (++polls > (1u << 24) || ((polls & 1023u) == 0u && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) { ok = false; break; }
            }
            if (!ok) {   // uniform: every lane polls the same flag
                if (lane == 0) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (lane < 5) acc = metrics5[5 * (size_t)pa + lane];
            c0 = counts2[2 * (size_t)pa + 0]; c1 = counts2[2 * (size_t)pa + 1];
        }
        for (uint64_t base = beg; base < end; base += 64) {
            const ScoreTerms nx2 = load_terms(base + 128 + lane, end, t_mag, t_raw, t_cos, t_wc, t_lc, t_meta);
            const int cnt = (int)((end - base) < 64 ? (end - base) : 64);
            c1 += wave_sum_int(cur.d_uniq);
            c0 += wave_sum_int(cur.d_pres);
            sh[lane * 5 + 0] = cur.d_raw; sh[lane * 5 + 1] = cur.d_cos; sh[lane * 5 + 2] = cur.d_wc; sh[lane * 5 + 3] = cur.d_lc;
            sh[lane * 5 + 4] = cur.d_mag;
            __builtin_amdgcn_wave_barrier();
            if (lane < 5) {
                int j = 0;
                for (; j + 8 <= cnt; j += 8) {
                    const double v0 = sh[(j + 0) * 5 + lane], v1 = sh[(j + 1) * 5 + lane], v2 = sh[(j + 2) * 5 + lane],
                                 v3 = sh[(j + 3) * 5 + lane], v4 = sh[(j + 4) * 5 + lane], v5 = sh[(j + 5) * 5 + lane],
                                 v6 = sh[(j + 6) * 5 + lane], v7 = sh[(j + 7) * 5 + lane];
                    acc += v0; acc += v1; acc += v2; acc += v3; acc += v4; acc += v5; acc += v6; acc += v7;
                }
                for (; j < cnt; ++j) acc += sh[j * 5 + lane];
            }
            __builtin_amdgcn_wave_barrier();
            cur = nx1;
            nx1 = nx2;
        }
        if (lane < 5) metrics5[5 * (size_t)nd + lane] = acc;
        if (lane == 0) { counts2[2 * (size_t)nd + 0] = c0; counts2[2 * (size_t)nd + 1] = c1; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // the wave's stores are visible before the flag
        if (lane == 0) __hip_atomic_store(&done[nd], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Pass 2, heavy-path form: the host cuts the tree into chains (a node followed by its heaviest child, and so on
// down to a leaf) and one wave walks one chain with the five accumulators staying in registers from node to node.
// Only the HEAD of a chain waits on a flag (its parent lies in a chain that started earlier), and only nodes with
// two or more children publish one: a root-to-leaf path crosses O(log n) chain boundaries instead of one kernel
// boundary (or flag) per level.  chain_nodes[k] = node id | publish << 31; chain_beg/end[k] = the node's change
// range, gathered in chain order so that the next node's terms are requested while this one is being added up.
// Chains are sorted by the BFS position of their head and wave w takes chains w, w + W, ...: the chain that holds
// a head's parent starts earlier, so the earliest unfinished chain never waits on an unfinished one (all W waves
// are resident: at most one workgroup per CU).  Per node the arithmetic is k_score_level's, bit for bit.
#define PMX_CHAIN_FLUSH 8u
#define PMX_CHAIN_AHEAD 3
__global__ void __launch_bounds__(256)
k_score_chains(const uint32_t* __restrict__ chain_off, int64_t n_chains, const uint32_t* __restrict__ chain_nodes,
               const uint64_t* __restrict__ chain_beg, const uint64_t* __restrict__ chain_end, const uint32_t* __restrict__ parent,
               const double* __restrict__ t_mag, const double* __restrict__ t_raw, const double* __restrict__ t_cos,
               const double* __restrict__ t_wc, const double* __restrict__ t_lc, const uint8_t* __restrict__ t_meta, double* metrics5,
               int64_t* counts2, uint32_t* done, uint32_t epoch, uint32_t* status) {
    __shared__ double stage[4][64 * 5];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    double* sh = stage[wib];
    for (int64_t c = (int64_t)blockIdx.x * (blockDim.x >> 6) + wib; c < n_chains; c += n_waves) {
        const uint32_t cb = chain_off[c], ce = chain_off[c + 1];
        // Software pipeline over the chain: the first two 64-change chunks of the nodes k+1 .. k+PMX_CHAIN_AHEAD are
        // in flight while node k is added up (a node's own arithmetic is a fraction of a microsecond, a load from
        // HBM two), and the (node, range) records run one node further ahead still.
        constexpr int AH = PMX_CHAIN_AHEAD;
        uint32_t m_ent[AH + 2];
        uint64_t m_beg[AH + 2], m_end[AH + 2];
#pragma unroll
        for (int a_ = 0; a_ < AH + 2; ++a_) {
            const bool in = cb + (uint32_t)a_ < ce;
            m_ent[a_] = in ? chain_nodes[cb + a_] : 0u;
            m_beg[a_] = in ? chain_beg[cb + a_] : 0;
            m_end[a_] = in ? chain_end[cb + a_] : 0;
        }
        ScoreTerms q_cur[AH + 1], q_nx1[AH + 1];
#pragma unroll
        for (int a_ = 0; a_ < AH + 1; ++a_) {
            q_cur[a_] = load_terms(m_beg[a_] + lane, m_end[a_], t_mag, t_raw, t_cos, t_wc, t_lc, t_meta);
            q_nx1[a_] = load_terms(m_beg[a_] + 64 + lane, m_end[a_], t_mag, t_raw, t_cos, t_wc, t_lc, t_meta);
        }
        uint32_t ent = m_ent[0];
        double acc = 0.0;              // lane k < 5: accumulator k (raw, cos, wc, lc, mag)
        int64_t c0 = 0, c1 = 0;
        const uint32_t head = ent & 0x7fffffffu;
        if (head != 0) {
            const uint32_t pa = parent[head];
            uint32_t polls = 0;
            bool ok = true;
            while (__hip_atomic_load(&done[pa], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                __builtin_amdgcn_s_sleep(1);
                // (the shared status word is looked at every 1024 polls only: one word read by every waiting wave is a hot spot)
                if (++polls > (1u << 24) || ((polls & 1023u) == 0u && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) { ok = false; break; }
            }
            if (!ok) {   // uniform: every lane polls the same flag
                if (lane == 0) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (lane < 5) acc = metrics5[5 * (size_t)pa + lane];
            c0 = counts2[2 * (size_t)pa + 0]; c1 = counts2[2 * (size_t)pa + 1];
        }
        // Flags are published in batches: the release fence (an L2 write-back plus a wait, ~2-6 us) would otherwise sit
        // on the chain's critical path at every node that has a second child -- on a caterpillar-shaped tree that is
        // every node of the spine.  One fence per PMX_CHAIN_FLUSH publishing nodes (and one at the chain's end)
        // covers all their stores; the lanes then set the flags of chain positions [k_from, k] in one instruction.
        uint32_t k_from = cb, n_pend = 0;
        for (uint32_t k = cb; k < ce; ++k) {
            ent = m_ent[0];
            const uint32_t nd = ent & 0x7fffffffu;
            const bool publish = (ent >> 31) != 0;
            const uint64_t beg = m_beg[0], end = m_end[0];
            ScoreTerms cur = q_cur[0], nx1 = q_nx1[0];
            // request the chunks of node k + AH + 1 (its record arrived an iteration ago) and the record after that
            const ScoreTerms f_cur = load_terms(m_beg[AH + 1] + lane, m_end[AH + 1], t_mag, t_raw, t_cos, t_wc, t_lc, t_meta);
            const ScoreTerms f_nx1 = load_terms(m_beg[AH + 1] + 64 + lane, m_end[AH + 1], t_mag, t_raw, t_cos, t_wc, t_lc, t_meta);
            const bool in_f = k + (uint32_t)(AH + 2) < ce;
            const uint32_t f_ent = in_f ? chain_nodes[k + AH + 2] : 0u;
            const uint64_t f_beg = in_f ? chain_beg[k + AH + 2] : 0, f_end = in_f ? chain_end[k + AH + 2] : 0;
            for (uint64_t base = beg; base < end; base += 64) {
                const ScoreTerms nx2 = load_terms(base + 128 + lane, end, t_mag, t_raw, t_cos, t_wc, t_lc, t_meta);
                const int cnt = (int)((end - base) < 64 ? (end - base) : 64);
                c1 += wave_sum_int(cur.d_uniq);
                c0 += wave_sum_int(cur.d_pres);
                sh[lane * 5 + 0] = cur.d_raw; sh[lane * 5 + 1] = cur.d_cos; sh[lane * 5 + 2] = cur.d_wc; sh[lane * 5 + 3] = cur.d_lc;
                sh[lane * 5 + 4] = cur.d_mag;
                __builtin_amdgcn_wave_barrier();
                if (lane < 5) {
                    // eight terms per step, the next eight requested from LDS before these are added: the chain of FP64
                    // adds is the kernel's critical path and no longer waits for an LDS round trip per step
                    int j = 0;
                    if (cnt >= 8) {
                        const double* q = sh + lane;
                        double a0 = q[0], a1 = q[5], a2 = q[10], a3 = q[15], a4 = q[20], a5 = q[25], a6 = q[30], a7 = q[35];
                        for (; j + 16 <= cnt; j += 8) {
                            q += 40;
                            const double b0 = q[0], b1 = q[5], b2 = q[10], b3 = q[15], b4 = q[20], b5 = q[25], b6 = q[30], b7 = q[35];
                            acc += a0; acc += a1; acc += a2; acc += a3; acc += a4; acc += a5; acc += a6; acc += a7;
                            a0 = b0; a1 = b1; a2 = b2; a3 = b3; a4 = b4; a5 = b5; a6 = b6; a7 = b7;
                        }
                        acc += a0; acc += a1; acc += a2; acc += a3; acc += a4; acc += a5; acc += a6; acc += a7;
                        j += 8;
                    }
                    for (; j < cnt; ++j) acc += sh[j * 5 + lane];
                }
                __builtin_amdgcn_wave_barrier();
                cur = nx1;
                nx1 = nx2;
            }
            if (lane < 5) metrics5[5 * (size_t)nd + lane] = acc;
            if (lane == 0) { counts2[2 * (size_t)nd + 0] = c0; counts2[2 * (size_t)nd + 1] = c1; }
            n_pend += publish ? 1u : 0u;   // children in other chains wait for this node
            if (n_pend >= PMX_CHAIN_FLUSH || (k + 1 == ce && n_pend > 0)) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                for (uint32_t q = k_from + (uint32_t)lane; q <= k; q += 64u) {
                    const uint32_t e = chain_nodes[q];
                    if (e >> 31) __hip_atomic_store(&done[e & 0x7fffffffu], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                k_from = k + 1;
                n_pend = 0;
            }
#pragma unroll
            for (int a_ = 0; a_ < AH; ++a_) { q_cur[a_] = q_cur[a_ + 1]; q_nx1[a_] = q_nx1[a_ + 1]; }
            q_cur[AH] = f_cur; q_nx1[AH] = f_nx1;
#pragma unroll
            for (int a_ = 0; a_ < AH + 1; ++a_) { m_ent[a_] = m_ent[a_ + 1]; m_beg[a_] = m_beg[a_ + 1]; m_end[a_] = m_end[a_ + 1]; }
            m_ent[AH + 1] = f_ent; m_beg[AH + 1] = f_beg; m_end[AH + 1] = f_end;
        }
    }
}

// score getters (src/placement.hpp:120-149); TSV order log_raw, log_cosine, containment,
// weighted_containment, log_containment
// scores5[node][metric] for the node outputs, and scores_bfs[metric][visit position] for the host's sequential
// best/tie rule (one contiguous array per metric, in the order the rule visits the nodes)
// (the three denominators are read from the device scalars [sum L^2, sum L, wc_den]: no host round trip before this launch)
__global__ void k_score_getters(const double* __restrict__ metrics5, const int64_t* __restrict__ counts2, int64_t n_nodes,
                                const double* __restrict__ scalars, int64_t n_kept, double* scores5,
                                const uint32_t* __restrict__ order, double* scores_bfs) {
    const double log_mag = sqrt(scalars[0]), log_cont_den = scalars[1], wc_den = scalars[2];
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_nodes; j += (int64_t)gridDim.x * blockDim.x) {
        const int64_t nd = (int64_t)order[j];
        const double* m = metrics5 + 5 * nd;
        double* sc = scores5 + 5 * nd;
        sc[0] = log_mag <= 0.0 ? 0.0 : m[0] / log_mag;
        const double gm = sqrt(m[4]);
        if (log_mag <= 0.0 || gm <= 0.0) sc[1] = 0.0;
        else {
            const double v = m[1] / (log_mag * gm);
            sc[1] = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
        }
        sc[2] = n_kept > 0 ? (double)(uint64_t)counts2[2 * nd] / (double)(uint64_t)n_kept : 0.0;
        sc[3] = wc_den > 0.0 ? m[2] / wc_den : 0.0;
        sc[4] = log_cont_den > 0.0 ? m[3] / log_cont_den : 0.0;
        for (int q = 0; q < 5; ++q) scores_bfs[(size_t)q * (size_t)n_nodes + (size_t)j] = sc[q];
    }
}

__global__ void k_fill_u64(uint64_t* p, uint64_t v, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = v;
}

}  // namespace pmx
