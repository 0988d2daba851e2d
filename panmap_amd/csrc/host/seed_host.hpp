// Host-side seeding primitives used by the (CPU, one-time) index stage: ntHash-style rolling
// hashes and syncmer detection over a node's local genome window.
// Behaviour follows src/seeding.cpp:47-229 / src/seeding.hpp:100-120 of the reference; the
// read-side (hot) seeding runs on the GPU in seed_kernels.hip with the same definitions.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace pmx {

static inline uint64_t h_rol(uint64_t h, unsigned r) { r &= 63u; return r ? (h << r) | (h >> (64u - r)) : h; }
static inline uint64_t h_ror(uint64_t h, unsigned r) { r &= 63u; return r ? (h >> r) | (h << (64u - r)) : h; }

// per-base constants (A,C,G,T; anything else 0); index 0..3 = A,C,G,T
static const uint64_t kBaseHash[4] = {0x3c8bfbb395c60474ULL, 0x3193c18562a02b4cULL, 0x20323ed082572324ULL,
                                      0x295549f54be24456ULL};

static inline int base_code(char c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;
    }
}
static inline uint64_t base_hash(int code) { return code < 4 ? kBaseHash[code] : 0; }
static inline uint64_t base_hash_comp(int code) { return code < 4 ? kBaseHash[3 - code] : 0; }

struct SyncmerParams {
    int k = 19, s = 8, t = 0, l = 3;
    bool open = false;
    // --meta: a k-min-mer that reads right-to-left on the genome (R < F) is keyed hash ^ kOrientXor, so that the index
    // counts the two orientations of one hash apart (what the reference's MGSR index keeps as seedInfos[].isReverse)
    bool oriented = false;
};
static const uint64_t kOrientXor = 0x9e3779b97f4a7c15ULL;   // == PMX_ORIENT_XOR (include/panmap_amd.h)

// For every k-mer start i in [0, n-k]: is_sync[i] and, when set, hash[i] = min(F,R).
// Windows with a non-ACGT base or F==R are never syncmers.
inline void host_syncmers(const char* seq, int64_t n, const SyncmerParams& p, std::vector<uint8_t>& is_sync,
                          std::vector<uint64_t>& hash) {
    const int k = p.k, s = p.s, t = p.t;
    is_sync.clear();
    hash.clear();
    if (n < k) return;
    const int64_t nk = n - k + 1, ns = n - s + 1;
    const int w = k - s + 1;
    is_sync.assign(nk, 0);
    hash.assign(nk, 0);
    std::vector<uint64_t> fS(ns), rS(ns);
    std::vector<uint8_t> code(n);
    for (int64_t i = 0; i < n; ++i) code[i] = (uint8_t)base_code(seq[i]);
    uint64_t f = 0, r = 0;
    for (int q = 0; q < s; ++q) {
        f ^= h_rol(base_hash(code[q]), (unsigned)(s - 1 - q));
        r ^= h_rol(base_hash_comp(code[q]), (unsigned)q);
    }
    fS[0] = f;
    rS[0] = r;
    for (int64_t j = 1; j < ns; ++j) {
        int out = code[j - 1], in = code[j + s - 1];
        f = h_rol(f, 1) ^ h_rol(base_hash(out), (unsigned)s) ^ base_hash(in);
        r = h_ror(r, 1) ^ h_ror(base_hash_comp(out), 1) ^ h_rol(base_hash_comp(in), (unsigned)(s - 1));
        fS[j] = f;
        rS[j] = r;
    }
    uint64_t fK = 0, rK = 0;
    int64_t last_amb = -1;
    for (int q = 0; q < k; ++q) {
        fK ^= h_rol(base_hash(code[q]), (unsigned)(k - 1 - q));
        rK ^= h_rol(base_hash_comp(code[q]), (unsigned)q);
        if (code[q] > 3) last_amb = q;
    }
    for (int64_t i = 0; i < nk; ++i) {
        if (i > 0) {
            int out = code[i - 1], in = code[i + k - 1];
            fK = h_rol(fK, 1) ^ h_rol(base_hash(out), (unsigned)k) ^ base_hash(in);
            rK = h_ror(rK, 1) ^ h_ror(base_hash_comp(out), 1) ^ h_rol(base_hash_comp(in), (unsigned)(k - 1));
            if (in > 3) last_amb = i + k - 1;
        }
        if (last_amb >= i || fK == rK) continue;
        uint64_t fmin = UINT64_MAX, rmin = UINT64_MAX;
        for (int j = 0; j < w; ++j) {
            fmin = fS[i + j] < fmin ? fS[i + j] : fmin;
            rmin = rS[i + j] < rmin ? rS[i + j] : rmin;
        }
        bool fs, rs;
        if (p.open) {
            fs = fS[i + t] == fmin;
            rs = rS[i + k - s - t] == rmin;
        } else {
            fs = fS[i + t] == fmin || fS[i + k - s - t] == fmin;
            rs = rS[i + k - s - t] == rmin || rS[i + t] == rmin;
        }
        if (fs || rs) {
            is_sync[i] = 1;
            hash[i] = fK < rK ? fK : rK;
        }
    }
}

// k-min-mer seed of l consecutive syncmer hashes (src/placement.cpp:1650-1664,
// src/index_single_mode.cpp:1990-2008).  Returns false when the window yields no seed (F == R).
inline bool kminmer_seed(const uint64_t* h, int k, int l, uint64_t* out, bool oriented = false, bool* is_rev = nullptr) {
    if (l <= 1) {
        *out = h[0];       // (the orientation of a lone syncmer is its k-mer's: not kept on this path, l >= 2 for --meta)
        if (is_rev) *is_rev = false;
        return true;
    }
    uint64_t F = 0, R = 0;
    for (int q = 0; q < l; ++q) {
        F ^= h_rol(h[q], (unsigned)(k * (l - 1 - q)));
        R ^= h_rol(h[q], (unsigned)(k * q));
    }
    if (F == R) return false;
    const bool rev = R < F;
    *out = (rev ? R : F) ^ (oriented && rev ? kOrientXor : 0ULL);
    if (is_rev) *is_rev = rev;
    return true;
}

}  // namespace pmx
