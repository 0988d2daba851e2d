// Small math helpers shared by host and device code (PMX_HD functions compile under g++ and hipcc).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define PMX_HD __host__ __device__ __forceinline__
#if defined(PMX_FULL_INLINE)
#define PMX_HDN __host__ __device__ __forceinline__   // tier-1 kernel: one flat function so LDS address spaces can be inferred
#else
#define PMX_HDN __host__ __device__ __attribute__((noinline)) inline   // big phase functions: own register allocation
#endif
#else
#define PMX_HD inline
#define PMX_HDN inline
#endif

namespace pmx {

PMX_HD uint64_t rotl64(uint64_t h, unsigned r) { r &= 63u; return r ? (h << r) | (h >> (64u - r)) : h; }
PMX_HD uint64_t rotr64(uint64_t h, unsigned r) { r &= 63u; return r ? (h >> r) | (h << (64u - r)) : h; }

// table slot hash for open addressing (murmur3 finaliser); NOT a reference hash
PMX_HD uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

// Locality key of a read for launch orders (seeding, thread-per-pair alignment): the smallest hashed canonical 16-mer
// among the read's first 17 bases' worth of windows (positions 0..15, from its first packed word: 32 bases).  Reads
// that start within a few bases of each other on the same strand share it, so equal keys = neighbours on the genome.
// Only an ordering hint: any value is correct.
PMX_HD uint32_t read_locality_key(uint64_t first_word) {
    uint64_t best = ~0ULL;
    for (int p = 0; p < 16; ++p) {
        const uint64_t f = (first_word >> (2 * p)) & 0xffffffffULL;          // 16 bases, base p in the low bits
        uint64_t r = ~f & 0xffffffffULL;                                      // complement (codes 0..3 -> 3..0) ...
        r = (r & 0x33333333ULL) << 2 | (r >> 2 & 0x33333333ULL);              // ... reversed base by base
        r = (r & 0x0f0f0f0fULL) << 4 | (r >> 4 & 0x0f0f0f0fULL);
        r = (r & 0x00ff00ffULL) << 8 | (r >> 8 & 0x00ff00ffULL);
        r = (r & 0x0000ffffULL) << 16 | (r >> 16 & 0x0000ffffULL);
        const uint64_t h = mix64(f < r ? f : r);
        best = h < best ? h : best;
    }
    return (uint32_t)(best >> 32);
}

PMX_HD uint32_t dbl_hi(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)(u >> 32); }
PMX_HD double dbl_set_hi(double x, uint32_t hi) { uint64_t u; memcpy(&u, &x, 8); u = (u & 0xffffffffULL) | ((uint64_t)hi << 32); memcpy(&x, &u, 8); return x; }

// log1p(c) for an integer count 1 <= c < 2^52, bit-identical to glibc's double log1p
// (sysdeps/ieee754/dbl-64/s_log1p.c, the fdlibm algorithm with glibc's split polynomial).
// The reference takes log1p of seed counts on the host (src/placement.cpp:275-276, 970); the
// device evaluates the same IEEE-754 operation sequence (compile with -ffp-contract=off).
// Verified exhaustively against the host libm for c in [1, 2^24] (tests/test_math_parity.py).
PMX_HD double log1p_count(int64_t c) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
                 Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
                 Lp7 = 1.479819860511658591e-01;
    if (c <= 0) return 0.0;
    const double x = (double)c;
    double u = 1.0 + x;                 // exact for c < 2^52
    uint32_t hu = dbl_hi(u);
    int k = (int)(hu >> 20) - 1023;
    double cc = 0.0;                    // correction term: 1-(u-x) == 0 for exact u
    hu &= 0x000fffffu;
    if (hu < 0x6a09eu) {
        u = dbl_set_hi(u, hu | 0x3ff00000u);
    } else {
        k += 1;
        u = dbl_set_hi(u, hu | 0x3fe00000u);
        hu = (0x00100000u - hu) >> 2;
    }
    const double f = u - 1.0;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    if (hu == 0) {                      // |f| < 2^-20
        if (f == 0.0) { cc += dk * ln2_lo; return dk * ln2_hi + cc; }
        const double R = hfsq * (1.0 - 0.66666666666666666 * f);
        return dk * ln2_hi - ((R - (dk * ln2_lo + cc)) - f);
    }
    const double s = f / (2.0 + f);
    const double z = s * s;
#ifndef PMX_LOG1P_HORNER
    const double R1 = z * Lp1, z2 = z * z;
    const double R2 = Lp2 + z * Lp3, z4 = z2 * z2;
    const double R3 = Lp4 + z * Lp5, z6 = z4 * z2;
    const double R4 = Lp6 + z * Lp7;
    const double R = R1 + z2 * R2 + z4 * R3 + z6 * R4;
#else
    const double R = z * (Lp1 + z * (Lp2 + z * (Lp3 + z * (Lp4 + z * (Lp5 + z * (Lp6 + z * Lp7))))));
#endif
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + (dk * ln2_lo + cc))) - f);
}

}  // namespace pmx
