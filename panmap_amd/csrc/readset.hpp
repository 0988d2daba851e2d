// Device-resident read set shared by the place and align translation units.
#pragma once
#include "device/dev_util.hpp"

struct pmx_readset {
    int64_t n = 0;          // reads
    int64_t total = 0;      // bases
    int64_t max_len = 0;
    int64_t n_words = 0;    // 32-base words
    bool packed = false;
    pmx::DevBuf<uint8_t> ascii;    // concatenated ASCII
    pmx::DevBuf<int64_t> off;      // n+1 byte offsets into ascii
    pmx::DevBuf<int64_t> woff;     // n+1 word offsets into words/amb
    pmx::DevBuf<uint64_t> words;   // 2 bits per base
    pmx::DevBuf<uint32_t> amb;     // 1 bit per base: not A/C/G/T
    pmx::DevBuf<uint8_t> qual;     // optional: Phred+33 per base, same offsets as ascii (--min-seed-quality)
    bool has_qual = false;
    pmx::DevBuf<int64_t> nw_tmp;   // rewrap: words per read (scan input)
    pmx::DevBuf<char> scan_tmp;    // rewrap: rocprim temp storage
    pmx::DevBuf<unsigned long long> stats;
    int64_t off0 = 0;              // first offset (non-zero for a wrapped slice of a larger offsets array)
    // locality order of the reads (read_locality_key, device/pmx_math.h): computed once per packing, on first request, and
    // shared by the seeding launch order (place stage) and the pair order of the align stage
    mutable pmx::DevBuf<uint32_t> loc_key, loc_key2, loc_idx, loc_perm;
    mutable pmx::DevBuf<char> loc_tmp;
    mutable bool has_order = false;
};

struct pmx_ctx;
namespace pmx {
// reads sorted by locality key (stable), as a permutation of 0..n-1 on the device; enqueued on the context's stream the
// first time it is asked for after a (re)packing.  nullptr for read sets too small or too large for it.
const uint32_t* readset_locality_order(pmx_ctx* ctx, const pmx_readset* rs);
}
