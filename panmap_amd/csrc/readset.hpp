// Device-resident read set shared by the place and align translation units.
#pragma once
#include <utility>
#include <vector>

#include "device/dev_util.hpp"

namespace pmx {
// a set of disjoint half-open read ranges (streaming: which reads of a read set have been packed / ordered so far)
struct RangeSet {
    std::vector<std::pair<int64_t, int64_t>> iv;   // sorted, merged
    void clear() { iv.clear(); }
    void add(int64_t a, int64_t b) {
        if (b <= a) return;
        std::vector<std::pair<int64_t, int64_t>> out;
        for (const auto& r : iv) {
            if (r.second < a || r.first > b) out.push_back(r);
            else { a = a < r.first ? a : r.first; b = b > r.second ? b : r.second; }
        }
        out.emplace_back(a, b);
        for (size_t i = out.size() - 1; i > 0 && out[i].first < out[i - 1].first; --i) std::swap(out[i], out[i - 1]);
        iv.swap(out);
    }
    bool covers(int64_t a, int64_t b) const {
        if (b <= a) return true;
        for (const auto& r : iv)
            if (r.first <= a && b <= r.second) return true;
        return false;
    }
};
}  // namespace pmx

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
    // Read RECORDS (reads of up to 160 bases): per read one aligned 64-byte line = its five 2-bit words, five ambiguity
    // words and its length, written by the pack kernels next to the arrays above.  The kernels that visit reads in a
    // scattered order (k_collapse_reads, the compact align tier) fetch ONE line per read instead of pieces of two arrays
    // plus two offset look-ups (k_collapse_reads: 470 B fetched per read before, PMC)
    pmx::DevBuf<uint8_t> recs;
    bool has_recs = false;
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
    // pair order of the align stage (pairs by both mates' locality keys: api_align.hip), made ahead of time on a side stream
    // by pmx_readset_order_pairs -- it depends on the reads alone, so it can run beside the place stage's scoring instead of
    // between the placement and the first align kernel -- or by the aligner itself when nobody asked
    mutable pmx::DevBuf<uint64_t> pp_key, pp_key2;
    mutable pmx::DevBuf<uint32_t> pp_idx, pp_idx2;
    mutable pmx::DevBuf<char> pp_tmp;
    mutable bool has_pair_order = false;
    mutable hipEvent_t pair_ev = nullptr;     // recorded behind the sort when it ran on a side stream
    mutable bool pair_ev_pending = false;
    ~pmx_readset() { if (pair_ev) (void)hipEventDestroy(pair_ev); }
    // streaming (pmx_readset_pack_range / pmx_place_add_reads_range): the ranges packed so far (`packed` once they cover the
    // set) and the ranges whose slice of loc_perm holds their reads in locality order (`has_order` once they cover the set:
    // a permutation sorted range by range serves the align stage as well as one sorted as a whole)
    pmx::RangeSet packed_ranges;
    mutable pmx::RangeSet ordered_ranges;
};

struct pmx_ctx;
namespace pmx {
// reads sorted by locality key (stable), as a permutation of 0..n-1 on the device; enqueued on the context's stream the
// first time it is asked for after a (re)packing.  nullptr for read sets too small or too large for it.
const uint32_t* readset_locality_order(pmx_ctx* ctx, const pmx_readset* rs);
// the same for the reads [r0, r1) alone: their slice of the permutation (absolute read indices), sorted by locality key
const uint32_t* readset_locality_order_range(pmx_ctx* ctx, const pmx_readset* rs, int64_t r0, int64_t r1);
const uint32_t* readset_pair_order(pmx_ctx* ctx, const pmx_readset* rs, hipStream_t side);   // api_align.hip
}
