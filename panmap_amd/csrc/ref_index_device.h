// Device build of the aligner's reference index (ref_index_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "align/aln_types.hpp"
#include "device/dev_util.hpp"

namespace pmx {
namespace aln {

struct RefIndexDevice {
    DevBuf<char> ascii, tmp;
    DevBuf<uint8_t> seq;
    DevBuf<uint64_t> pk, pk_amb, keys, keys2, pos;
    DevBuf<HtEnt> ht;
    DevBuf<uint32_t> ht_pv;
    DevBuf<unsigned long long> ctr;   // [0] minimizers, [1] distinct ones, [2] distinct ones with ten or more occurrences
    uint32_t ht_mask = 0;
    int64_t n_mv = 0, n_keys = 0;
};

// k odd, 2k + 22 <= 64, w <= 12, reference shorter than 2^21 bases
bool ref_index_device_supported(const Opt& o, int64_t ref_len);
// Enqueues the build on `st` (one host round trip inside) and fills d; o.mid_occ / o.ref_len are set as build_ref_index
// sets them.  false: this reference needs the host build (mid_occ not decidable from the counters), nothing of d is valid.
bool build_ref_index_device(hipStream_t st, const char* reference, int64_t ref_len, Opt& o, RefIndexDevice& d);

}  // namespace aln
}  // namespace pmx
