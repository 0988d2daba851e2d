// Launch interface of the ALIGN kernel (align_kernel.hip), shared with api_align.hip.
#pragma once
#include "align/aln_host.hpp"

#define PMX_ALIGN_WORK_BYTES 1280   // LDS bytes reserved for the Work descriptor

namespace pmx {
namespace aln {

struct AlignArgs {
    // packed reads
    const uint64_t* words;
    const uint32_t* amb;
    const int64_t* woff;
    const int64_t* off;
    int64_t n_items;        // pairs (paired) or reads, or worklist length
    const uint32_t* worklist;   // tier 2: item ids to (re)process; NULL = 0..n_items-1
    uint32_t* retry_list;       // tier 1: items whose capacities overflowed (NULL in tier 2)
    unsigned long long* retry_count;
    unsigned long long* prof;   // 16 phase-cycle accumulators (diagnostic; NULL = off)
    int paired;
    int revcomp_mate2;
    // reference + options
    Opt opt;
    RefIndex ri;
    Layout layout;
    uint8_t* slow_base;
    size_t slow_stride;
    // outputs
    AlnRecord* records;
    uint32_t* cigars;
    uint64_t cigar_cap;
    unsigned long long* cigar_used;
};

__global__ void k_align_reads(AlignArgs A);
__global__ void k_align_reads_w4(AlignArgs A);
__global__ void k_align_reads_t1(AlignArgs A);
__global__ void k_align_reads_t1_w4(AlignArgs A);
__global__ void k_align_reads_tpp(AlignArgs A);

}  // namespace aln
}  // namespace pmx
