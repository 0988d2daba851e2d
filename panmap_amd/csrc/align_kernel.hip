// ALIGN stage, tier-2 (general) kernels for gfx950: work arrays may live in LDS or in the per-wave HBM
// slab (generic pointers).  Used for pairs that exceed the tier-1 capacities and for long reads.
// Integer hashing + int8 DP + a little fp32: VALU/LDS work, no MFMA.
#include "align_kernel_body.hpp"

namespace pmx {
namespace aln {

// two register budgets of the same body: 256 VGPRs (2 waves/SIMD) and 128 VGPRs (4 waves/SIMD)
__global__ void __launch_bounds__(64, 2) k_align_reads(AlignArgs A) { align_reads_body<2>(A); }
__global__ void __launch_bounds__(64, 4) k_align_reads_w4(AlignArgs A) { align_reads_body<4>(A); }

}  // namespace aln
}  // namespace pmx
