// ALIGN stage, tier-1 kernel for gfx950: the compact layout keeps every work array except the traceback
// matrix in LDS, and this translation unit tells the compiler so (PMX_ALL_LDS -> PMX_LDS() assumes), so
// the per-pair bookkeeping compiles to ds_read/ds_write instead of flat_load/flat_store (the flat path
// was the measured bottleneck: ~9k flat memory instructions per pair, 56% of wave cycles waiting).
#define PMX_ALL_LDS 1
#include "align_kernel_body.hpp"

namespace pmx {
namespace aln {

__global__ void __launch_bounds__(64, 2) k_align_reads_t1(AlignArgs A) { align_reads_body<2>(A); }
__global__ void __launch_bounds__(64, 4) k_align_reads_t1_w4(AlignArgs A) { align_reads_body<4>(A); }

}  // namespace aln
}  // namespace pmx
