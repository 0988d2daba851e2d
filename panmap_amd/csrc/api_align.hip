// C ABI, device part 2: ALIGN stage (placeholder until the alignment kernels land).
#include "device/dev_util.hpp"

extern "C" {
void pmx_align_reads_direct(const char*, const char*, int, const char**, const char**, const char**, const int*, align_pair_result_t*, bool, int) {
    pmx::set_error("align stage not implemented yet");
}
int pmx_aligner_create(pmx_ctx*, const char*, int64_t, int, pmx_aligner**) { pmx::set_error("align stage not implemented yet"); return PMX_ERR_UNSUPPORTED; }
void pmx_aligner_free(pmx_ctx*, pmx_aligner*) {}
int pmx_align_readset(pmx_ctx*, pmx_aligner*, const pmx_readset*, int, int) { return PMX_ERR_UNSUPPORTED; }
int64_t pmx_align_num_records(const pmx_aligner*) { return 0; }
int64_t pmx_align_cigar_words(pmx_ctx*, pmx_aligner*) { return 0; }
int pmx_align_fetch(pmx_ctx*, pmx_aligner*, pmx_aln_record*, int64_t, uint32_t*, int64_t) { return PMX_ERR_UNSUPPORTED; }
const void* pmx_align_device_records(const pmx_aligner*) { return nullptr; }
const void* pmx_align_device_cigars(const pmx_aligner*) { return nullptr; }
}
