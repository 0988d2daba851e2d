// BAM + BAI egress of alignment results (host).  See bam_writer.cpp.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "panmap_amd.h"

namespace pmx {

// Writes `bam_path` (coordinate-sorted BAM of the mapped fragments) and, if write_index, `bam_path`.bai, from
// the results of align_reads_direct / pmx_align_reads_direct.  seqs / quals / names are the arrays that were
// handed to the aligner (R2 already reverse-complemented and its qualities reversed, src/seeding.cpp:231-269).
// Returns 0, or 1 when a file could not be written (like alignAndWriteBam); throws on I/O setup errors.
int write_bam(const std::string& bam_path, const std::string& ref_name, int64_t ref_len, const std::vector<std::string>& seqs,
              const std::vector<std::string>& quals, const std::vector<std::string>& names, const align_pair_result_t* results, int64_t n_results,
              bool paired, bool write_index);

}  // namespace pmx
