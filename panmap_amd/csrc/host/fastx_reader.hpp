// Host FASTA/FASTQ(.gz) ingest with the reference's conventions (src/seeding.cpp:231-269 readFastqPaired,
// src/placement.cpp:164-197 extractReadSequences; both read through kseq.h).  One pass over the inflated stream,
// flat storage (concatenated bases / qualities / names + offsets) that uploads to the device as is.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace pmx {

struct FastxReads {
    std::vector<char> seq, qual, names;          // concatenated; names NUL-free, one per read
    std::vector<int64_t> off, name_off;          // n + 1 offsets each
    int64_t n() const { return (int64_t)off.size() - 1; }
};

// kseq_read semantics: name = header up to the first white space, multi-line sequence and quality, CR stripped, a
// record whose quality length differs from its sequence length ends the file (kseq returns -2 there).
// quals: FASTA records get an empty quality.  Throws std::runtime_error when the file cannot be opened.
void read_fastx(const std::string& path, FastxReads& out);

// readFastqPaired: R2 reverse-complemented (upper-case ACGT only) with its qualities reversed, mates interleaved
// (r1_0, r2_0, r1_1, ...), missing qualities -> 'I' x length.  path2 empty: single-end.  Throws on a mate-count mismatch.
void read_fastq_paired(const std::string& path1, const std::string& path2, FastxReads& out);

}  // namespace pmx
