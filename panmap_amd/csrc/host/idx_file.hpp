// `.idx` single-sample index container: what the reference's place stage loads
// (src/placement.cpp:1009-1092, src/main.cpp:207-214) and its builder writes (src/index_single_mode.cpp:1561-1640).
//   32-byte header  "PMI1", version 1, k, s, t, l (u32 each), hpc, open, uncompressed (bytes 24..26)
//   payload         one Cap'n Proto flat-array message (schema src/index_lite.capnp:36-70, formatVersion 4), raw when
//                   `uncompressed`, else a concatenation of independent zstd frames of 64 MiB input each, checksum on
//                   (src/zstd_compression.cpp:14-100).
// zstd is resolved at run time from the system's libzstd.so.1 (the image ships the runtime, not the headers).
#pragma once
#include <string>

#include "index_build.hpp"

namespace pmx {

struct IdxHeader {
    int32_t k = 0, s = 0, t = 0, l = 0;
    bool hpc = false, open = false, uncompressed = false;
};
constexpr uint16_t kIdxFormatVersion = 4;   // panmapUtils::INDEX_FORMAT_VERSION (src/panmap_utils.hpp:27)

// false: the file is absent or does not start with this header (src/index_single_mode.cpp:1574-1590)
bool read_idx_header(const std::string& path, IdxHeader& out);
// throws std::runtime_error (messages follow src/placement.cpp:1013-1047 where the reference has one)
void load_idx(const std::string& path, LiteIndex& out);
void save_idx(const LiteIndex& ix, const std::string& path, int zstd_level, bool uncompressed);

}  // namespace pmx
