// Shared internals of the C ABI translation units.
#pragma once
#include <string>

#include "../../include/panmap_amd.h"

namespace pmx {
void set_error(const std::string& s);
struct LiteIndex;
}  // namespace pmx

// host-side accessors used by the device TU
const pmx::LiteIndex* pmx_index_internal(const pmx_index* idx);
