// CPU producer of the single-sample seed index (the reference's `index` stage,
// src/index_single_mode.cpp:1647-2205, 2291-2571) in the SoA layout the place stage consumes
// (src/index_lite.capnp:36-70): nodes in DFS pre-order, per node the
// seed changes (hash, parentCount, childCount) sorted by hash, only entries whose count changed.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "panman.hpp"
#include "seed_host.hpp"

namespace pmx {

struct LiteIndex {
    SyncmerParams params;
    int flank_mask = 250;
    bool hpc = false;
    std::vector<std::string> node_id;   // DFS pre-order
    std::vector<uint32_t> parent;       // parent[0] == 0
    std::vector<uint64_t> offsets;      // n_nodes + 1
    std::vector<uint64_t> hash;
    std::vector<int16_t> parent_count, child_count;
    size_t n_nodes() const { return parent.size(); }
};

// Builds the index over the whole tree (state undone on exit).  mode 0 = automatic: incremental DFS (work
// proportional to the mutated columns of every node), or the from-scratch producer (every node's genome
// re-seeded and diffed against its parent) when the PanMAN has inverted blocks; 1 = from scratch, 2 =
// incremental.  max_nodes < n_nodes stops after that many nodes in DFS order (tests).
// Throws std::runtime_error on count overflow.
void build_lite_index(const Panman& pm, const SyncmerParams& p, int flank_mask, LiteIndex& out, int mode = 0,
                      size_t max_nodes = (size_t)-1);

// From-scratch seed multiset of a genome string with the hard flank mask applied (test helper
// mirroring src/test/helpers/seed_helpers.cpp:12 extractSeeds + k-min-mers).
void genome_seed_counts(const std::string& genome, const SyncmerParams& p, int flank_mask,
                        std::vector<std::pair<uint64_t, int32_t>>& sorted_counts);

}  // namespace pmx
