// --refine: alignment-based re-ranking of the placement's top candidates (refineTopCandidates, src/placement.cpp:516-698).
// Host logic only: which nodes to align against, and which of them each metric ends up with; the alignment score of a
// node is the caller's business (device: pmx_align_score_reads on the node's genome).
#pragma once
#include <cstdint>
#include <functional>
#include <vector>

namespace pmx {

struct RefineParams {       // src/main.cpp:186-190
    double top_pct = 0.01;
    int max_top_n = 150;
    int neighbor_radius = 2;
    int max_neighbor_n = 150;
};

struct RefineResult {
    bool ran = false;
    int64_t score[5] = {0, 0, 0, 0, 0};                  // refined_<metric> score (minus the total edit distance)
    uint32_t node[5] = {UINT32_MAX, UINT32_MAX, UINT32_MAX, UINT32_MAX, UINT32_MAX};
    std::vector<uint32_t> candidates;                    // every node that was aligned against, ascending
    std::vector<int64_t> candidate_scores;
};

// nodes within `radius` branches of `start` (parent first, then the children in DFS order), breadth first, at most
// max_nodes of them, the start node excluded (getNodesWithinRadius, src/placement.cpp:440-475)
std::vector<uint32_t> nodes_within_radius(const std::vector<uint32_t>& parent, const std::vector<std::vector<uint32_t>>& children, uint32_t start, int radius,
                                          int max_nodes);

// parent[i] = DFS index of node i's parent (parent[0] is ignored); scores5[i * 5 + m] = metric m of node i in the order of
// the placement TSV (log_raw, log_cosine, containment, weighted_containment, log_containment); best[m] = the placement's
// winner of metric m or UINT32_MAX.  score_node(node, &score) returns false to abort (the result is then not `ran`).
RefineResult refine_top_candidates(const uint32_t* parent, int64_t n_nodes, const double* scores5, const uint32_t best[5], const RefineParams& rp,
                                   const std::function<bool(uint32_t, int64_t*)>& score_node);

}  // namespace pmx
