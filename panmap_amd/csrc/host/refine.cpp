#include "refine.hpp"

#include <algorithm>
#include <deque>
#include <map>
#include <set>
#include <utility>

namespace pmx {

std::vector<uint32_t> nodes_within_radius(const std::vector<uint32_t>& parent, const std::vector<std::vector<uint32_t>>& children, uint32_t start, int radius,
                                          int max_nodes) {
    std::vector<uint32_t> out;
    if (radius <= 0 || max_nodes <= 0) return out;
    std::set<uint32_t> visited{start};
    std::deque<std::pair<uint32_t, int>> queue{{start, 0}};
    while (!queue.empty() && (int)out.size() < max_nodes) {
        const uint32_t node = queue.front().first;
        const int dist = queue.front().second;
        queue.pop_front();
        if (node != start) out.push_back(node);
        if (dist >= radius) continue;
        if (node != 0 && visited.insert(parent[node]).second) queue.push_back({parent[node], dist + 1});
        for (uint32_t c : children[node])
            if (visited.insert(c).second) queue.push_back({c, dist + 1});
    }
    return out;
}

RefineResult refine_top_candidates(const uint32_t* parent_in, int64_t n_nodes, const double* scores5, const uint32_t best[5], const RefineParams& rp,
                                   const std::function<bool(uint32_t, int64_t*)>& score_node) {
    RefineResult res;
    if (n_nodes <= 0) return res;
    const std::vector<uint32_t> parent(parent_in, parent_in + n_nodes);
    std::vector<std::vector<uint32_t>> children((size_t)n_nodes);
    for (int64_t i = 1; i < n_nodes; ++i) children[parent[(size_t)i]].push_back((uint32_t)i);
    // the reference keeps the per-node scores as floats (PlacementResult::nodeScores, src/placement.hpp:181)
    auto seed_score = [&](uint32_t node, int m) { return (float)scores5[(size_t)node * 5 + m]; };

    std::set<uint32_t> expanded[5], all;
    for (int m = 0; m < 5; ++m) {
        // step 1: the metric's top candidates (src/placement.cpp:532-575)
        std::vector<std::pair<double, uint32_t>> scored;
        for (int64_t i = 0; i < n_nodes; ++i) {
            const double s = (double)seed_score((uint32_t)i, m);
            if (s > 0) scored.push_back({s, (uint32_t)i});
        }
        std::set<uint32_t> base;
        if (!scored.empty()) {
            std::sort(scored.begin(), scored.end(), std::greater<std::pair<double, uint32_t>>());
            size_t n_top = std::min((size_t)((double)scored.size() * rp.top_pct), (size_t)rp.max_top_n);
            n_top = std::max(n_top, (size_t)1);
            for (size_t i = 0; i < n_top && i < scored.size(); ++i) base.insert(scored[i].second);
        }
        if (best[m] != UINT32_MAX && (int64_t)best[m] < n_nodes) base.insert(best[m]);
        // step 2: with their neighbours (src/placement.cpp:577-608)
        for (uint32_t node : base) {
            expanded[m].insert(node);
            all.insert(node);
            for (uint32_t nb : nodes_within_radius(parent, children, node, rp.neighbor_radius, rp.max_neighbor_n)) {
                expanded[m].insert(nb);
                all.insert(nb);
            }
        }
    }
    if (all.empty()) return res;   // "no nodes with positive scores"
    // step 3: every candidate is aligned against once (src/placement.cpp:618-640)
    std::map<uint32_t, int64_t> score_of;
    for (uint32_t node : all) {
        int64_t s = 0;
        if (!score_node(node, &s)) return res;
        score_of[node] = s;
        res.candidates.push_back(node);
        res.candidate_scores.push_back(s);
    }
    // step 4: per metric, the best alignment score within its own set; ties by seed score, then by the lower index
    // (src/placement.cpp:643-690: the scan order of the hash set cannot matter with these tie rules)
    for (int m = 0; m < 5; ++m) {
        bool have = false;
        int64_t bs = 0;
        uint32_t bi = UINT32_MAX;
        for (uint32_t node : expanded[m]) {
            const int64_t s = score_of[node];
            bool take = !have || s > bs;
            if (have && s == bs) {
                const float a = seed_score(node, m), b = seed_score(bi, m);
                take = a > b || (a == b && node < bi);
            }
            if (take) { have = true; bs = s; bi = node; }
        }
        if (have) { res.score[m] = bs; res.node[m] = bi; }
    }
    res.ran = true;
    return res;
}

}  // namespace pmx
