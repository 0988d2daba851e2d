// Incremental DFS index producer (see index_build.hpp).
//
// Inverted blocks (rsv_4K: 93 inverted block insertions) are handled by walking GENOME-ORDER coordinates (see
// Builder::mirror): a strand flip dirties its whole block, whose k-mers are then re-seeded in the new orientation
// (src/index_single_mode.cpp:28-427 does the same with its strand-flip recompute).  Pinned by the reference's own test
// contract (src/test/test_index.cpp:80-110, 145-264: seeds reconstructed from the index deltas == seeds extracted from
// the genome) on rsv_4K and on synthetic pure inversions, tests/test_host_stage.py.
// Rule implemented (validated end-to-end against the reference's golden placement TSV, SURVEY.md
// Appendix E-4/E-7):
//   * apply the node's mutations to the column array, recording each mutated column range;
//   * the *dirty* k-mer starts are the starts of every k-mer overlapping a recorded range;
//   * a dirty start inside the node's own hard flank mask [col of base #flank, col of base #n-flank+1]
//     takes the from-scratch syncmer verdict for this node's genome; everything else (dirty starts
//     outside the mask, all non-dirty columns) inherits the parent's state
//     (src/index_single_mode.cpp:1768-1780, 1850-1854, 1883-1914);
//   * the node's seed multiset is the k-min-mers of the syncmers in column order; the stored changes
//     are the multiset difference to the parent, sorted by hash (:2105-2162, :2530-2534).
// The syncmer map and the k-min-mer counts are updated incrementally: only windows of l consecutive
// syncmers whose column span contains a changed column are removed / re-added.
#include "index_build.hpp"
#include "device/pmx_options.hpp"

#include <algorithm>
#include <atomic>
#include <map>
#include <thread>
#include <stdexcept>
#include <unordered_map>

namespace pmx {
namespace {

struct SynChange {
    uint32_t col;
    bool had;
    uint64_t old_hash;
};

struct NodeUndo {
    UndoLog cols;
    std::vector<SynChange> syn;
    std::vector<std::pair<uint64_t, int32_t>> counts;  // (hash, delta applied)
};

struct Pending {
    bool present;
    uint64_t hash;
};

struct Builder {
    const Panman& pm;
    SyncmerParams p;
    int flank;
    PanmanState st;
    std::map<uint32_t, uint64_t> syn;                // column of the k-mer's first base -> syncmer hash
    std::unordered_map<uint64_t, int32_t> counts;    // running k-min-mer multiset
    // scratch
    std::vector<uint32_t> lc;
    std::string lseq;
    std::vector<uint8_t> is_sync;
    std::vector<uint64_t> shash;

    Builder(const Panman& pm_, const SyncmerParams& p_, int flank_) : pm(pm_), p(p_), flank(flank_) { st.init(pm); }

    // GENOME-ORDER coordinates.  The genome of a node is its existing blocks in id order, an inverted block contributing
    // the reverse complement of its columns (src/panmap_utils.cpp:134-180).  Everything below walks positions x in
    // [0, n_cols) in genome order: x is the column itself in a forward block and the column mirrored inside its block
    // in an inverted one (an involution: the same formula maps back), and the base at x is complemented there.  With no
    // inverted block x == column throughout.
    inline uint32_t mirror(uint32_t x) const {
        const uint32_t b = pm.col_block[x];
        return st.block_fwd[b] ? x : pm.block_col0[b] + pm.block_col0[b + 1] - 1 - x;
    }
    inline char base_at(uint32_t x) const {
        const uint32_t b = pm.col_block[x];
        const char ch = st.cols[st.block_fwd[b] ? x : pm.block_col0[b] + pm.block_col0[b + 1] - 1 - x];
        return st.block_fwd[b] || ch == '-' || ch == 'x' ? ch : complement_iupac(ch);
    }
    inline bool is_base(uint32_t x) const {
        const char ch = st.cols[mirror(x)];
        return ch != '-' && ch != 'x' && st.block_exists[pm.col_block[x]];
    }
    // smallest base position >= x, or n_cols
    uint32_t next_base(uint32_t x) const {
        const uint32_t n = pm.n_cols;
        while (x < n) {
            uint32_t b = pm.col_block[x];
            if (!st.block_exists[b]) { x = pm.block_col0[b + 1]; continue; }
            char ch = st.cols[mirror(x)];
            if (ch != '-' && ch != 'x') return x;
            ++x;
        }
        return n;
    }
    // largest base position <= x, or -1
    int64_t prev_base(int64_t x) const {
        while (x >= 0) {
            uint32_t b = pm.col_block[x];
            if (!st.block_exists[b]) { x = (int64_t)pm.block_col0[b] - 1; continue; }
            char ch = st.cols[mirror((uint32_t)x)];
            if (ch != '-' && ch != 'x') return x;
            --x;
        }
        return -1;
    }

    void enum_windows(const std::vector<uint32_t>& changed, std::vector<uint64_t>& seeds) const {
        const int l = p.l < 1 ? 1 : p.l;
        int64_t last_start = -1;
        uint64_t h[32];
        for (uint32_t c : changed) {
            auto it = syn.lower_bound(c);
            const bool c_in = it != syn.end() && it->first == c;
            auto first = it;
            for (int back = 0; back < l - 1 && first != syn.begin(); ++back) --first;
            for (auto s = first; s != syn.end(); ++s) {
                if (c_in ? s->first > c : s->first >= c) break;
                if ((int64_t)s->first <= last_start) continue;
                auto q = s;
                int cnt = 0;
                uint32_t lastcol = 0;
                for (; cnt < l && q != syn.end(); ++q, ++cnt) { h[cnt] = q->second; lastcol = q->first; }
                if (cnt < l) break;
                if (lastcol < c) continue;
                last_start = s->first;
                uint64_t seed;
                if (kminmer_seed(h, p.k, l, &seed, p.oriented)) seeds.push_back(seed);
            }
        }
    }

    void process(int32_t ni, NodeUndo& u, std::vector<uint64_t>& out_hash, std::vector<int16_t>& out_pc,
                 std::vector<int16_t>& out_cc) {
        std::vector<ColRange> ranges;
        apply_node(pm, ni, st, &u.cols, &ranges);
        if (ranges.empty()) return;
        for (ColRange& r : ranges) {   // column ranges (each inside one block) -> genome-order ranges
            const uint32_t xa = mirror(r.a), xb = mirror(r.b);
            r.a = std::min(xa, xb);
            r.b = std::max(xa, xb);
        }
        std::sort(ranges.begin(), ranges.end(), [](const ColRange& x, const ColRange& y) { return x.a < y.a; });
        std::vector<ColRange> merged;
        for (const ColRange& r : ranges) {
            if (!merged.empty() && (uint64_t)r.a <= (uint64_t)merged.back().b + 1) merged.back().b = std::max(merged.back().b, r.b);
            else merged.push_back(r);
        }
        // hard flank mask of THIS node's genome (src/panmap_utils.hpp:893-970)
        int64_t hms = 0, hme = (int64_t)pm.n_cols - 1;
        bool all_masked = false;
        if (flank > 0) {
            uint32_t c = 0;
            int got = 0;
            hms = -1;
            while ((c = next_base(c)) < pm.n_cols) { if (++got == flank) { hms = c; break; } ++c; }
            int64_t d = (int64_t)pm.n_cols - 1;
            got = 0;
            hme = -1;
            while ((d = prev_base(d)) >= 0) { if (++got == flank) { hme = d; break; } --d; }
            if (hms < 0 || hme < 0 || hms > hme) all_masked = true;
        }
        auto masked = [&](uint32_t c) { return all_masked || (int64_t)c < hms || (int64_t)c > hme; };

        std::map<uint32_t, Pending> pending;
        const int K = p.k;
        for (const ColRange& r : merged) {
            uint32_t L = next_base(r.a);
            // walk K-1 bases to the left of L
            int64_t s0 = L;
            {
                int64_t cur = (int64_t)L;
                for (int i = 0; i < K - 1; ++i) {
                    int64_t pb = prev_base(cur - 1);
                    if (pb < 0) break;
                    cur = pb;
                }
                s0 = cur;
            }
            int64_t R = prev_base(r.b);
            if (R >= 0 && s0 <= R && s0 < (int64_t)pm.n_cols) {
                lc.clear();
                lseq.clear();
                size_t n_starts = 0;
                uint32_t c = (uint32_t)s0;
                while ((c = next_base(c)) < pm.n_cols) {
                    if ((int64_t)c <= R) ++n_starts;
                    else if (lc.size() >= n_starts + (size_t)(K - 1)) break;
                    lc.push_back(c);
                    lseq.push_back(base_at(c));
                    ++c;
                }
                host_syncmers(lseq.data(), (int64_t)lseq.size(), p, is_sync, shash);
                for (size_t j = 0; j < n_starts; ++j) {
                    uint32_t col = lc[j];
                    if (masked(col)) continue;
                    bool is_s = j < is_sync.size() && is_sync[j];
                    pending[col] = Pending{is_s, is_s ? shash[j] : 0};
                }
            }
            // gap columns inside the range that still hold a syncmer
            for (auto it = syn.lower_bound(r.a); it != syn.end() && it->first <= r.b; ++it) {
                uint32_t col = it->first;
                if (!is_base(col) && !masked(col)) pending[col] = Pending{false, 0};
            }
        }
        // keep real changes only
        std::vector<uint32_t> changed;
        for (auto it = pending.begin(); it != pending.end();) {
            auto cur = syn.find(it->first);
            bool had = cur != syn.end();
            if ((had && it->second.present && cur->second == it->second.hash) || (!had && !it->second.present)) it = pending.erase(it);
            else { changed.push_back(it->first); ++it; }
        }
        if (changed.empty()) return;

        std::vector<uint64_t> old_seeds, new_seeds;
        enum_windows(changed, old_seeds);
        for (const auto& kv : pending) {
            auto cur = syn.find(kv.first);
            if (cur != syn.end()) {
                u.syn.push_back(SynChange{kv.first, true, cur->second});
                if (kv.second.present) cur->second = kv.second.hash;
                else syn.erase(cur);
            } else {
                u.syn.push_back(SynChange{kv.first, false, 0});
                syn.emplace(kv.first, kv.second.hash);
            }
        }
        enum_windows(changed, new_seeds);

        std::unordered_map<uint64_t, int32_t> delta;
        for (uint64_t h : old_seeds) --delta[h];
        for (uint64_t h : new_seeds) ++delta[h];
        std::vector<std::pair<uint64_t, std::pair<int32_t, int32_t>>> changes;
        for (const auto& kv : delta) {
            if (kv.second == 0) continue;
            auto it = counts.find(kv.first);
            int32_t pc = it == counts.end() ? 0 : it->second;
            int32_t cc = pc + kv.second;
            if (cc < 0) throw std::runtime_error("index build: negative seed count (internal error)");
            if (cc > INT16_MAX || pc > INT16_MAX) throw std::runtime_error("index build: seed count exceeds int16");
            if (cc == 0) counts.erase(it);
            else if (it == counts.end()) counts.emplace(kv.first, cc);
            else it->second = cc;
            u.counts.push_back({kv.first, kv.second});
            changes.push_back({kv.first, {pc, cc}});
        }
        std::sort(changes.begin(), changes.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
        for (const auto& c : changes) {
            out_hash.push_back(c.first);
            out_pc.push_back((int16_t)c.second.first);
            out_cc.push_back((int16_t)c.second.second);
        }
    }

    void undo(const NodeUndo& u) {
        for (size_t i = u.counts.size(); i-- > 0;) {
            auto it = counts.find(u.counts[i].first);
            int32_t v = (it == counts.end() ? 0 : it->second) - u.counts[i].second;
            if (v == 0) { if (it != counts.end()) counts.erase(it); }
            else if (it == counts.end()) counts.emplace(u.counts[i].first, v);
            else it->second = v;
        }
        for (size_t i = u.syn.size(); i-- > 0;) {
            const SynChange& c = u.syn[i];
            if (c.had) syn[c.col] = c.old_hash;
            else syn.erase(c.col);
        }
        undo_node(st, u.cols);
    }
};

}  // namespace

namespace {
bool has_inverted_blocks(const Panman& pm) {
    for (const auto& nd : pm.nodes)
        for (const auto& bm : nd.block_muts)
            if (bm.inversion) return true;
    return false;
}

// From-scratch producer: every node's genome is materialised (incremental column state, undone on the way
// back up), its seed multiset extracted as the reference's test helpers do (src/test/helpers/
// seed_helpers.cpp: extractSeeds + k-min-mers) and diffed against the parent's.  O(genome) per node instead
// of O(mutated columns); used for PanMANs with inverted blocks, which the incremental builder does not track,
// and by tests/test_host_stage.py as an independent check of the incremental builder.
void build_from_scratch(const Panman& pm, const SyncmerParams& p, int flank_mask, size_t max_nodes, LiteIndex& out) {
    typedef std::vector<std::pair<uint64_t, int32_t>> Counts;
    const size_t n = std::min(pm.nodes.size(), max_nodes);
    PanmanState st;
    st.init(pm);
    std::vector<int32_t> stack;
    std::vector<UndoLog> undos;
    std::vector<Counts> counts;
    for (size_t i = 0; i < n; ++i) {
        const int32_t par = pm.nodes[i].parent;
        while (!stack.empty() && stack.back() != par) {
            undo_node(st, undos.back());
            undos.pop_back();
            counts.pop_back();
            stack.pop_back();
        }
        undos.emplace_back();
        stack.push_back((int32_t)i);
        apply_node(pm, (int32_t)i, st, &undos.back(), nullptr);
        counts.emplace_back();
        genome_seed_counts(genome_of_state(pm, st), p, flank_mask, counts.back());
        static const Counts empty;
        const Counts& pc = counts.size() >= 2 ? counts[counts.size() - 2] : empty;
        const Counts& cc = counts.back();
        size_t a = 0, b = 0;
        auto emit = [&](uint64_t h, int32_t x, int32_t y) {
            if (x == y) return;
            if (x > INT16_MAX || y > INT16_MAX) throw std::runtime_error("index build: seed count exceeds int16");
            out.hash.push_back(h);
            out.parent_count.push_back((int16_t)x);
            out.child_count.push_back((int16_t)y);
        };
        while (a < pc.size() || b < cc.size()) {
            if (b == cc.size() || (a < pc.size() && pc[a].first < cc[b].first)) { emit(pc[a].first, pc[a].second, 0); ++a; }
            else if (a == pc.size() || cc[b].first < pc[a].first) { emit(cc[b].first, 0, cc[b].second); ++b; }
            else { emit(pc[a].first, pc[a].second, cc[b].second); ++a; ++b; }
        }
        out.offsets[i + 1] = out.hash.size();
    }
    for (size_t i = n; i < pm.nodes.size(); ++i) out.offsets[i + 1] = out.hash.size();
}
}  // namespace

void build_lite_index(const Panman& pm, const SyncmerParams& p, int flank_mask, LiteIndex& out, int mode, size_t max_nodes) {
    if (p.l > 32) throw std::runtime_error("index build: l > 32 unsupported");
    out = LiteIndex();
    out.params = p;
    out.flank_mask = flank_mask;
    const size_t n = pm.nodes.size();
    out.node_id.resize(n);
    out.parent.resize(n);
    out.offsets.assign(n + 1, 0);
    for (size_t i = 0; i < n; ++i) {
        out.node_id[i] = pm.nodes[i].id;
        out.parent[i] = pm.nodes[i].parent < 0 ? 0u : (uint32_t)pm.nodes[i].parent;
    }
    if (n == 0) return;
    (void)has_inverted_blocks;
    if (mode == 1) {
        build_from_scratch(pm, p, flank_mask, max_nodes, out);
        return;
    }
    const size_t n_do = std::min(n, max_nodes);
    // Pre-order numbering == node index, so visiting nodes in index order with an explicit ancestor stack is the DFS; undo
    // when leaving a subtree.  `b` must be in the state of node `first`'s parent path: `stack` / `undos` hold that path.
    auto run_range = [&](Builder& b, std::vector<int32_t>& stack, std::vector<NodeUndo>& undos, size_t first, size_t last, std::vector<uint64_t>& o_hash,
                         std::vector<int16_t>& o_pc, std::vector<int16_t>& o_cc, std::vector<uint64_t>& o_end) {
        for (size_t i = first; i < last; ++i) {
            const int32_t par = pm.nodes[i].parent;
            while (!stack.empty() && stack.back() != par) {
                b.undo(undos.back());
                undos.pop_back();
                stack.pop_back();
            }
            undos.emplace_back();
            stack.push_back((int32_t)i);
            b.process((int32_t)i, undos.back(), o_hash, o_pc, o_cc);
            o_end.push_back(o_hash.size());
        }
    };
    unsigned n_thr = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char* e = pmx::opt_str(pmx::O_INDEX_THREADS)) n_thr = (unsigned)std::max(1, atoi(e));
    if (n_thr <= 1 || n_do < 4096) {
        Builder b(pm, p, flank_mask);
        std::vector<int32_t> stack;
        std::vector<NodeUndo> undos;
        std::vector<uint64_t> ends;
        run_range(b, stack, undos, 0, n_do, out.hash, out.parent_count, out.child_count, ends);
        for (size_t i = 0; i < n_do; ++i) out.offsets[i + 1] = ends[i];
    } else {
        // Parallel producer (the reference splits the DFS the same way, src/index_single_mode.cpp:2291-2470): the nodes are
        // cut into contiguous chunks of the pre-order; a worker starts from a copy of the state after the root (processed
        // once: it builds every seed of the tree's first genome), replays the path from the root down to its chunk's first
        // node -- the state of a node depends on its root path alone -- and then runs the serial procedure over the chunk.
        // The chunks' outputs are concatenated in order: the same arrays as the serial build.
        Builder root_b(pm, p, flank_mask);
        std::vector<int32_t> root_stack;
        std::vector<NodeUndo> root_undos;
        std::vector<uint64_t> root_hash, root_ends;
        std::vector<int16_t> root_pc, root_cc;
        run_range(root_b, root_stack, root_undos, 0, 1, root_hash, root_pc, root_cc, root_ends);
        const size_t n_chunks = std::min<size_t>((size_t)n_thr * 6, std::max<size_t>(1, (n_do - 1) / 512));
        struct Chunk { size_t first, last; std::vector<uint64_t> hash, ends; std::vector<int16_t> pc, cc; std::string err; };
        std::vector<Chunk> chunks(n_chunks);
        for (size_t c = 0; c < n_chunks; ++c) { chunks[c].first = 1 + (n_do - 1) * c / n_chunks; chunks[c].last = 1 + (n_do - 1) * (c + 1) / n_chunks; }
        std::atomic<size_t> next{0};
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < std::min<size_t>(n_thr, n_chunks); ++t)
            pool.emplace_back([&]() {
                for (;;) {
                    const size_t c = next.fetch_add(1);
                    if (c >= n_chunks) break;
                    Chunk& ck = chunks[c];
                    if (ck.first >= ck.last) continue;
                    try {
                        Builder b(root_b);                                   // the state after the root
                        std::vector<int32_t> stack(1, 0);
                        std::vector<NodeUndo> undos(1);                      // (the root is never undone inside a chunk)
                        std::vector<int32_t> path;                           // ancestors of the first node below the root, top down
                        for (int32_t v = pm.nodes[ck.first].parent; v > 0; v = pm.nodes[(size_t)v].parent) path.push_back(v);
                        std::vector<uint64_t> sink_h, sink_e;
                        std::vector<int16_t> sink_p, sink_c;
                        for (size_t k = path.size(); k-- > 0;) {
                            undos.emplace_back();
                            stack.push_back(path[k]);
                            b.process(path[k], undos.back(), sink_h, sink_p, sink_c);
                        }
                        run_range(b, stack, undos, ck.first, ck.last, ck.hash, ck.pc, ck.cc, ck.ends);
                    } catch (const std::exception& e) { ck.err = e.what(); if (ck.err.empty()) ck.err = "index build failed"; }
                }
            });
        for (auto& th : pool) th.join();
        for (const Chunk& ck : chunks)
            if (!ck.err.empty()) throw std::runtime_error(ck.err);
        out.hash = std::move(root_hash); out.parent_count = std::move(root_pc); out.child_count = std::move(root_cc);
        out.offsets[1] = out.hash.size();
        for (const Chunk& ck : chunks) {
            const uint64_t base = out.hash.size();
            out.hash.insert(out.hash.end(), ck.hash.begin(), ck.hash.end());
            out.parent_count.insert(out.parent_count.end(), ck.pc.begin(), ck.pc.end());
            out.child_count.insert(out.child_count.end(), ck.cc.begin(), ck.cc.end());
            for (size_t i = ck.first; i < ck.last; ++i) out.offsets[i + 1] = base + ck.ends[i - ck.first];
        }
    }
    for (size_t i = n_do; i < n; ++i) out.offsets[i + 1] = out.hash.size();
}

void genome_seed_counts(const std::string& genome, const SyncmerParams& p, int flank_mask,
                        std::vector<std::pair<uint64_t, int32_t>>& sorted_counts) {
    std::vector<uint8_t> is_sync;
    std::vector<uint64_t> sh;
    host_syncmers(genome.data(), (int64_t)genome.size(), p, is_sync, sh);
    std::vector<uint64_t> h;
    const int64_t n = (int64_t)genome.size();
    for (int64_t i = 0; i < (int64_t)is_sync.size(); ++i) {
        if (!is_sync[i]) continue;
        if (flank_mask > 0 && (i < flank_mask - 1 || i > n - flank_mask)) continue;
        h.push_back(sh[i]);
    }
    std::unordered_map<uint64_t, int32_t> cnt;
    const int l = p.l < 1 ? 1 : p.l;
    for (size_t j = 0; j + l <= h.size(); ++j) {
        uint64_t seed;
        if (kminmer_seed(&h[j], p.k, l, &seed, p.oriented)) ++cnt[seed];
    }
    sorted_counts.assign(cnt.begin(), cnt.end());
    std::sort(sorted_counts.begin(), sorted_counts.end());
}

}  // namespace pmx
