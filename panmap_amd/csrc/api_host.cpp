// C ABI, host-only part: PanMAN access, node genomes, seed-index build (see include/panmap_amd.h).
#include <cstring>
#include <memory>
#include <mutex>
#include <string>

#include "api_internal.hpp"
#include "host/bam_writer.hpp"
#include "host/fastx_reader.hpp"
#include "host/idx_file.hpp"
#include "host/refine.hpp"
#include "host/index_build.hpp"
#include "host/panman.hpp"

namespace pmx {
thread_local std::string g_last_error;
void set_error(const std::string& s) { g_last_error = s; }
}  // namespace pmx

struct pmx_panman {
    pmx::Panman pm;
    // the state after the root's mutations, made at the first genome request (pmx_panman_node_genome is on the critical
    // path of a sample: the GPU waits for the placed genome between the place and the align stage)
    mutable std::mutex root_mu;
    mutable std::shared_ptr<const pmx::PanmanState> root;
};

struct pmx_index {
    pmx::LiteIndex ix;
};

extern "C" {

const char* pmx_last_error(void) { return pmx::g_last_error.c_str(); }
const char* pmx_version(void) { return "panmap_amd 0.1 (gfx950)"; }

int pmx_panman_open(const char* path, pmx_panman** out) {
    if (!path || !out) return PMX_ERR_ARG;
    try {
        pmx_panman* p = new pmx_panman();
        pmx::load_panman(path, p->pm);
        *out = p;
        return PMX_OK;
    } catch (const std::exception& e) {
        pmx::set_error(e.what());
        return PMX_ERR_FORMAT;
    }
}
void pmx_panman_close(pmx_panman* pm) { delete pm; }
int64_t pmx_panman_num_nodes(const pmx_panman* pm) { return pm ? (int64_t)pm->pm.nodes.size() : 0; }
int64_t pmx_panman_num_blocks(const pmx_panman* pm) { return pm ? pm->pm.n_blocks : 0; }
int64_t pmx_panman_num_columns(const pmx_panman* pm) { return pm ? pm->pm.n_cols : 0; }
const char* pmx_panman_node_id(const pmx_panman* pm, int64_t i) {
    if (!pm || i < 0 || i >= (int64_t)pm->pm.nodes.size()) return nullptr;
    return pm->pm.nodes[i].id.c_str();
}
int64_t pmx_panman_parent(const pmx_panman* pm, int64_t i) {
    if (!pm || i < 0 || i >= (int64_t)pm->pm.nodes.size()) return -1;
    return pm->pm.nodes[i].parent;
}
int64_t pmx_panman_find_node(const pmx_panman* pm, const char* id) {
    if (!pm || !id) return -1;
    return pm->pm.find_node(id);
}
int64_t pmx_panman_node_genome(const pmx_panman* pm, int64_t i, char* buf, int64_t cap) {
    if (!pm || i < 0 || i >= (int64_t)pm->pm.nodes.size()) return -1;
    try {
        std::shared_ptr<const pmx::PanmanState> root;
        {
            std::lock_guard<std::mutex> lk(pm->root_mu);
            if (!pm->root) pm->root = std::make_shared<const pmx::PanmanState>(pmx::root_state_of(pm->pm));
            root = pm->root;
        }
        std::string g = pmx::node_genome(pm->pm, (int32_t)i, root.get());
        if (buf && cap >= (int64_t)g.size()) std::memcpy(buf, g.data(), g.size());
        return (int64_t)g.size();
    } catch (const std::exception& e) {
        pmx::set_error(e.what());
        return -1;
    }
}

int64_t pmx_panman_test_invert_block(pmx_panman* pm, int64_t i, int min_bases) {
    if (!pm || i < 0 || i >= (int64_t)pm->pm.nodes.size()) return -1;
    try {
        const pmx::Panman& P = pm->pm;
        std::vector<int32_t> path;
        for (int32_t x = (int32_t)i; x >= 0; x = P.nodes[x].parent) path.push_back(x);
        pmx::PanmanState st;
        st.init(P);
        for (auto it = path.rbegin(); it != path.rend(); ++it) pmx::apply_node(P, *it, st, nullptr, nullptr);
        int32_t best = -1;
        int64_t best_len = 0;
        for (int32_t b = 0; b < P.n_blocks; ++b) {
            if (!st.block_exists[b] || !st.block_fwd[b]) continue;
            int64_t bases = 0;
            for (uint32_t c = P.block_col0[b]; c < P.block_col0[b + 1]; ++c)
                bases += st.cols[c] != '-' && st.cols[c] != 'x';
            if (bases >= min_bases && bases > best_len) { best_len = bases; best = b; }
        }
        if (best >= 0) {
            pm->pm.nodes[i].block_muts.push_back(pmx::BlockMut{best, false, true});
            std::lock_guard<std::mutex> lk(pm->root_mu);
            pm->root.reset();   // (the root itself may be the node that changed)
        }
        return best;
    } catch (const std::exception& e) {
        pmx::set_error(e.what());
        return -1;
    }
}

int pmx_index_build(const pmx_panman* pm, int k, int s, int t, int l, int open_syncmer, int flank_mask, pmx_index** out) {
    return pmx_index_build_ex(pm, k, s, t, l, open_syncmer, flank_mask, 0, -1, out);
}

int pmx_index_build_ex(const pmx_panman* pm, int k, int s, int t, int l, int open_syncmer, int flank_mask, int mode, int64_t max_nodes,
                       pmx_index** out) {
    const bool oriented = mode >= 0 && (mode & PMX_INDEX_ORIENTED) != 0;
    if (mode >= 0) mode &= ~PMX_INDEX_ORIENTED;
    if (!pm || !out || mode < 0 || mode > 2) return PMX_ERR_ARG;
    if (oriented && l < 2) { pmx::set_error("an oriented (--meta) index needs l >= 2"); return PMX_ERR_UNSUPPORTED; }
    // same validation as the CLI (src/main.cpp:2221-2235)
    if (k <= 0 || s <= 0 || s > k || t < 0 || t > k - s || l < 0 || k > 64) {
        pmx::set_error("invalid seeding parameters");
        return PMX_ERR_ARG;
    }
    try {
        pmx_index* ix = new pmx_index();
        pmx::SyncmerParams p;
        p.k = k; p.s = s; p.t = t; p.l = l; p.open = open_syncmer != 0; p.oriented = oriented;
        pmx::build_lite_index(pm->pm, p, flank_mask, ix->ix, mode, max_nodes < 0 ? (size_t)-1 : (size_t)max_nodes);
        *out = ix;
        return PMX_OK;
    } catch (const std::exception& e) {
        pmx::set_error(e.what());
        return PMX_ERR_UNSUPPORTED;
    }
}

int pmx_index_from_arrays(const pmx_index_info* info, const uint32_t* parent, const uint64_t* offsets, const uint64_t* hash,
                          const int16_t* pc, const int16_t* cc, pmx_index** out) {
    if (!info || !parent || !offsets || !out || info->n_nodes <= 0) return PMX_ERR_ARG;
    const int64_t n = info->n_nodes, m = (int64_t)offsets[n];
    if (m != info->n_changes || (m > 0 && (!hash || !pc || !cc))) { pmx::set_error("index arrays inconsistent"); return PMX_ERR_ARG; }
    for (int64_t i = 0; i < n; ++i)
        if (offsets[i] > offsets[i + 1] || (i > 0 && parent[i] >= (uint32_t)i)) { pmx::set_error("index arrays: offsets not monotone or parent >= child"); return PMX_ERR_FORMAT; }
    pmx_index* ix = new pmx_index();
    pmx::LiteIndex& L = ix->ix;
    L.params.k = info->k; L.params.s = info->s; L.params.t = info->t; L.params.l = info->l;
    L.params.open = info->open_syncmer != 0;
    L.hpc = info->hpc != 0;
    L.flank_mask = info->flank_mask;
    L.parent.assign(parent, parent + n);
    L.offsets.assign(offsets, offsets + n + 1);
    L.hash.assign(hash, hash + m);
    L.parent_count.assign(pc, pc + m);
    L.child_count.assign(cc, cc + m);
    L.node_id.resize(n);
    *out = ix;
    return PMX_OK;
}

// ---- .idx container (host/idx_file.cpp)
int pmx_index_save(const pmx_index* idx, const char* path, int zstd_level, int uncompressed) {
    if (!idx || !path) return PMX_ERR_ARG;
    try {
        pmx::save_idx(idx->ix, path, zstd_level, uncompressed != 0);
        return PMX_OK;
    } catch (const std::exception& e) {
        pmx::set_error(e.what());
        return PMX_ERR_IO;
    }
}
int pmx_index_load(const char* path, pmx_index** out) {
    if (!path || !out) return PMX_ERR_ARG;
    try {
        std::unique_ptr<pmx_index> ix(new pmx_index());
        pmx::load_idx(path, ix->ix);
        ix->ix.flank_mask = -1;   // not recorded in the file
        *out = ix.release();
        return PMX_OK;
    } catch (const std::exception& e) {
        pmx::set_error(e.what());
        return PMX_ERR_FORMAT;
    }
}
int pmx_index_read_header(const char* path, pmx_index_info* info, int* uncompressed) {
    if (!path || !info) return PMX_ERR_ARG;
    pmx::IdxHeader h;
    if (!pmx::read_idx_header(path, h)) { pmx::set_error(std::string(path) + ": absent or without a PMI1 header"); return PMX_ERR_FORMAT; }
    std::memset(info, 0, sizeof(*info));
    info->k = h.k; info->s = h.s; info->t = h.t; info->l = h.l; info->open_syncmer = h.open; info->hpc = h.hpc; info->flank_mask = -1;
    if (uncompressed) *uncompressed = h.uncompressed ? 1 : 0;
    return PMX_OK;
}
const char* pmx_index_node_id(const pmx_index* idx, int64_t dfs_index) {
    if (!idx || dfs_index < 0 || (size_t)dfs_index >= idx->ix.node_id.size()) return nullptr;
    return idx->ix.node_id[(size_t)dfs_index].c_str();
}

void pmx_index_close(pmx_index* idx) { delete idx; }

int pmx_index_get_info(const pmx_index* idx, pmx_index_info* info) {
    if (!idx || !info) return PMX_ERR_ARG;
    const pmx::LiteIndex& L = idx->ix;
    std::memset(info, 0, sizeof(*info));
    info->k = L.params.k; info->s = L.params.s; info->t = L.params.t; info->l = L.params.l;
    info->open_syncmer = L.params.open; info->hpc = L.hpc; info->flank_mask = L.flank_mask;
    info->n_nodes = (int64_t)L.parent.size();
    info->n_changes = (int64_t)L.hash.size();
    return PMX_OK;
}
const uint32_t* pmx_index_parents(const pmx_index* idx) { return idx ? idx->ix.parent.data() : nullptr; }
const uint64_t* pmx_index_offsets(const pmx_index* idx) { return idx ? idx->ix.offsets.data() : nullptr; }
const uint64_t* pmx_index_hashes(const pmx_index* idx) { return idx ? idx->ix.hash.data() : nullptr; }
const int16_t* pmx_index_parent_counts(const pmx_index* idx) { return idx ? idx->ix.parent_count.data() : nullptr; }
const int16_t* pmx_index_child_counts(const pmx_index* idx) { return idx ? idx->ix.child_count.data() : nullptr; }

}  // extern "C"

const pmx::LiteIndex* pmx_index_internal(const pmx_index* idx) { return idx ? &idx->ix : nullptr; }

extern "C" int pmx_write_bam(const char* bam_path, const char* ref_name, int64_t ref_len, int n_reads, const char** reads, const char** quality,
                             const char** read_names, const int* r_lens, const align_pair_result_t* results, bool pairedEndReads) {
    if (!bam_path || !ref_name || n_reads < 0 || (n_reads > 0 && (!reads || !quality || !read_names || !r_lens || !results))) return PMX_ERR_ARG;
    try {
        std::vector<std::string> seqs((size_t)n_reads), quals((size_t)n_reads), names((size_t)n_reads);
        for (int i = 0; i < n_reads; ++i) {
            seqs[(size_t)i].assign(reads[i], (size_t)r_lens[i]);
            quals[(size_t)i] = quality[i] ? std::string(quality[i]) : std::string();
            if ((int)quals[(size_t)i].size() < r_lens[i]) quals[(size_t)i].resize((size_t)r_lens[i], 'I');   // missing qualities -> 'I' (src/seeding.cpp:231-269)
            names[(size_t)i] = read_names[i] ? read_names[i] : "";
        }
        const int64_t n_results = pairedEndReads ? n_reads / 2 : n_reads;
        const int rc = pmx::write_bam(bam_path, ref_name, ref_len, seqs, quals, names, results, n_results, pairedEndReads, true);
        if (rc != 0) { pmx::set_error("failed to write BAM / BAI"); return PMX_ERR_IO; }
        return PMX_OK;
    } catch (const std::exception& e) {
        pmx::set_error(e.what());
        return PMX_ERR_IO;
    }
}

// ---------------------------------------------------------------------------------- FASTA / FASTQ ingest
struct pmx_fastx {
    pmx::FastxReads r;
};

extern "C" int pmx_fastx_read_paired(const char* path1, const char* path2, pmx_fastx** out) {
    if (!path1 || !out) return PMX_ERR_ARG;
    pmx_fastx* fx = new pmx_fastx();
    try {
        pmx::read_fastq_paired(path1, path2 ? path2 : "", fx->r);
    } catch (const std::exception& e) {
        const std::string what = e.what();
        delete fx;
        pmx::set_error(what);
        return what.rfind("Error:", 0) == 0 ? PMX_ERR_ARG : PMX_ERR_IO;   // mate-count mismatch vs unreadable file
    }
    *out = fx;
    return PMX_OK;
}
extern "C" int pmx_fastx_read(const char* path, pmx_fastx** out) {
    if (!path || !out) return PMX_ERR_ARG;
    pmx_fastx* fx = new pmx_fastx();
    try {
        pmx::read_fastx(path, fx->r);
    } catch (const std::exception& e) {
        pmx::set_error(e.what());
        delete fx;
        return PMX_ERR_IO;
    }
    *out = fx;
    return PMX_OK;
}
extern "C" int64_t pmx_fastx_num_reads(const pmx_fastx* fx) { return fx ? fx->r.n() : -1; }
extern "C" int pmx_fastx_views(const pmx_fastx* fx, const char** seq_concat, const char** qual_concat, const int64_t** offsets,
                               const char** names_concat, const int64_t** name_offsets) {
    if (!fx) return PMX_ERR_ARG;
    if (seq_concat) *seq_concat = fx->r.seq.data();
    if (qual_concat) *qual_concat = fx->r.qual.data();
    if (offsets) *offsets = fx->r.off.data();
    if (names_concat) *names_concat = fx->r.names.data();
    if (name_offsets) *name_offsets = fx->r.name_off.data();
    return PMX_OK;
}
extern "C" void pmx_fastx_free(pmx_fastx* fx) { delete fx; }

extern "C" int pmx_refine_top_candidates(const uint32_t* parent, int64_t n_nodes, const double* scores5, const uint32_t best_index[5],
                                         const pmx_refine_params* rp, pmx_refine_score_fn fn, void* user, pmx_refine_result* out,
                                         uint32_t* cand_nodes, int64_t* cand_scores, int64_t cand_cap) {
    if (!parent || n_nodes <= 0 || !scores5 || !best_index || !rp || !fn || !out) return PMX_ERR_ARG;
    try {
        pmx::RefineParams p;
        p.top_pct = rp->top_pct; p.max_top_n = rp->max_top_n; p.neighbor_radius = rp->neighbor_radius; p.max_neighbor_n = rp->max_neighbor_n;
        int cb_rc = 0;
        const pmx::RefineResult r = pmx::refine_top_candidates(parent, n_nodes, scores5, best_index, p, [&](uint32_t node, int64_t* s) {
            cb_rc = fn(user, node, s);
            return cb_rc == 0;
        });
        if (cb_rc != 0) return cb_rc;
        memset(out, 0, sizeof(*out));
        out->ran = r.ran ? 1 : 0;
        out->n_candidates = (int32_t)r.candidates.size();
        for (int m = 0; m < 5; ++m) { out->score[m] = r.score[m]; out->node[m] = r.node[m]; }
        for (size_t i = 0; i < r.candidates.size() && (int64_t)i < cand_cap; ++i) {
            if (cand_nodes) cand_nodes[i] = r.candidates[i];
            if (cand_scores) cand_scores[i] = r.candidate_scores[i];
        }
        return PMX_OK;
    } catch (const std::exception& e) {
        pmx::set_error(e.what());
        return PMX_ERR_IO;
    }
}

// the candidate set alone (steps 1-2 of refineTopCandidates), ascending: lets a caller score the candidates in any order /
// concurrently before pmx_refine_top_candidates picks the winners from a lookup
extern "C" int64_t pmx_refine_candidates(const uint32_t* parent, int64_t n_nodes, const double* scores5, const uint32_t best_index[5],
                                         const pmx_refine_params* rp, uint32_t* cand_nodes, int64_t cand_cap) {
    if (!parent || n_nodes <= 0 || !scores5 || !best_index || !rp) return PMX_ERR_ARG;
    try {
        pmx::RefineParams p;
        p.top_pct = rp->top_pct; p.max_top_n = rp->max_top_n; p.neighbor_radius = rp->neighbor_radius; p.max_neighbor_n = rp->max_neighbor_n;
        const pmx::RefineResult r = pmx::refine_top_candidates(parent, n_nodes, scores5, best_index, p, [](uint32_t, int64_t* s) { *s = 0; return true; });
        for (size_t i = 0; i < r.candidates.size() && (int64_t)i < cand_cap; ++i)
            if (cand_nodes) cand_nodes[i] = r.candidates[i];
        return (int64_t)r.candidates.size();
    } catch (const std::exception& e) {
        pmx::set_error(e.what());
        return PMX_ERR_IO;
    }
}

// mgsr::getDust (src/mgsr.cpp:1505-1568; contract src/test/test_mgsr.cpp:12-29): Prinseq-scaled DUST score of a read over
// its base triplets.  State: how often each of the 64 triplets occurs among the last `window` triplets (a ring of their
// codes); the score of a window is the number of unordered pairs of equal triplets in it, kept incrementally (a triplet that
// enters adds its current count, one that leaves takes away its count after leaving).  Everything but ACGT / acgt is skipped
// as if it were not there.  Integer state, one double expression at the end: equal to the reference bit for bit.
extern "C" double pmx_read_dust(const char* seq, int64_t len, int32_t window) {
    if (!seq || len <= 0 || window < 3) return 0.0;
    int32_t occ[64] = {0};
    std::vector<uint8_t> ring((size_t)window, 0);
    long long pairs = 0, most = 0;
    unsigned code = 0;
    long long seen = -3;                 // triplets completed so far, minus one
    for (int64_t i = 0; i < len; ++i) {
        unsigned b;
        switch (seq[i]) {
            case 'A': case 'a': b = 0; break;
            case 'C': case 'c': b = 1; break;
            case 'G': case 'g': b = 2; break;
            case 'T': case 't': b = 3; break;
            default: continue;
        }
        code = ((code << 2) | b) & 63u;
        if (++seen < 0) continue;
        const size_t slot = (size_t)(seen % window);
        const bool full = seen >= window;
        if (full) {
            const unsigned old = ring[slot];
            if (occ[old] > 0) pairs -= --occ[old];
        }
        pairs += occ[code]++;
        if (full && pairs > most) most = pairs;
        ring[slot] = (uint8_t)code;
    }
    if (seen >= window) return (200.0 * most) / (window * (window - 1));
    if (seen + 1 > 1) return (200.0 * pairs) / (seen * (seen + 1));
    return 0.0;
}

// ------------------------------------------------------------------------------------------------- the options table
// (device/pmx_options.hpp).  Readers take the current table through an atomic pointer; a reload builds a new one and swaps
// it in (the old one is kept: a reader may still hold a value pointer into it; reloads are rare).
#include <atomic>
#include <mutex>

#include "device/pmx_options.hpp"
namespace pmx {
namespace {
struct OptTable {
    std::string val[O_COUNT];
    bool set[O_COUNT];
};
const char* const kOptName[O_COUNT] = {
#define PMX_OPT_NAME(name, cls, doc) "PMX_" #name,
    PMX_OPTION_TABLE(PMX_OPT_NAME)
#undef PMX_OPT_NAME
};
const char* const kOptClass[O_COUNT] = {
#define PMX_OPT_CLASS(name, cls, doc) cls,
    PMX_OPTION_TABLE(PMX_OPT_CLASS)
#undef PMX_OPT_CLASS
};
const char* const kOptDoc[O_COUNT] = {
#define PMX_OPT_DOC(name, cls, doc) doc,
    PMX_OPTION_TABLE(PMX_OPT_DOC)
#undef PMX_OPT_DOC
};
std::atomic<const OptTable*> g_opt{nullptr};
std::mutex g_opt_mu;
const OptTable* read_env() {
    OptTable* t = new OptTable();
    for (int i = 0; i < O_COUNT; ++i) {
        const char* v = getenv(kOptName[i]);
        t->set[i] = v != nullptr;
        if (v) t->val[i] = v;
    }
    return t;
}
const OptTable* table() {
    const OptTable* t = g_opt.load(std::memory_order_acquire);
    if (t) return t;
    std::lock_guard<std::mutex> lk(g_opt_mu);
    t = g_opt.load(std::memory_order_acquire);
    if (!t) { t = read_env(); g_opt.store(t, std::memory_order_release); }
    return t;
}
}  // namespace
const char* opt_str(OptId id) {
    const OptTable* t = table();
    return t->set[id] ? t->val[id].c_str() : nullptr;
}
void options_reload() {
    std::lock_guard<std::mutex> lk(g_opt_mu);
    g_opt.store(read_env(), std::memory_order_release);
}
size_t options_describe(char* buf, size_t cap) {
    const OptTable* t = table();
    std::string s;
    for (int i = 0; i < O_COUNT; ++i) {
        s += kOptName[i]; s += " ["; s += kOptClass[i]; s += "] "; s += kOptDoc[i];
        if (t->set[i]) { s += " (= "; s += t->val[i]; s += ")"; }
        s += "\n";
    }
    if (buf && cap > 0) { const size_t n = std::min(cap - 1, s.size()); memcpy(buf, s.data(), n); buf[n] = 0; }
    return s.size() + 1;
}
}  // namespace pmx
extern "C" void pmx_options_reload(void) { pmx::options_reload(); }
extern "C" int64_t pmx_options_describe(char* buf, int64_t cap) { return (int64_t)pmx::options_describe(buf, cap > 0 ? (size_t)cap : 0); }
