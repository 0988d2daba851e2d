// PanMAN reader / genome materialiser (see panman.hpp).  Format facts: SURVEY.md Appendix A.
#include "panman.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>

#include "capnp_lite.hpp"

namespace pmx {

// ------------------------------------------------------------------------------------ xz
std::vector<uint8_t> xz_decompress(const uint8_t* in, size_t n) {
    // lzma_stream_buffer_decode(uint64_t* memlimit, uint32_t flags, const lzma_allocator*,
    //                           const uint8_t* in, size_t* in_pos, size_t in_size,
    //                           uint8_t* out, size_t* out_pos, size_t out_size) -> lzma_ret
    using decode_fn = int (*)(uint64_t*, uint32_t, const void*, const uint8_t*, size_t*, size_t, uint8_t*, size_t*, size_t);
    static decode_fn fn = nullptr;
    if (!fn) {
        void* h = dlopen("liblzma.so.5", RTLD_NOW | RTLD_GLOBAL);
        if (!h) throw std::runtime_error(std::string("cannot load liblzma.so.5: ") + dlerror());
        fn = reinterpret_cast<decode_fn>(dlsym(h, "lzma_stream_buffer_decode"));
        if (!fn) throw std::runtime_error("liblzma.so.5 lacks lzma_stream_buffer_decode");
    }
    size_t cap = n * 12 + (1u << 20);
    for (int attempt = 0; attempt < 8; ++attempt) {
        std::vector<uint8_t> out(cap);
        uint64_t memlimit = UINT64_MAX;
        size_t in_pos = 0, out_pos = 0;
        int ret = fn(&memlimit, 0, nullptr, in, &in_pos, n, out.data(), &out_pos, out.size());
        if (ret == 0) {  // LZMA_OK
            out.resize(out_pos);
            return out;
        }
        if (ret == 10) {  // LZMA_BUF_ERROR: output too small
            cap *= 4;
            continue;
        }
        throw std::runtime_error("xz decode failed, lzma_ret=" + std::to_string(ret));
    }
    throw std::runtime_error("xz decode: output larger than expected");
}

// ------------------------------------------------------------------------------ alphabet
char nuc_from_code(int code) {
    switch (code) {
        case 1: return 'A';
        case 2: return 'C';
        case 4: return 'G';
        case 8: return 'T';
        case 5: return 'R';
        case 10: return 'Y';
        case 6: return 'S';
        case 9: return 'W';
        case 12: return 'K';
        case 3: return 'M';
        case 14: return 'B';
        case 13: return 'D';
        case 11: return 'H';
        case 7: return 'V';
        case 15: return 'N';
        default: return '-';
    }
}

char complement_iupac(char c) {
    switch (c) {
        case 'A': return 'T';
        case 'T': return 'A';
        case 'C': return 'G';
        case 'G': return 'C';
        case 'R': return 'Y';
        case 'Y': return 'R';
        case 'K': return 'M';
        case 'M': return 'K';
        case 'B': return 'V';
        case 'V': return 'B';
        case 'D': return 'H';
        case 'H': return 'D';
        default: return c;  // S, W, N
    }
}

// -------------------------------------------------------------------------------- newick
namespace {
struct NewickNode {
    std::string id;
    int parent;
};

// Pre-order list of (label, parent index); children in file order; internal labels follow ')'.
std::vector<NewickNode> parse_newick(const std::string& s) {
    std::vector<NewickNode> out;
    std::vector<int> stack;  // open internal nodes
    size_t i = 0, n = s.size();
    auto read_label = [&](std::string& lab) {
        size_t j = i;
        while (j < n && s[j] != ':' && s[j] != ',' && s[j] != ')' && s[j] != '(' && s[j] != ';') ++j;
        lab.assign(s, i, j - i);
        i = j;
        if (i < n && s[i] == ':') {  // branch length
            ++i;
            while (i < n && s[i] != ',' && s[i] != ')' && s[i] != ';') ++i;
        }
    };
    while (i < n) {
        char c = s[i];
        if (c == '(') {
            NewickNode nd;
            nd.parent = stack.empty() ? -1 : stack.back();
            out.push_back(nd);
            stack.push_back((int)out.size() - 1);
            ++i;
        } else if (c == ',') {
            ++i;
        } else if (c == ')') {
            ++i;
            if (stack.empty()) throw std::runtime_error("newick: unbalanced ')'");
            int idx = stack.back();
            stack.pop_back();
            read_label(out[idx].id);
        } else if (c == ';' || c == '\n' || c == '\r' || c == ' ') {
            ++i;
        } else {  // leaf
            NewickNode nd;
            nd.parent = stack.empty() ? -1 : stack.back();
            read_label(nd.id);
            out.push_back(nd);
        }
    }
    if (!stack.empty()) throw std::runtime_error("newick: unbalanced '('");
    return out;
}
}  // namespace

// --------------------------------------------------------------------------------- parse
void parse_panman(const uint8_t* buf, size_t len, Panman& pm) {
    capnp::Message msg;
    msg.parse(buf, len);
    capnp::StructR tg = capnp::root(msg);
    capnp::ListR trees = capnp::as_list(tg.ptr(0));
    if (trees.size() < 1) throw std::runtime_error("panman: no tree in TreeGroup");
    capnp::StructR tree = trees.struct_at(0);

    std::string newick = capnp::as_text(tree.ptr(0));
    std::vector<NewickNode> nw = parse_newick(newick);
    capnp::ListR cnodes = capnp::as_list(tree.ptr(1));
    if (cnodes.size() < nw.size()) throw std::runtime_error("panman: fewer Node records than Newick nodes");

    pm = Panman();
    pm.nodes.resize(nw.size());
    int32_t max_block = -1;
    for (size_t i = 0; i < nw.size(); ++i) {
        PanmanNode& nd = pm.nodes[i];
        nd.id = nw[i].id;
        nd.parent = nw[i].parent;
        if (nd.parent >= 0) pm.nodes[nd.parent].children.push_back((int32_t)i);
        capnp::StructR cn = cnodes.struct_at((uint32_t)i);
        capnp::ListR muts = capnp::as_list(cn.ptr(0));
        for (uint32_t j = 0; j < muts.size(); ++j) {
            capnp::StructR mu = muts.struct_at(j);
            int64_t block_id = (int64_t)mu.word(0);
            int32_t primary = (int32_t)(block_id >> 32);
            uint64_t w1 = mu.word(1);
            bool mut_exist = (w1 >> 1) & 1, mut_info = (w1 >> 2) & 1, inversion = (w1 >> 3) & 1;
            if (primary > max_block) max_block = primary;
            if (mut_exist) nd.block_muts.push_back(BlockMut{primary, mut_info, inversion});
            capnp::ListR nms = capnp::as_list(mu.ptr(0));
            for (uint32_t q = 0; q < nms.size(); ++q) {
                capnp::StructR nm = nms.struct_at(q);
                NucMut x;
                x.block = primary;
                x.pos = nm.get<int32_t>(0);
                int32_t gap_pos = nm.get<int32_t>(4);
                bool gap_exist = nm.bit(64);
                x.gap = gap_exist ? gap_pos : -1;
                uint32_t mi = nm.get<uint32_t>(12);
                x.len = (uint8_t)((mi & 0xff) >> 4);
                x.type = (uint8_t)(mi & 0xf);
                x.nucs = x.len <= 6 ? ((mi >> 8) << (24 - 4 * x.len)) & 0xffffffu : 0;
                nd.nuc_muts.push_back(x);
            }
        }
    }

    // blocks: several ids may share one consensus sequence (ConsensusSeqToBlockIds)
    capnp::ListR cmap = capnp::as_list(tree.ptr(2));
    for (uint32_t i = 0; i < cmap.size(); ++i) {
        capnp::ListR ids = capnp::as_list(cmap.struct_at(i).ptr(0));
        for (uint32_t j = 0; j < ids.size(); ++j) {
            int32_t primary = (int32_t)(ids.prim<int64_t>(j) >> 32);
            if (primary > max_block) max_block = primary;
        }
    }
    pm.n_blocks = max_block + 1;
    std::vector<std::string> consensus(pm.n_blocks);
    for (uint32_t i = 0; i < cmap.size(); ++i) {
        capnp::StructR e = cmap.struct_at(i);
        capnp::ListR ids = capnp::as_list(e.ptr(0));
        capnp::ListR seq = capnp::as_list(e.ptr(1));
        std::string s;
        bool done = false;
        for (uint32_t w = 0; w < seq.size() && !done; ++w) {
            uint32_t v = seq.prim<uint32_t>(w);
            for (int j = 0; j < 8; ++j) {  // src/panmap_utils.hpp:204-213
                int code = (v >> (4 * (7 - j))) & 15;
                if (code == 0) { done = true; break; }
                s.push_back(nuc_from_code(code));
            }
        }
        for (uint32_t j = 0; j < ids.size(); ++j) consensus[(int32_t)(ids.prim<int64_t>(j) >> 32)] = s;
    }
    pm.block_len.resize(pm.n_blocks);
    pm.gap_len.resize(pm.n_blocks);
    for (int32_t b = 0; b < pm.n_blocks; ++b) {
        pm.block_len[b] = (int32_t)consensus[b].size() + 1;  // + 'x' sentinel (src/panmap_utils.hpp:243)
        pm.gap_len[b].assign(pm.block_len[b], 0);
    }
    capnp::ListR gaps = capnp::as_list(tree.ptr(3));
    for (uint32_t i = 0; i < gaps.size(); ++i) {
        capnp::StructR g = gaps.struct_at(i);
        int32_t primary = (int32_t)((int64_t)g.word(0) >> 32);
        capnp::ListR glen = capnp::as_list(g.ptr(0));
        capnp::ListR gpos = capnp::as_list(g.ptr(1));
        if (primary < 0 || primary >= pm.n_blocks) continue;
        for (uint32_t j = 0; j < gpos.size() && j < glen.size(); ++j) {
            int32_t p = gpos.prim<int32_t>(j), l = glen.prim<int32_t>(j);
            if (p >= 0 && p < pm.block_len[primary] && l >= 0) pm.gap_len[primary][p] = (uint32_t)l;  // last wins
        }
    }
    // column layout
    pm.block_col0.assign(pm.n_blocks + 1, 0);
    pm.pos_col.resize(pm.n_blocks);
    uint32_t col = 0;
    for (int32_t b = 0; b < pm.n_blocks; ++b) {
        pm.block_col0[b] = col;
        pm.pos_col[b].resize(pm.block_len[b]);
        for (int32_t p = 0; p < pm.block_len[b]; ++p) {
            col += pm.gap_len[b][p];
            pm.pos_col[b][p] = col++;
        }
    }
    pm.block_col0[pm.n_blocks] = col;
    pm.n_cols = col;
    pm.consensus_cols.assign(col, '-');
    pm.col_block.assign(col, 0);
    for (int32_t b = 0; b < pm.n_blocks; ++b) {
        for (uint32_t c = pm.block_col0[b]; c < pm.block_col0[b + 1]; ++c) pm.col_block[c] = (uint32_t)b;
        for (int32_t p = 0; p < pm.block_len[b]; ++p)
            pm.consensus_cols[pm.pos_col[b][p]] = p + 1 == pm.block_len[b] ? 'x' : consensus[b][p];
    }
}

void load_panman(const std::string& path, Panman& out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<uint8_t> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    std::vector<uint8_t> dec;
    if (raw.size() >= 6 && raw[0] == 0xFD && raw[1] == '7' && raw[2] == 'z' && raw[3] == 'X' && raw[4] == 'Z' && raw[5] == 0)
        dec = xz_decompress(raw.data(), raw.size());
    else
        dec.swap(raw);
    // capnp needs 8-byte alignment; std::vector<uint8_t> storage from operator new is 16-byte aligned
    parse_panman(dec.data(), dec.size(), out);
}

int32_t Panman::find_node(const std::string& id) const {
    for (size_t i = 0; i < nodes.size(); ++i)
        if (nodes[i].id == id) return (int32_t)i;
    return -1;
}

// --------------------------------------------------------------------------------- state
void PanmanState::init(const Panman& pm) {
    cols = pm.consensus_cols;
    block_exists.assign(pm.n_blocks, 0);
    block_fwd.assign(pm.n_blocks, 1);
}

void apply_node(const Panman& pm, int32_t ni, PanmanState& st, UndoLog* undo, std::vector<ColRange>* ranges) {
    const PanmanNode& nd = pm.nodes[ni];
    std::vector<uint8_t> old_exists, old_fwd;
    if (ranges) { old_exists = st.block_exists; old_fwd = st.block_fwd; }
    for (const BlockMut& bm : nd.block_muts) {  // src/panmap_utils.hpp:742-781
        int32_t b = bm.block;
        if (b < 0 || b >= pm.n_blocks) continue;
        if (undo) undo->blocks.push_back({b, {st.block_exists[b], st.block_fwd[b]}});
        if (bm.insertion) {
            st.block_exists[b] = 1;
            st.block_fwd[b] = !bm.inversion;
        } else if (bm.inversion) {
            st.block_fwd[b] = !st.block_fwd[b];
        } else {
            st.block_exists[b] = 0;
            st.block_fwd[b] = 1;
        }
        if (ranges && pm.block_col0[b + 1] > pm.block_col0[b])
            ranges->push_back(ColRange{pm.block_col0[b], pm.block_col0[b + 1] - 1});
    }
    for (const NucMut& nm : nd.nuc_muts) {  // src/panmap_utils.hpp:783-842
        int32_t b = nm.block;
        if (b < 0 || b >= pm.n_blocks) continue;
        int last = -1;
        int64_t c0 = -1, c_last = -1;
        for (int i = 0; i < nm.len; ++i) {
            int32_t pos = nm.gap < 0 ? nm.pos + i : nm.pos;
            int32_t gap = nm.gap < 0 ? -1 : nm.gap + i;
            // the trailing sentinel's main base and anything past the block are skipped (:792-795)
            if ((pos == pm.block_len[b] - 1 && gap == -1) || pos >= pm.block_len[b] || pos < 0) continue;
            int64_t c = pm.column(b, pos, gap);
            if (c < 0) continue;
            last = i;
            c_last = c;
            int code = (nm.nucs >> (4 * (5 - i))) & 0xf;
            char nw = nuc_from_code(code);
            char old = st.cols[c];
            if (old == nw) continue;
            if (undo) undo->col_changes.push_back({(uint32_t)c, old});
            st.cols[c] = nw;
        }
        if (last >= 0 && ranges) {
            // range from offset 0 (even if offset 0 itself was skipped) to the last valid offset
            int32_t pos0 = nm.pos, gap0 = nm.gap;
            c0 = pm.column(b, pos0, gap0);
            if (c0 < 0) c0 = c_last;
            bool same = old_exists[b] && st.block_exists[b] && old_fwd[b] == st.block_fwd[b];
            if (same) {
                uint32_t a = (uint32_t)std::min(c0, c_last), e = (uint32_t)std::max(c0, c_last);
                ranges->push_back(ColRange{a, e});
            }
        }
    }
}

void undo_node(PanmanState& st, const UndoLog& undo) {
    for (size_t i = undo.col_changes.size(); i-- > 0;) st.cols[undo.col_changes[i].first] = undo.col_changes[i].second;
    for (size_t i = undo.blocks.size(); i-- > 0;) {
        st.block_exists[undo.blocks[i].first] = undo.blocks[i].second.first;
        st.block_fwd[undo.blocks[i].first] = undo.blocks[i].second.second;
    }
}

std::string genome_of_state(const Panman& pm, const PanmanState& st) {
    std::string g;
    g.reserve(pm.n_cols);
    for (int32_t b = 0; b < pm.n_blocks; ++b) {
        if (!st.block_exists[b]) continue;
        uint32_t c0 = pm.block_col0[b], c1 = pm.block_col0[b + 1];
        if (st.block_fwd[b]) {
            for (uint32_t c = c0; c < c1; ++c) {
                char ch = st.cols[c];
                if (ch != '-' && ch != 'x') g.push_back(ch);
            }
        } else {  // inverted block: reverse complement (src/panmap_utils.cpp:156-173)
            for (uint32_t c = c1; c-- > c0;) {
                char ch = st.cols[c];
                if (ch != '-' && ch != 'x') g.push_back(complement_iupac(ch));
            }
        }
    }
    return g;
}

PanmanState root_state_of(const Panman& pm) {
    PanmanState st;
    st.init(pm);
    if (!pm.nodes.empty()) apply_node(pm, 0, st, nullptr, nullptr);
    return st;
}

std::string node_genome(const Panman& pm, int32_t ni, const PanmanState* root_state) {
    std::vector<int32_t> path;
    for (int32_t x = ni; x >= 0; x = pm.nodes[x].parent) path.push_back(x);
    std::reverse(path.begin(), path.end());
    PanmanState st;
    size_t first = 0;
    if (root_state && !path.empty() && path[0] == 0) { st = *root_state; first = 1; }
    else st.init(pm);
    for (size_t k = first; k < path.size(); ++k) apply_node(pm, path[k], st, nullptr, nullptr);
    return genome_of_state(pm, st);
}

}  // namespace pmx
