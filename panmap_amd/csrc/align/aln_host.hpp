// ALIGN stage, host side (plain C++, no HIP): option presets, reference minimizer index build, work
// memory layout.  Shared by the product (api_align.hip) and the CPU unit-test build (tests/hostsim).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <memory>
#include <vector>

#include "aln_map.hpp"

namespace pmx {
namespace aln {

// mm_mapopt_init + mm_idxopt_init (options.c:5-64), then the branch setup_minimap2(for_scoring=1) takes
// for the mean read length (src/mm_align.c:118-180; "map-hifi" per options.c:109-116).
// mid_occ for the long-read presets is filled in by finish_opt() once the index is known.
inline Opt make_opt(int mean_read_len) {
    Opt o;
    memset(&o, 0, sizeof(o));
    o.k = 15; o.w = 10;
    o.seed = 11;
    float mid_occ_frac = 2e-4f;
    (void)mid_occ_frac;
    o.q_occ_frac = 0.01f;
    o.min_cnt = 3; o.min_chain_score = 40;
    o.bw = 500; o.bw_long = 20000;
    o.max_gap = 5000; o.max_gap_ref = -1;
    o.max_chain_skip = 25; o.max_chain_iter = 5000;
    o.rmq_rescue_size = 1000; o.rmq_rescue_ratio = 0.1f; o.rmq_inner_dist = 1000; o.rmq_size_cap = 100000;
    o.chain_gap_scale = 0.8f; o.chain_skip_scale = 0.0f;
    o.max_max_occ = 4095; o.occ_dist = 500;
    o.mask_level = 0.5f; o.mask_len = INT32_MAX;
    o.pri_ratio = 0.8f; o.best_n = 5;
    o.alt_drop = 0.15f;
    o.a = 2; o.b = 4; o.q = 4; o.e = 2; o.q2 = 24; o.e2 = 1;
    o.sc_ambi = 1;
    o.zdrop = 400; o.zdrop_inv = 200;
    o.end_bonus = -1;
    o.min_dp_max = o.min_chain_score * o.a;
    o.min_ksw_len = 200;
    o.max_clip_ratio = 1.0f;
    o.max_sw_mat = 100000000;
    o.rank_min_len = 500; o.rank_frac = 0.9f;
    o.pe_ori = 0; o.pe_bonus = 33;
    o.mid_occ = 0; o.max_occ = 0;
    if (mean_read_len < 500) {   // src/mm_align.c:140-166
        o.is_sr_like = 1;
        o.k = 21; o.w = 11;
        o.a = 2; o.b = 8; o.q = 12; o.e = 2; o.q2 = 24; o.e2 = 1;
        o.zdrop = 100; o.zdrop_inv = 100;
        o.end_bonus = 10;
        o.max_frag_len = 800;
        o.max_gap = 100;
        o.bw = 100; o.bw_long = 100;
        o.pri_ratio = 0.5f;
        o.min_cnt = 2; o.min_chain_score = 25; o.min_dp_max = 40;
        o.best_n = 20;
        o.mid_occ = 1000; o.max_occ = 5000;
        o.pe_ori = 0 << 1 | 1;
        o.pe_bonus = 33;
    } else if (mean_read_len < 5000) {
        // "map-ont" == defaults
    } else {   // "map-hifi"
        o.k = 19; o.w = 19;
        o.max_gap = 10000;
        o.a = 1; o.b = 4; o.q = 6; o.q2 = 26; o.e = 2; o.e2 = 1;
        o.occ_dist = 500;
        o.min_dp_max = 200;
    }
    return o;
}

struct HostRefIndex {
    std::vector<uint8_t> seq;
    std::vector<HtEnt> ht;
    std::vector<uint64_t> pos;
    std::vector<float> logf_ratio, logf_int;
    std::vector<uint64_t> pk, pk_amb;   // RefIndex::pk / pk_amb
    std::vector<uint32_t> ht_pv;        // RefIndex::ht_pv
    std::vector<uint32_t> occ;   // occurrences per distinct minimizer
    int logf_a = 0;              // match score the logf tables were built for
    RefIndex view() const {
        RefIndex r;
        r.seq = seq.data();
        r.len = (int32_t)seq.size() - 8;   // seq carries 8 bytes of padding
        r.ht_mask = (uint32_t)ht.size() - 1;
        r.ht = ht.data();
        r.pos = pos.data();
        r.logf_ratio = logf_ratio.data();
        r.logf_int = logf_int.data();
        r.n_logf = (int32_t)logf_int.size();
        r.pk = pk.data();
        r.pk_amb = pk_amb.data();
        r.ht_pv = ht_pv.data();
        return r;
    }
};

inline uint8_t nt4_of_char(unsigned char c) {   // seq_nt4_table (sketch.c:9-26)
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': case 'U': case 'u': return 3;
        default: return 4;
    }
}

// mid_occ from the occurrence counts of the index when the preset left it unset (mm_mapopt_update, options.c:66-81);
// a_kk = the kk-th smallest count, kk = (uint32)((1 - 2e-4f) * n_keys) clamped to the last one (INT32_MAX - 1 without keys)
inline size_t mid_occ_rank(size_t n_keys) {
    const float f = 2e-4f;
    const size_t kk = (size_t)(uint32_t)((1. - f) * n_keys);
    return n_keys ? std::min(kk, n_keys - 1) : 0;
}
inline void set_mid_occ(Opt& o, int32_t a_kk_plus_1) {
    int min_mid_occ = 10, max_mid_occ = 1000000;
    if (o.k == 19 && o.w == 19) { min_mid_occ = 50; max_mid_occ = 500; }   // map-hifi
    o.mid_occ = a_kk_plus_1;
    if (o.mid_occ < min_mid_occ) o.mid_occ = min_mid_occ;
    if (max_mid_occ > min_mid_occ && o.mid_occ > max_mid_occ) o.mid_occ = max_mid_occ;
}
// what the options and the score tables take from the reference once its index exists; returns true when the logf tables
// were (re)computed (they only depend on (a, size) and are kept across references)
inline bool finish_ref_opt(Opt& o, int max_dp_score, HostRefIndex& out) {
    if (o.bw_long < o.bw) o.bw_long = o.bw;
    o.chn_pen_gap = (float)(o.chain_gap_scale * 0.01 * o.k);
    o.chn_pen_skip = (float)(o.chain_skip_scale * 0.01 * o.k);
    gen_simple_mat(o.mat, (int8_t)o.a, (int8_t)o.b, (int8_t)o.sc_ambi);
    // logf tables from the host libm (hit.c:440-457, pe.c:160)
    const int n = max_dp_score + 2;
    const bool keep = out.logf_a == o.a && (int)out.logf_ratio.size() == n && (int)out.logf_int.size() == n;
    if (!keep) {
        out.logf_ratio.resize(n);
        out.logf_int.resize(n);
        for (int i = 0; i < n; ++i) {
            out.logf_ratio[i] = logf((float)i / o.a);
            out.logf_int[i] = logf((float)i);
        }
    }
    out.logf_a = o.a;
    return !keep;
}

// mm_idx_str(w, k, 0, 14, 1, &ref) (index.c:408-451): sketch the reference, group occurrences by minimizer
#ifndef PMX_INTERLEAVED   // host-only (raw pointers); the thread-per-pair device pass skips it
inline bool build_ref_index(const char* ref, int64_t ref_len, Opt& o, int max_dp_score, HostRefIndex& out) {
    // the logf tables only depend on (a, size): keep them across references
    std::vector<float> keep_ratio, keep_int;
    const int keep_a = out.logf_a;
    keep_ratio.swap(out.logf_ratio);
    keep_int.swap(out.logf_int);
    out = HostRefIndex();
    out.logf_ratio.swap(keep_ratio);
    out.logf_int.swap(keep_int);
    out.logf_a = keep_a;
    out.seq.assign((size_t)ref_len + 8, 0);   // padded: readers fetch aligned 32-bit words
    out.pk.assign((size_t)(ref_len + 31) / 32 + 2, 0);
    out.pk_amb.assign((size_t)(ref_len + 31) / 32 + 2, 0);
    for (int64_t i = 0; i < ref_len; ++i) {
        const uint8_t c = nt4_of_char((unsigned char)ref[i]);
        out.seq[i] = c;
        if (c < 4) out.pk[(size_t)(i >> 5)] |= (uint64_t)c << (2 * (i & 31));
        else out.pk_amb[(size_t)(i >> 5)] |= 1ULL << (2 * (i & 31));
    }
    o.ref_len = (int)ref_len;
    Work W;
    memset(&W, 0, sizeof(W));
    const size_t mv_cap = (size_t)ref_len + 16;
    std::unique_ptr<A128[]> mv_buf(new A128[mv_cap]);   // uninitialised: the sketch writes what it uses
    std::vector<A128> buf(256);
    W.mv = mv_buf.get();
    W.sk_buf = buf.data();
    W.caps.max_mini = (int)mv_cap;
    W.n_mv = 0;
    if (ref_len > 0) {   // on a CPU the branchy ring walk is ~4x faster than the branch-free register form the kernels use
        RingMem ring{W.sk_buf};
        sketch_segment_t(W, ring, (Ptr<const uint8_t>)out.seq.data(), (int)ref_len, o.w, o.k, 0);
    }
    // by minimizer value (span byte excluded), then by position: the occurrence list of a key is ascending in y
    std::vector<A128> mv((size_t)W.n_mv);
    {
        const A128* src = mv_buf.get();
        const size_t n = (size_t)W.n_mv;
        const auto by_key_pos = [](const A128& a, const A128& b) { return (a.x >> 8) != (b.x >> 8) ? (a.x >> 8) < (b.x >> 8) : a.y < b.y; };
        if (2 * o.k <= 42 && n < ((size_t)1 << 21)) {
            // one 64-bit key per minimizer (value << 21 | emission index), then the rare equal-value runs by position
            std::vector<uint64_t> key(n);
            for (size_t i = 0; i < n; ++i) key[i] = (src[i].x >> 8) << 21 | (uint64_t)i;
            std::sort(key.begin(), key.end());
            for (size_t i = 0; i < n; ++i) mv[i] = src[key[i] & (((uint64_t)1 << 21) - 1)];
            for (size_t i = 0; i < n;) {
                size_t j = i + 1;
                while (j < n && mv[j].x >> 8 == mv[i].x >> 8) ++j;
                if (j - i > 1) std::sort(mv.begin() + (ptrdiff_t)i, mv.begin() + (ptrdiff_t)j, by_key_pos);
                i = j;
            }
        } else {
            std::copy(src, src + n, mv.begin());
            std::sort(mv.begin(), mv.end(), by_key_pos);
        }
    }
    size_t n_keys = 0;
    for (size_t i = 0; i < mv.size(); ++i)
        if (i == 0 || mv[i].x >> 8 != mv[i - 1].x >> 8) ++n_keys;
    size_t cap = 16;
    while (cap < n_keys * 2 + 2) cap <<= 1;
    out.ht.assign(cap, HtEnt{UINT64_MAX, 0u, 0u});
    out.ht_pv.assign(cap, 0xffffffffu);
    out.pos.resize(mv.size());
    for (size_t i = 0; i < mv.size();) {
        size_t j = i;
        while (j < mv.size() && mv[j].x >> 8 == mv[i].x >> 8) ++j;
        for (size_t q = i; q < j; ++q) out.pos[q] = mv[q].y;
        const uint64_t key = mv[i].x >> 8;
        uint32_t slot = (uint32_t)mix64(key) & (uint32_t)(cap - 1);
        while (out.ht[slot].key != UINT64_MAX) slot = (slot + 1) & (uint32_t)(cap - 1);
        out.ht[slot] = HtEnt{key, (uint32_t)i, (uint32_t)(j - i)};
        if (j - i == 1 && (mv[i].y >> 32) == 0) out.ht_pv[slot] = (uint32_t)mv[i].y;
        out.occ.push_back((uint32_t)(j - i));
        i = j;
    }
    // mm_mapopt_update (options.c:66-81): mid_occ from the index when the preset left it unset
    if (o.mid_occ <= 0) {
        int32_t thres = INT32_MAX;
        if (!out.occ.empty()) {
            std::vector<uint32_t> a = out.occ;
            const size_t kk = mid_occ_rank(a.size());
            std::nth_element(a.begin(), a.begin() + kk, a.end());
            thres = (int32_t)a[kk] + 1;
        }
        set_mid_occ(o, thres);
    }
    return finish_ref_opt(o, max_dp_score, out);
}

#endif
// ------------------------------------------------------------------------------ work memory layout
enum { PMX_FAST = 0, PMX_SLOW = 1, PMX_RAW = 2 };   // RAW: thread-per-pair layout only (per-thread contiguous structs)

struct Layout {
    Caps caps;
    size_t fast_bytes = 0, slow_bytes = 0, raw_bytes = 0;
    size_t tb_cap = 0;
    struct Ent { uint32_t space; size_t off; };
    Ent qseq, mv, sk_buf, seeds, seeds_b, mini_pos, heap, a, a2, cc, kidx, z, u, u2, regs0, regs1, regs2, reg_tmp, seg_a0, seg_a1, seg_u0,
        seg_u1, aux64, aux32, aux128, du, sf, qr, H, off_, tseq, cig_tmp, cig_pool, tb, tb_fast, dp_fast;
    size_t tb_fast_cap = 0;   // DP-service layout only: LDS traceback area for small DPs
};

// Capacities for a read-length regime; fast = LDS budget (bytes) per wave.
// tb_limit: cap of the per-wave traceback area (a DP that needs more reports PMX_ST_OVERFLOW and is re-run by a launch
// with the full capacity: long reads, whose band allows matrices up to max_sw_mat cells, align.c:326-328, 590-592)
// anchor_scale > 1: the last-resort layout for reads whose minimizers hit a repeat of the reference hundreds of times each (a
// poly-A mate against a genome that ends in a poly-A tail): anchors, chains and regions by that factor
// dp_fast_tlen > 0 (long reads: the DP arrays, sized for the longest allowed target, do not fit LDS): a second, small set
// of them in LDS for the DPs of at most that size -- nearly all of them (a gap fill between two anchors is a few hundred
// bases)
inline Layout plan_layout(int max_read_len, int n_segs, const Opt& o, size_t fast_budget, size_t tb_limit = 0, int anchor_scale = 1, int dp_fast_tlen = 0) {
    Layout L;
    Caps& c = L.caps;
    c.dp_fast_tlen = dp_fast_tlen;
    L.dp_fast.space = PMX_FAST; L.dp_fast.off = 0;
    c.max_qlen = (max_read_len + 15) / 16 * 16 + 16;
    const int qsum = c.max_qlen * n_segs;
    // a low-complexity read emits a minimizer at nearly every k-mer position (equal hashes in a window are all kept,
    // sketch.c:107-139), and a short tandem repeat re-emits the other copies of the minimum every time it slides out
    // of the window: the general layout takes four per base
    c.max_mini = std::max(64, qsum * 4 + 16);
    c.max_anchor = std::max(256, c.max_mini * 2) * anchor_scale;
    c.max_reg = anchor_scale > 1 ? 256 : 32;
    c.max_cigar = std::max(64, max_read_len / 2 + 16);
    // DP target length: extension <= query part + gap allowance (align.c:652-688)
    c.max_tlen = ((std::min(max_read_len, std::max(o.max_gap, o.min_ksw_len + 2 * o.k) * 4) + 2 * o.max_gap + 64) + 15) / 16 * 16 + 32;
    if (c.max_tlen < c.max_qlen + 64) c.max_tlen = c.max_qlen + 64;
    c.n_cig_slots = c.max_reg * 2 * n_segs + 4;
    const int wband = (int)(std::max(o.bw, o.bw_long) * 1.5 + 1.);
    const int n_col = ((std::min(std::min(c.max_tlen, c.max_qlen), wband + 1) + 15) / 16 + 1) * 16;
    L.tb_cap = (size_t)(c.max_qlen + c.max_tlen) * n_col;
    if (o.max_sw_mat > 0 && L.tb_cap > (size_t)o.max_sw_mat + (size_t)(c.max_qlen + c.max_tlen) * 32)   // larger matrices are never filled (align.c:326)
        L.tb_cap = (size_t)o.max_sw_mat + (size_t)(c.max_qlen + c.max_tlen) * 32;
    if (tb_limit > 0 && L.tb_cap > tb_limit) L.tb_cap = tb_limit;

    size_t used[2] = {0, 0};
    auto place = [&](Layout::Ent& e, size_t bytes, int pref) {
        bytes = (bytes + 15) & ~(size_t)15;
        int sp = pref;
        if (sp == PMX_FAST && used[PMX_FAST] + bytes > fast_budget) sp = PMX_SLOW;
        e.space = (uint32_t)sp;
        e.off = used[sp];
        used[sp] += bytes;
    };
    if (dp_fast_tlen > 0) place(L.dp_fast, (size_t)9 * (dp_fast_tlen + 32) + 64, PMX_FAST);
    if (L.dp_fast.space != PMX_FAST) c.dp_fast_tlen = 0;
    // hot small arrays first so they win the LDS budget
    place(L.du, (size_t)7 * (c.max_tlen + 32), PMX_FAST);       // u,v,x,y,x2,y2,s
    place(L.sf, (size_t)c.max_tlen + 32, PMX_FAST);
    place(L.qr, (size_t)c.max_tlen + 64, PMX_FAST);
    place(L.qseq, (size_t)4 * c.max_qlen, PMX_FAST);
    place(L.tseq, (size_t)c.max_tlen + 32, PMX_FAST);
    place(L.a, sizeof(A128) * c.max_anchor, PMX_FAST);
    place(L.cc, sizeof(ChainCell) * (size_t)c.max_anchor, PMX_FAST);
    L.kidx = L.cc;   // the chain cells are idle while regions are aligned
    place(L.regs0, sizeof(Reg) * c.max_reg, PMX_FAST);
    place(L.regs1, sizeof(Reg) * c.max_reg, PMX_FAST);
    place(L.regs2, sizeof(Reg) * c.max_reg, PMX_FAST);
    place(L.mv, sizeof(A128) * c.max_mini, PMX_FAST);
    place(L.sk_buf, sizeof(A128) * 256, PMX_SLOW);
    place(L.seeds, sizeof(SeedA) * c.max_mini, PMX_FAST);
    place(L.seeds_b, sizeof(SeedB) * c.max_mini, PMX_FAST);
    place(L.H, 4 * (size_t)(c.max_tlen + 32), PMX_FAST);
    place(L.off_, 8 * (size_t)(c.max_qlen + c.max_tlen), PMX_FAST);
    place(L.cig_tmp, 4 * (size_t)c.max_cigar, PMX_FAST);
    place(L.aux64, 8 * (size_t)c.max_reg * 8, PMX_FAST);
    place(L.aux32, 4 * (size_t)c.max_reg * 4, PMX_FAST);
    place(L.aux128, sizeof(A128) * c.max_reg * 4, PMX_FAST);
    place(L.u, 8 * (size_t)c.max_reg * 4, PMX_FAST);
    place(L.u2, 8 * (size_t)c.max_reg * 4, PMX_FAST);
    place(L.mini_pos, 8 * (size_t)c.max_mini, PMX_SLOW);
    place(L.heap, sizeof(A128) * c.max_mini, PMX_SLOW);
    place(L.a2, sizeof(A128) * c.max_anchor, PMX_SLOW);
    place(L.z, sizeof(A128) * c.max_anchor, PMX_SLOW);
    place(L.reg_tmp, sizeof(Reg) * c.max_reg, PMX_SLOW);
    place(L.seg_a0, sizeof(A128) * c.max_anchor, PMX_SLOW);
    L.seg_a1 = L.seg_a0;                       // seg_gen splits the block dynamically
    place(L.seg_u0, 8 * (size_t)c.max_reg * 4, PMX_SLOW);
    place(L.seg_u1, 8 * (size_t)c.max_reg * 4, PMX_SLOW);
    place(L.cig_pool, 4 * (size_t)c.max_cigar * c.n_cig_slots, PMX_SLOW);
    place(L.tb, L.tb_cap + 64, PMX_SLOW);
    L.fast_bytes = used[PMX_FAST];
    L.slow_bytes = used[PMX_SLOW];
    return L;
}

// Tier-1 layout for short read pairs: EVERYTHING except the traceback matrix lives in LDS, with small
// typical-case capacities and phase overlays (seeding scratch | chaining scratch | DP arrays share one
// region; they are never live together).  A pair that exceeds a capacity reports PMX_ST_OVERFLOW and is
// re-run by the tier-2 launch with the general layout.
inline Layout plan_layout_compact(int max_read_len, int n_segs, const Opt& o) {
    Layout L;
    Caps& c = L.caps;
    c.dp_fast_tlen = 0;
    L.dp_fast.space = PMX_FAST; L.dp_fast.off = 0;
    c.max_qlen = (max_read_len + 15) / 16 * 16 + 16;
    // sized for EIGHT waves per CU (20.2 KB of LDS per wave): the pairs that come here are few and the tier is bound by the
    // latency of a pair, so resident waves are what counts (96 / 128 / 8 regions / 20 CIGAR slots = 24.3 KB = six waves:
    // 2,760 bails of the bench batch took 1.80 ms, now 1.43); what overflows goes to the general layout as before
    c.max_mini = 80;
    c.max_anchor = 96;
    c.max_reg = 5;
    c.max_cigar = 32;
    c.max_tlen = (c.max_qlen + o.max_gap + 15) / 16 * 16 + 16;
    c.n_cig_slots = 14;
    const int wband = (int)(std::max(o.bw, o.bw_long) * 1.5 + 1.);
    const int n_col = ((std::min(std::min(c.max_tlen, c.max_qlen), wband + 1) + 15) / 16 + 1) * 16;
    L.tb_cap = (size_t)(c.max_qlen + c.max_tlen) * n_col;
    size_t top = 0;
    auto put = [&](Layout::Ent& e, size_t bytes, size_t& cursor, int space = PMX_FAST) {
        bytes = (bytes + 15) & ~(size_t)15;
        e.space = (uint32_t)space;
        e.off = cursor;
        cursor += bytes;
    };
    // persistent
    put(L.qseq, (size_t)4 * c.max_qlen, top);
    put(L.mv, sizeof(A128) * c.max_mini, top);
    put(L.a, sizeof(A128) * c.max_anchor, top);
    put(L.regs0, sizeof(Reg) * c.max_reg, top);
    put(L.regs1, sizeof(Reg) * c.max_reg, top);
    put(L.regs2, sizeof(Reg) * c.max_reg, top);
    put(L.reg_tmp, sizeof(Reg) * c.max_reg, top);
    put(L.seg_a0, sizeof(A128) * c.max_anchor, top);
    L.seg_a1 = L.seg_a0;                       // seg_gen splits the block dynamically
    put(L.seg_u0, 8 * (size_t)c.max_reg * 4, top);
    put(L.seg_u1, 8 * (size_t)c.max_reg * 4, top);
    put(L.u, 8 * (size_t)c.max_reg * 4, top);
    put(L.aux64, 8 * (size_t)c.max_reg * 8, top);
    put(L.aux32, 4 * (size_t)c.max_reg * 4, top);
    put(L.aux128, sizeof(A128) * c.max_reg * 4, top);
    put(L.cig_pool, 4 * (size_t)c.max_cigar * c.n_cig_slots, top);
    // overlay region
    size_t ca = top, cb = top, cd = top;
    put(L.sk_buf, sizeof(A128) * 32, ca);
    put(L.seeds, sizeof(SeedA) * c.max_mini, ca);
    put(L.seeds_b, sizeof(SeedB) * c.max_mini, ca);
    put(L.heap, sizeof(A128) * c.max_mini, ca);
    put(L.cc, sizeof(ChainCell) * (size_t)c.max_anchor, cb);
    L.kidx = L.cc;   // the chain cells are idle while regions are aligned
    put(L.z, sizeof(A128) * c.max_anchor, cb);
    put(L.a2, sizeof(A128) * c.max_anchor, cb);
    put(L.u2, 8 * (size_t)c.max_reg * 4, cb);
    put(L.du, (size_t)7 * (c.max_tlen + 32), cd);
    put(L.sf, (size_t)c.max_tlen + 32, cd);
    put(L.qr, (size_t)c.max_tlen + 64, cd);
    put(L.tseq, (size_t)c.max_tlen + 32, cd);
    put(L.H, 4 * (size_t)(c.max_tlen + 32), cd);
    put(L.off_, 8 * (size_t)(c.max_qlen + c.max_tlen), cd);
    put(L.cig_tmp, 4 * (size_t)c.max_cigar, cd);
    // (the seed-position cache of the heap merge lives in the persistent seg_a block, so the seeding, chaining
    //  and DP scratch regions may all overlay each other)
    L.fast_bytes = std::max(ca, std::max(cb, cd));
    size_t slow = 0;
    put(L.mini_pos, 8 * (size_t)c.max_mini, slow, PMX_SLOW);
    put(L.tb, L.tb_cap + 64, slow, PMX_SLOW);
    L.slow_bytes = slow;
    return L;
}

// DP-service layout (k_align_dp_serve): only what ksw_extd2 touches; the request's sequences go to the qseq
// block; DPs whose traceback matrix fits tb_fast_cap keep it in LDS (the serial traceback walk is
// latency-bound), larger ones use the wave's HBM slab.  Capacities as in the compact layout.
// small_qlen > 0 selects the SMALL class (extensions over a few dozen bases: nearly all requests), served by
// the register-resident ksw_extd2_reg: only the request's sequences, off[]/off_end[], the CIGAR buffer and an
// 8 KB traceback area live in LDS.
inline Layout plan_layout_dp(int max_read_len, int n_segs, const Opt& o, int small_qlen = 0, int small_tlen = 0) {
    Layout L = plan_layout_compact(max_read_len, n_segs, o);
    Caps& c = L.caps;
    c.dp_fast_tlen = 0;
    L.dp_fast.space = PMX_FAST; L.dp_fast.off = 0;
    if (small_qlen > 0) {
        c.max_qlen = small_qlen;
        c.max_tlen = small_tlen + 16;
        const int wband = (int)(std::max(o.bw, o.bw_long) * 1.5 + 1.);
        const int n_col = ((std::min(std::min(small_tlen, small_qlen), wband + 1) + 15) / 16 + 1) * 16;
        L.tb_cap = (size_t)(small_qlen + small_tlen) * n_col;   // HBM traceback area for the requests that do not fit LDS
    }
    size_t top = 0;
    auto put = [&](Layout::Ent& e, size_t bytes) {
        bytes = (bytes + 15) & ~(size_t)15;
        e.space = PMX_FAST;
        e.off = top;
        top += bytes;
    };
    const Layout::Ent none{PMX_FAST, 0};   // unused arrays alias offset 0 (never dereferenced by ksw_extd2)
    L.mv = L.a = L.regs0 = L.regs1 = L.regs2 = L.reg_tmp = L.seg_a0 = L.seg_a1 = L.seg_u0 = L.seg_u1 = L.u = L.aux64 = L.aux32 = L.aux128 = none;
    L.cig_pool = L.sk_buf = L.seeds = L.seeds_b = L.heap = L.cc = L.kidx = L.z = L.a2 = L.u2 = L.tseq = L.mini_pos = none;
    put(L.qseq, small_qlen > 0 ? (size_t)(((small_qlen + 15) & ~15) + small_tlen) : (size_t)std::max(4 * c.max_qlen, 512));
    if (small_qlen > 0) L.du = L.sf = L.qr = L.H = none;   // register-resident DP
    else {
        put(L.du, (size_t)7 * (c.max_tlen + 32));
        put(L.sf, (size_t)c.max_tlen + 32);
        put(L.qr, (size_t)c.max_tlen + 64);
        put(L.H, 4 * (size_t)(c.max_tlen + 32));
    }
    put(L.off_, 8 * (size_t)(c.max_qlen + c.max_tlen));
    put(L.cig_tmp, 4 * (size_t)c.max_cigar);
    L.tb_fast_cap = small_qlen > 0 ? 8 * 1024 : 12 * 1024;
    put(L.tb_fast, L.tb_fast_cap);
    L.fast_bytes = top;
    L.tb.space = PMX_SLOW;
    L.tb.off = 0;
    L.slow_bytes = L.tb_cap + 64;
    return L;
}

// Thread-per-pair layout (k_align_reads_tpp): the compact layout's arrays, addressed through IPtr (logical
// offsets: FAST region first, SLOW region behind it, interleaved across the wave by IPtr::phys), except the
// Reg struct arrays: strided structs in a per-wave RAW region (base pointer = region + lane * 4).
// tb_bytes > 0 adds a per-thread traceback area (in-lane DPs; off by default).
inline Layout plan_layout_tpp(int max_read_len, int n_segs, const Opt& o, size_t tb_bytes) {
    Layout L = plan_layout_compact(max_read_len, n_segs, o);   // capacities
    const Caps& c = L.caps;
    size_t raw = 0, top = 0;
    auto put_raw = [&](Layout::Ent& e, size_t n_regs) {   // strided Reg arrays: bytes per WAVE (see struct Reg)
        e.space = PMX_RAW;
        e.off = raw;
        raw += n_regs * (size_t)PMX_REG_STRIDED_BYTES;
    };
    auto put = [&](Layout::Ent& e, size_t bytes) {   // one region each, no overlays (see IPtr)
        e.space = PMX_FAST;
        e.off = top;
        top += (bytes + 63) & ~(size_t)63;   // 64-byte aligned: a region may be viewed through any granule (ptr_region_cast)
    };
    put_raw(L.regs0, (size_t)c.max_reg);
    put_raw(L.regs1, (size_t)c.max_reg);
    put_raw(L.regs2, (size_t)c.max_reg);
    put_raw(L.reg_tmp, (size_t)c.max_reg);
    L.raw_bytes = raw;
    put(L.qseq, (size_t)4 * c.max_qlen);
    put(L.tseq, (size_t)c.max_tlen + 32);
    put(L.mv, sizeof(A128) * c.max_mini);
    put(L.seeds, sizeof(SeedA) * c.max_mini);
    put(L.seeds_b, sizeof(SeedB) * c.max_mini);
    put(L.sk_buf, sizeof(A128) * 32);
    put(L.heap, sizeof(A128) * c.max_mini);
    put(L.mini_pos, 8 * (size_t)c.max_mini);
    put(L.a, sizeof(A128) * c.max_anchor);
    put(L.a2, sizeof(A128) * c.max_anchor);
    put(L.z, sizeof(A128) * c.max_anchor);
    put(L.cc, sizeof(ChainCell) * (size_t)c.max_anchor);
    put(L.kidx, 4 * (size_t)c.max_anchor);   // own region: a different interleave granule than the cells
    put(L.u, 8 * (size_t)c.max_reg * 4);
    put(L.u2, 8 * (size_t)c.max_reg * 4);
    put(L.seg_a0, sizeof(A128) * c.max_anchor);
    L.seg_a1 = L.seg_a0;
    put(L.seg_u0, 8 * (size_t)c.max_reg * 4);
    put(L.seg_u1, 8 * (size_t)c.max_reg * 4);
    put(L.aux64, 8 * (size_t)c.max_reg * 8);
    put(L.aux32, 4 * (size_t)c.max_reg * 4);
    put(L.aux128, sizeof(A128) * c.max_reg * 4);
    put(L.cig_tmp, 4 * (size_t)c.max_cigar);
    put(L.cig_pool, 4 * (size_t)c.max_cigar * c.n_cig_slots);
    // DP arrays: only touched by in-lane DPs (tb_bytes > 0)
    put(L.du, (size_t)7 * (c.max_tlen + 32));
    put(L.sf, (size_t)c.max_tlen + 32);
    put(L.qr, (size_t)c.max_tlen + 64);
    put(L.H, 4 * (size_t)(c.max_tlen + 32));
    put(L.off_, 8 * (size_t)(c.max_qlen + c.max_tlen));
    put(L.tb, tb_bytes + 16);
    L.tb_cap = tb_bytes;
    L.fast_bytes = top;
    L.slow_bytes = 0;
    return L;
}

// logical size of one thread's interleaved arena (FAST then SLOW), a multiple of the 16-byte granule
inline size_t tpp_arena_bytes(const Layout& L) { return ((L.fast_bytes + 63) & ~(size_t)63) + ((L.slow_bytes + 15) & ~(size_t)15); }

// Binds the Work pointers.  Wave-per-pair kernels / host: fast = LDS arena (or host buffer), slow = the wave's
// HBM slab.  Thread-per-pair kernel: fast/slow are unused (IPtr offsets), raw = the thread's struct region.
PMX_HD void bind_work(Work& W, const Layout& L, uint8_t* fast, uint8_t* slow, uint8_t* raw = nullptr) {
    uint8_t* base[3] = {fast, slow, raw};
#ifdef PMX_INTERLEAVED
    const uint32_t lbase[2] = {0u, (uint32_t)((L.fast_bytes + 63) & ~(size_t)63)};
#define PMX_AT(T, e) (IPtr<T>(lbase[L.e.space] + (uint32_t)L.e.off))
#else
#define PMX_AT(T, e) ((T*)(base[L.e.space] + L.e.off))
#endif
#define PMX_AT_RAW(T, e) ((T*)(base[L.e.space] + L.e.off))
    W.caps = L.caps;
    const int mq = L.caps.max_qlen;
    Ptr<uint8_t> qs = PMX_AT(uint8_t, qseq);
    W.qseq[0][0] = qs; W.qseq[0][1] = qs + mq; W.qseq[1][0] = qs + 2 * mq; W.qseq[1][1] = qs + 3 * mq;
    W.mv = PMX_AT(A128, mv); W.sk_buf = PMX_AT(A128, sk_buf); W.seeds = PMX_AT(SeedA, seeds); W.seeds_b = PMX_AT(SeedB, seeds_b);
    W.mini_pos = PMX_AT(uint64_t, mini_pos); W.heap = PMX_AT(A128, heap);
    W.a = PMX_AT(A128, a); W.a2 = PMX_AT(A128, a2);
    W.cc = PMX_AT(ChainCell, cc); W.kidx = PMX_AT(int32_t, kidx);
    W.z = PMX_AT(A128, z); W.u = PMX_AT(uint64_t, u); W.u2 = PMX_AT(uint64_t, u2);
    W.regs0 = PMX_AT_RAW(Reg, regs0); W.regs[0] = PMX_AT_RAW(Reg, regs1); W.regs[1] = PMX_AT_RAW(Reg, regs2);
    W.reg_tmp = PMX_AT_RAW(Reg, reg_tmp);
    W.seg_a[0] = PMX_AT(A128, seg_a0); W.seg_a[1] = PMX_AT(A128, seg_a1);
    W.seg_u[0] = PMX_AT(uint64_t, seg_u0); W.seg_u[1] = PMX_AT(uint64_t, seg_u1);
    W.aux64 = PMX_AT(uint64_t, aux64); W.aux32 = PMX_AT(int32_t, aux32); W.aux128 = PMX_AT(A128, aux128);
    Ptr<int8_t> d = PMX_AT(int8_t, du);
    const int T = L.caps.max_tlen + 32;
    W.du = d; W.dv = d + T; W.dx = d + 2 * T; W.dy = d + 3 * T; W.dx2 = d + 4 * T; W.dy2 = d + 5 * T; W.ds = d + 6 * T;
    W.sf = PMX_AT(uint8_t, sf); W.qr = PMX_AT(uint8_t, qr);
    W.H = PMX_AT(int32_t, H);
    W.off = PMX_AT(int32_t, off_); W.off_end = W.off + (L.caps.max_qlen + L.caps.max_tlen);
    W.tb = PMX_AT(uint8_t, tb); W.tb_cap = L.tb_cap;
    W.dp_fast = L.caps.dp_fast_tlen > 0 ? (int8_t*)(base[L.dp_fast.space] + L.dp_fast.off) : nullptr;
    W.tseq = PMX_AT(uint8_t, tseq);
    W.cig_tmp = PMX_AT(uint32_t, cig_tmp); W.cig_pool = PMX_AT(uint32_t, cig_pool);
#undef PMX_AT
#undef PMX_AT_RAW
    W.status = 0;
    W.cig_next = 0;
    W.dp_req_base = nullptr; W.dp_res = nullptr; W.dp_slot_ctr = nullptr;
    W.dp_slot = -1; W.dp_slot_cap = 0; W.dp_n_cached = 0; W.dp_calls = 0; W.dp_post_end = 0; W.status_pre = 0;
    W.last_dp_shortcut = 0; W.skip_shortcut = 0;
    W.dp_run_calls = 0; W.dp_run_cells = 0;
    W.sk_lds_x = nullptr; W.sk_lds_y = nullptr;
}

// fixed-size output record (== pmx_aln_record in include/panmap_amd.h)
struct AlnRecord {
    int32_t rs, re, qs, qe;
    uint8_t mapq, rev, proper_frag, mapped;
    uint16_t n_cigar;
    uint16_t flags;
    uint32_t cigar_off;
    int32_t score;
};
#define PMX_REC_OVERFLOW 0x1
#define PMX_REC_UNSUPPORTED 0x2
#define PMX_REC_HAS_ALN 0x4

// count_read_errors (src/mm_align.c:122-133): edit distance of the segment's first region (block length - matches +
// ambiguous bases), or the read length when it has none
PMX_HD int32_t read_errors(const Work& W, int s) {
    if (W.n_regs[s] > 0) {
        const Reg& g = W.regs[s][0];
        if (g.has_p && g.blen > 0) return g.blen - g.mlen + (int32_t)g.n_ambi;
    }
    return W.qlen[s];
}

// extract_align_result + the mapped test of align_worker_func (src/mm_align.c:271-354)
PMX_HD bool frag_is_mapped(const Work& W, int paired) {
    if (paired) return W.n_regs[0] > 0 && W.n_regs[1] > 0 && W.regs[0][0].score > 0 && W.regs[1][0].score > 0;
    return W.n_regs[0] > 0 && W.regs[0][0].score > 0 && W.regs[0][0].score <= W.qlen[0];
}

}  // namespace aln
}  // namespace pmx
