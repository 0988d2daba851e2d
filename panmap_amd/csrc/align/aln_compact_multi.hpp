// ALIGN stage, COMPACT tier, several regions per mate: the region bookkeeping between the chains and the record for a pair
// whose fragment chains do not split into exactly one region per mate -- mates that overlap on the reference (amplicon
// reads: a fragment no longer than a read) chain into two or three fragment chains that share a mate.  Everything here is
// the logic of aln_hit.hpp / aln_map.hpp (mm_gen_regs, mm_set_parent, mm_select_sub_multi, mm_seg_gen, mm_filter_regs,
// mm_hit_sort, mm_select_sub, mm_set_sam_pri, mm_set_mapq, mm_pair: hit.c:54-94, 132-191, 193-237, 239-285, 301-322,
// 345-400, 421-466, pe.c:6-43, 76-177) restated on a plain 24-word region record and lists of at most PMX_CM_MAXC entries
// held in the thread's private memory; a pair is followed only as long as every region is aligned by the closed forms
// (c_align1), so there are no Z-drop splits, inversions or ALT contigs, one reference sequence (rid == 0), rep_len == 0.
// included by aln_compact.hpp (needs CList, CReg, c_align1, c_reg_set_coor)
#pragma once


namespace pmx {
namespace aln {

// The region record.  On the device the records live in a per-wave workspace in HBM, STRIDED like Reg in the thread-per-pair
// kernel (aln_types.hpp): every 4-byte field is followed by 252 bytes that belong to the other lanes, so that a wave touching
// one field of its 64 records touches 256 contiguous bytes while the code keeps plain struct syntax (copies are field-wise).
// (As private arrays the records were 1.7 KB of scratch per lane: a dispatch of that size makes the runtime allocate and
// release the scratch around every launch, which serialised the two batches in flight -- bench `value` 320 -> 215 M reads/s.)
#if defined(__HIP_DEVICE_COMPILE__)
#define PMX_CM_STRIDE 64
#define PMX_SP(n) char pad_##n[252];
#else
#define PMX_CM_STRIDE 1
#define PMX_SP(n)
#endif
#define PMX_SREG_FIELDS(X)                                                                                                          \
    X(id) X(cnt) X(score) X(qs) X(qe) X(rs) X(re) X(parent) X(subsc) X(as) X(mlen) X(blen) X(n_sub) X(score0) X(hash) X(mapq) X(rev) \
    X(sam_pri) X(proper_frag) X(has_p) X(dp_score) X(dp_max) X(dp_max2) X(m_len)
struct SReg {
    int32_t id; PMX_SP(0) int32_t cnt; PMX_SP(1) int32_t score; PMX_SP(2) int32_t qs; PMX_SP(3) int32_t qe; PMX_SP(4) int32_t rs; PMX_SP(5)
    int32_t re; PMX_SP(6) int32_t parent; PMX_SP(7) int32_t subsc; PMX_SP(8) int32_t as; PMX_SP(9) int32_t mlen; PMX_SP(10) int32_t blen; PMX_SP(11)
    int32_t n_sub; PMX_SP(12) int32_t score0; PMX_SP(13) uint32_t hash; PMX_SP(14) int32_t mapq; PMX_SP(15) int32_t rev; PMX_SP(16)
    int32_t sam_pri; PMX_SP(17) int32_t proper_frag; PMX_SP(18) int32_t has_p; PMX_SP(19) int32_t dp_score; PMX_SP(20) int32_t dp_max; PMX_SP(21)
    int32_t dp_max2; PMX_SP(22) int32_t m_len; PMX_SP(23)
#if defined(__HIP_DEVICE_COMPILE__)
    SReg() = default;
#define PMX_X(f) f = o.f;
    __device__ __forceinline__ SReg(const SReg& o) { PMX_SREG_FIELDS(PMX_X) }
    __device__ __forceinline__ SReg& operator=(const SReg& o) { PMX_SREG_FIELDS(PMX_X) return *this; }
#undef PMX_X
#endif
};
#if defined(__HIP_DEVICE_COMPILE__)
static_assert(sizeof(SReg) == 24 * 256, "strided SReg layout");
#else
static_assert(sizeof(SReg) == 24 * 4, "SReg has 24 four-byte fields");
#endif
// a pair's workspace: the records and a few short index lists (word i of the lane at ints[i * PMX_CM_STRIDE])
struct SWork {
    SReg* regs;
    uint32_t* ints;
    PMX_HD uint32_t& I(int i) const { return ints[i * PMX_CM_STRIDE]; }
};
// places in SWork::ints
#define PMX_CMI_PRI 0        // [4] set_parent: the primaries so far
#define PMX_CMI_NOW 4        // [4] sync_regs: id -> position
#define PMX_CMI_HI 8         // [16] pair_hits: the pair scores
#define PMX_CMI_ORD 24       // [4] chains by position
#define PMX_CMI_KEY 28       // [8] order keys (low, high)
#define PMX_CMI_KEEP 36      // [8] member sets of the fragment regions (low, high)
#define PMX_CMI_AS 44        // [4] per-mate chain lists: offset
#define PMX_CMI_CNT 48       // [4] anchors

PMX_HD uint32_t c_wang_hash(uint32_t key) {   // __ac_Wang_hash (khash.h)
    key += ~(key << 15);
    key ^= (key >> 10);
    key += (key << 3);
    key ^= (key >> 6);
    key += ~(key << 11);
    key ^= (key >> 16);
    return key;
}

PMX_HD void s_clear(SReg& r) {
#define PMX_X(f) r.f = 0;
    PMX_SREG_FIELDS(PMX_X)
#undef PMX_X
    r.parent = PMX_PARENT_UNSET;
}

// the part of [qs, qe) no primary covers (see uncovered_by_primaries, aln_hit.hpp)
PMX_HD int32_t s_uncovered(const SWork& W, const SReg* r, int n_pri, int32_t qs, int32_t qe) {
    int32_t reach = qs, uncovered = 0;
    int64_t last = -1;
    for (;;) {
        int64_t next = INT64_MAX;
        for (int j = 0; j < n_pri; ++j) {
            const SReg& p = r[W.I(PMX_CMI_PRI + j)];
            if (p.qe <= qs || p.qs >= qe) continue;
            const int64_t v = (int64_t)(p.qs > qs ? p.qs : qs) << 32 | (uint32_t)(p.qe < qe ? p.qe : qe);
            if (v > last && v < next) next = v;
        }
        if (next == INT64_MAX) break;
        const int32_t s = (int32_t)(next >> 32), e = (int32_t)next;
        if (s > reach) uncovered += s - reach;
        reach = e > reach ? e : reach;
        last = next;
    }
    if (qe > reach) uncovered += qe - reach;
    return uncovered;
}

// mm_set_parent (hit.c:132-191)
PMX_HD void s_set_parent(const SWork& W, float mask_level, int mask_len, int n, SReg* r, int sub_diff) {
    if (n <= 0) return;
    for (int i = 0; i < n; ++i) r[i].id = i;
    int n_pri = 1;
    W.I(PMX_CMI_PRI) = 0;
    r[0].parent = 0;
    for (int i = 1; i < n; ++i) {
        SReg& me = r[i];
        const int32_t qs = me.qs, qe = me.qe, my_len = qe - qs;
        bool touches = false;
        for (int j = 0; j < n_pri && !touches; ++j) { const SReg& p = r[W.I(PMX_CMI_PRI + j)]; touches = !(p.qe <= qs || p.qs >= qe); }
        int owner = -1;
        if (touches) {
            const int32_t uncov = s_uncovered(W, r, n_pri, qs, qe);
            for (int j = 0; j < n_pri; ++j) {
                const SReg& p = r[W.I(PMX_CMI_PRI + j)];
                if (p.qe <= qs || p.qs >= qe) continue;
                const int32_t p_len = p.qe - p.qs;
                const int32_t shorter = p_len < my_len ? p_len : my_len, longer = p_len > my_len ? p_len : my_len;
                const int32_t lo = qs > p.qs ? qs : p.qs, hi = qe < p.qe ? qe : p.qe;
                const int32_t ol = hi > lo ? hi - lo : 0;
                if ((float)ol / shorter - (float)uncov / longer > mask_level && uncov <= mask_len) { owner = j; break; }
            }
        }
        if (owner < 0) {
            W.I(PMX_CMI_PRI + n_pri++) = (uint32_t)i;
            me.parent = i;
            me.n_sub = 0;
            continue;
        }
        SReg& p = r[W.I(PMX_CMI_PRI + owner)];
        const int32_t p_len = p.qe - p.qs;
        const int32_t shorter = p_len < my_len ? p_len : my_len;
        const int32_t lo = qs > p.qs ? qs : p.qs, hi = qe < p.qe ? qe : p.qe;
        const int32_t ol = hi > lo ? hi - lo : 0;
        bool rival = me.cnt >= p.cnt;
        me.parent = p.parent;
        if (p.subsc < me.score) p.subsc = me.score;
        if (p.has_p && me.has_p && (p.rs != me.rs || p.re != me.re || ol != shorter)) {
            if (p.dp_max2 < me.dp_max) p.dp_max2 = me.dp_max;
            rival = rival || p.dp_max - me.dp_max <= sub_diff;
        }
        if (rival) ++p.n_sub;
    }
}

// mm_set_sam_pri (hit.c:227-237)
PMX_HD void s_set_sam_pri(int n, SReg* r) {
    int n_pri = 0;
    for (int i = 0; i < n; ++i) {
        const bool is_pri = r[i].id == r[i].parent;
        n_pri += is_pri ? 1 : 0;
        r[i].sam_pri = is_pri && n_pri == 1;
    }
}

// mm_sync_regs (hit.c:239-262); ids of a list never exceed its original length
PMX_HD void s_sync_regs(const SWork& W, int n, SReg* r) {
    if (n <= 0) return;
    for (int i = 0; i < PMX_CM_MAXC; ++i) W.I(PMX_CMI_NOW + i) = 0xffffffffu;
    for (int i = 0; i < n; ++i)
        if (r[i].id >= 0 && r[i].id < PMX_CM_MAXC) W.I(PMX_CMI_NOW + r[i].id) = (uint32_t)i;
    for (int i = 0; i < n; ++i) {
        const int32_t par = r[i].parent;
        r[i].id = i;
        r[i].parent = (par >= 0 && par < PMX_CM_MAXC && (int32_t)W.I(PMX_CMI_NOW + par) >= 0) ? (int32_t)W.I(PMX_CMI_NOW + par) : PMX_PARENT_UNSET;
    }
    s_set_sam_pri(n, r);
}

// one in-place thinning pass (thin_regs, aln_hit.hpp: keep(i) sees the array as it is when i is reached)
template <class Keep>
PMX_HD int s_thin(const SWork& W, int n, SReg* r, Keep keep) {
    int k = 0;
    for (int i = 0; i < n; ++i) {
        if (!keep(i)) continue;
        if (k != i) r[k] = r[i];
        ++k;
    }
    if (k != n) s_sync_regs(W, k, r);
    return k;
}

// mm_select_sub (hit.c:264-285) with check_strand == 0
PMX_HD int s_select_sub(const SWork& W, float pri_ratio, int min_diff, int best_n, int n, SReg* r) {
    if (!(pri_ratio > 0.0f) || n <= 0) return n;
    int n_2nd = 0;
    return s_thin(W, n, r, [&](int i) {
        const SReg& me = r[i];
        const int p = me.parent;
        if (p == i) return true;
        const SReg& par = r[p];
        if ((me.score >= par.score * pri_ratio || me.score + min_diff >= par.score) && n_2nd < best_n) {
            if (me.qs == par.qs && me.qe == par.qe && me.rs == par.rs && me.re == par.re) return false;
            ++n_2nd;
            return true;
        }
        return false;
    });
}

// mm_select_sub_multi (pe.c:6-43) for two segments
PMX_HD int s_select_sub_multi(const SWork& W, float pri_ratio, float pri1, float pri2, int max_gap_ref, int min_diff, int best_n, int qlen0, int qlen1, int n, SReg* r) {
    if (!(pri_ratio > 0.0f) || n <= 0) return n;
    const int max_dist = qlen0 + qlen1 + max_gap_ref;
    const int32_t mate_border = qlen0;
    int n_2nd = 0;
    return s_thin(W, n, r, [&](int i) {
        const SReg& q = r[i];
        if (q.parent == i) return true;
        const SReg& p = r[q.parent];
        bool worth;
        if (q.score + min_diff >= p.score) worth = true;
        else if (p.rev == q.rev && q.re - p.rs < max_dist && p.re - q.rs < max_dist) worth = q.score >= p.score * pri1;
        else {
            const bool p_both = p.qs < mate_border && p.qe > mate_border;
            const bool q_both = q.qs < mate_border && q.qe > mate_border;
            worth = q.score >= p.score * ((q_both || q_both == p_both) ? pri_ratio : pri2);
        }
        if (worth && n_2nd++ >= best_n) worth = false;
        return worth;
    });
}

// mm_filter_regs (hit.c:301-322) for segment splits (the min_cnt test does not apply)
PMX_HD int s_filter_regs(const Opt& o, int qlen, int n, SReg* r) {
    const float clip = qlen * o.max_clip_ratio;
    int k = 0;
    for (int i = 0; i < n; ++i) {
        const SReg& g = r[i];
        bool drop = false;
        if (g.has_p) drop = g.mlen < o.min_chain_score || g.dp_max < o.min_dp_max || (g.qs > clip && qlen - g.qe > clip);
        if (drop) continue;
        if (k != i) r[k] = r[i];
        ++k;
    }
    return k;
}

// mm_hit_sort (hit.c:193-225)
PMX_HD int s_hit_sort(const SWork& W, int n, SReg* r) {
    if (n <= 1) return n;
    SReg* t = W.regs + 3 * PMX_CM_MAXC;
    auto listed = [&](int i) { return r[i].cnt > 0; };
    auto key_of = [&](int i) { return (uint64_t)(int64_t)(r[i].has_p ? r[i].dp_max : r[i].score) << 32 | r[i].hash; };
    int n_out = 0;
    for (int i = 0; i < n; ++i) {
        if (!listed(i)) continue;
        const uint64_t ki = key_of(i);
        int at = 0;
        for (int j = 0; j < n; ++j) {
            if (!listed(j)) continue;
            const uint64_t kj = key_of(j);
            at += (kj > ki || (kj == ki && j > i)) ? 1 : 0;
        }
        t[at] = r[i];
        ++n_out;
    }
    for (int i = 0; i < n_out; ++i) r[i] = t[i];
    return n_out;
}

// mm_set_mapq (hit.c:421-466) with rep_len == 0, is_sr == 0, no inversions.  false: a value outside the host's logf tables
PMX_HD bool s_set_mapq(const RefIndex& ri, int n, SReg* regs, int min_chain_sc, int match_sc) {
    if (n == 0) return true;
    int64_t pri_score_sum = 0;
    for (int i = 0; i < n; ++i) pri_score_sum += regs[i].parent == regs[i].id ? regs[i].score : 0;
    const float uniq_ratio = (float)pri_score_sum / (float)(pri_score_sum + 0);
    const float q_coef = 40.0f;
    for (int i = 0; i < n; ++i) {
        SReg& r = regs[i];
        if (r.parent != r.id) { r.mapq = 0; continue; }
        const bool tables_ok = !(r.has_p && (r.dp_max < 0 || r.dp_max >= ri.n_logf)) && r.score >= 0 && r.score < ri.n_logf && r.n_sub + 1 < ri.n_logf;
        if (!tables_ok) return false;
        const float pen_s1 = (r.score > 100 ? 1.0f : 0.01f * r.score) * uniq_ratio;
        float pen_cm = r.cnt > 10 ? 1.0f : 0.1f * r.cnt;
        pen_cm = pen_s1 < pen_cm ? pen_s1 : pen_cm;
        const int subsc = r.subsc > min_chain_sc ? r.subsc : min_chain_sc;
        int mapq;
        if (r.has_p && r.dp_max2 > 0 && r.dp_max > 0) {
            const float identity = (float)r.mlen / r.blen;
            const float x = (float)r.dp_max2 * subsc / r.dp_max / r.score0;
            mapq = (int)(identity * pen_cm * q_coef * (1.0f - x * x) * ri.logf_ratio[r.dp_max]);
            const int mapq_alt = (int)(6.02f * identity * identity * (r.dp_max - r.dp_max2) / match_sc + .499f);
            mapq = mapq < mapq_alt ? mapq : mapq_alt;
        } else {
            const float x = (float)subsc / r.score0;
            if (r.has_p) {
                const float identity = (float)r.mlen / r.blen;
                mapq = (int)(identity * pen_cm * q_coef * (1.0f - x) * ri.logf_ratio[r.dp_max]);
            } else mapq = (int)(pen_cm * q_coef * (1.0f - x) * ri.logf_int[r.score]);
        }
        mapq -= (int)(4.343f * ri.logf_int[r.n_sub + 1] + .499f);
        mapq = mapq > 0 ? mapq : 0;
        r.mapq = mapq < 60 ? mapq : 60;
        if (r.has_p && r.dp_max > r.dp_max2 && r.mapq == 0) r.mapq = 1;
    }
    return true;
}

// mm_pair (pe.c:76-177): see pair_hits (aln_map.hpp) for the sweep by selection.  false: outside the logf tables
PMX_HD bool s_pair_hits(const SWork& W, const RefIndex& ri, int max_gap_ref, int pe_bonus, int sub_diff, int match_sc, int n0, SReg* m0, int n1, SReg* m1) {
    const int n_ends = n0 + n1;
    if (n0 == 0 || n1 == 0) return true;
    auto mate_of = [&](int e) { return e >= n0 ? 1 : 0; };
    auto reg_of = [&](int e) -> SReg& { return e >= n0 ? m1[e - n0] : m0[e]; };
    auto key_of = [&](int e) {
        const SReg& r = reg_of(e);
        return (uint64_t)(uint32_t)(r.rs << 1) | (uint32_t)(mate_of(e) ^ r.rev);
    };
    int floor_dp = -pe_bonus;
    {
        int top = 0;
        for (int i = 0; i < n0; ++i) top = top > m0[i].dp_max ? top : m0[i].dp_max;
        floor_dp += top;
        top = 0;
        for (int i = 0; i < n1; ++i) top = top > m1[i].dp_max ? top : m1[i].dp_max;
        floor_dp += top;
    }
    if (floor_dp < 0) floor_dp = 0;
    int n_pairs = 0;
    int64_t best = -1, second = -1;
    int best_end[2] = {-1, -1};
    auto next_end = [&](bool started, uint64_t k_prev, int e_prev, uint32_t closing) {
        int pick = -1;
        uint64_t k_pick = 0;
        for (int e = 0; e < n_ends; ++e) {
            const uint64_t k = key_of(e);
            if ((uint32_t)(k & 1) != closing) continue;
            if (started && (k < k_prev || (k == k_prev && e <= e_prev))) continue;
            if (pick < 0 || k < k_pick) { pick = e; k_pick = k; }
        }
        return pick;
    };
    auto opening_before = [&](uint64_t k_lim, int e_lim, int rev, int want) {
        int pick = -1;
        uint64_t k_pick = 0;
        for (int e = 0; e < n_ends; ++e) {
            const uint64_t k = key_of(e);
            if ((k & 1) || (int)reg_of(e).rev != rev || (want >= 0 && mate_of(e) != want)) continue;
            if (k > k_lim || (k == k_lim && e >= e_lim)) continue;
            if (pick < 0 || k > k_pick || (k == k_pick && e > pick)) { pick = e; k_pick = k; }
        }
        return pick;
    };
    bool started = false;
    uint64_t k_cur = 0;
    int e_cur = -1;
    for (;;) {
        const int c = next_end(started, k_cur, e_cur, 1u);
        if (c < 0) break;
        started = true; k_cur = key_of(c); e_cur = c;
        const SReg& rc = reg_of(c);
        const int near = opening_before(k_cur, c, rc.rev, -1);
        if (near < 0) continue;
        { const SReg& q = reg_of(near); if (rc.rs - q.re > max_gap_ref) continue; }
        uint64_t k_lim = k_cur;
        int e_lim = c;
        for (;;) {
            const int p = opening_before(k_lim, e_lim, rc.rev, 1 - mate_of(c));
            if (p < 0) break;
            k_lim = key_of(p); e_lim = p;
            const SReg& q = reg_of(p);
            if (rc.rs - q.re > max_gap_ref) break;
            if (rc.dp_max + q.dp_max < floor_dp) continue;
            const int64_t score = (int64_t)(rc.dp_max + q.dp_max) << 32 | (uint32_t)(rc.hash + q.hash);
            if (score > best) { second = best; best = score; best_end[mate_of(p)] = p; best_end[mate_of(c)] = c; }
            else if (score > second) second = score;
            if (n_pairs < PMX_CM_MAXC * PMX_CM_MAXC) W.I(PMX_CMI_HI + n_pairs) = (uint32_t)(rc.dp_max + q.dp_max);
            ++n_pairs;
        }
    }
    if (n_pairs > PMX_CM_MAXC * PMX_CM_MAXC) return false;   // (cannot happen: every pair of ends is found at most once)
    if (n_pairs > 0 && best > 0) {
        SReg* won[2] = {&reg_of(best_end[0]), &reg_of(best_end[1])};
        won[0]->proper_frag = won[1]->proper_frag = 1;
        for (int m = 0; m < 2; ++m) {
            SReg* all = m ? m1 : m0;
            const int nm = m ? n1 : n0;
            SReg* w = won[m];
            if (w->id != w->parent) {
                SReg& old_pri = all[w->parent];
                const int old_id = old_pri.id;
                for (int i = 0; i < nm; ++i)
                    if (all[i].parent == old_id) all[i].parent = w->id;
                old_pri.mapq = 0;
            }
            if (!w->sam_pri) {
                for (int i = 0; i < nm; ++i) all[i].sam_pri = 0;
                w->sam_pri = 1;
            }
        }
        const uint32_t best_hi = (uint32_t)((uint64_t)best >> 32);
        int pe_q = won[0]->mapq > won[1]->mapq ? won[0]->mapq : won[1]->mapq;
        if (n_pairs > 1) {
            int close = 0;
            for (int i = 0; i < n_pairs; ++i) close += (uint64_t)W.I(PMX_CMI_HI + i) + (uint64_t)sub_diff >= (uint64_t)best_hi;
            if (close >= ri.n_logf) return false;
            const int by_margin = (int)(6.02f * (float)((best >> 32) - (second >> 32)) / match_sc - 4.343f * ri.logf_int[close]);
            pe_q = pe_q < by_margin ? pe_q : by_margin;
        }
        const int at_least = n_pairs == 1 ? 2 : (best_hi > (uint32_t)((uint64_t)second >> 32) ? 1 : 0);
        for (int m = 0; m < 2; ++m) {
            SReg* w = won[m];
            if (w->mapq < pe_q) w->mapq = (int)(uint8_t)(int)(.2f * w->mapq + .8f * pe_q + .499f);
            if (w->mapq < at_least) w->mapq = at_least;
        }
    }
    return true;
}

// Place of entry i when n keys are listed in descending order, later entries first among equals (rank_desc_later_first)
PMX_HD uint64_t s_key(const SWork& W, int i) { return (uint64_t)W.I(PMX_CMI_KEY + 2 * i) | (uint64_t)W.I(PMX_CMI_KEY + 2 * i + 1) << 32; }
PMX_HD void s_set_key(const SWork& W, int i, uint64_t v) { W.I(PMX_CMI_KEY + 2 * i) = (uint32_t)v; W.I(PMX_CMI_KEY + 2 * i + 1) = (uint32_t)(v >> 32); }
PMX_HD int s_rank(const SWork& W, int n, int i) {
    const uint64_t ki = s_key(W, i);
    int before = 0;
    for (int j = 0; j < n; ++j) { const uint64_t kj = s_key(W, j); before += (kj > ki || (kj == ki && j > i)) ? 1 : 0; }
    return before;
}

// From the fragment chains of a pair (n_u of them in the order the backtrack found them, in the workspace: score at
// PMX_CMI_USC + c, the members as a set of anchor indices at PMX_CMI_UKEEP + 2c) to the pair's result.  The anchors are in
// m.X / m.Y; m.G is free and receives the per-mate anchor lists.
#define PMX_CMI_USC 52       // [4] chains as found: score
#define PMX_CMI_UKEEP 56     // [8] member set (low, high)
static_assert(PMX_CMI_UKEEP + 2 * PMX_CM_MAXC <= PMX_CM_INTS, "workspace index words");
PMX_HD uint64_t s_get64(const SWork& W, int at) { return (uint64_t)W.I(at) | (uint64_t)W.I(at + 1) << 32; }
PMX_HD void s_set64(const SWork& W, int at, uint64_t v) { W.I(at) = (uint32_t)v; W.I(at + 1) = (uint32_t)(v >> 32); }

template <class PT, int CAP>
PMX_HD int compact_regions_multi(const SWork& W, const CMemT<PT, CAP>& m, const Opt& o, const RefIndex& ri, const CRead* rd, int n_u, int max_chain_gap_ref,
                                 CResult& out, bool want_edits) {
    typedef CMemT<PT, CAP> MT;
    const int k = o.k;
    const int qlen0 = rd[0].len, qlen1 = rd[1].len, qlen_sum = qlen0 + qlen1;
    const uint32_t hash = c_wang_hash(c_wang_hash((uint32_t)qlen_sum) + c_wang_hash((uint32_t)o.seed));
    // the 64-bit anchor words of the reference (mmpriv.h:17-23) of anchor ai, its query position lowered by `shift`
    auto anchor_x = [&](int ai) { return MT::x64(m.X(ai)); };
    auto anchor_y = [&](int ai, int shift) {
        const uint32_t y = m.Y(ai);
        return (uint64_t)k << 32 | (uint64_t)((y >> 10) & 1u) << PMX_SEED_SEG_SHIFT | ((y & PMX_CQ_TANDEM) ? PMX_SEED_TANDEM : 0ULL) |
               (uint64_t)(uint32_t)((int32_t)(y & 0x3ffu) - shift);
    };
    auto tie_of = [&](int ai, int shift) { return (uint32_t)hit_hash64((hit_hash64(anchor_x(ai)) + hit_hash64(anchor_y(ai, shift))) ^ hash); };
    auto u_keep = [&](int c) { return s_get64(W, PMX_CMI_UKEEP + 2 * c); };

    // ---- the chains in the order mm_chain_dp leaves them: by the position word of their first anchor (lchain.c:78-111;
    //      its sorter is stable for a handful of entries)
    for (int i = 0; i < n_u; ++i) {
        const uint64_t xi = anchor_x(__builtin_ctzll(u_keep(i)));
        int at = 0;
        for (int j = 0; j < n_u; ++j) {
            const uint64_t xj = anchor_x(__builtin_ctzll(u_keep(j)));
            at += (xj < xi || (xj == xi && j < i)) ? 1 : 0;
        }
        W.I(PMX_CMI_ORD + at) = (uint32_t)i;
    }
    // ---- mm_gen_regs on the fragment chains
    SReg* f = W.regs;
    for (int i = 0; i < n_u; ++i) {
        const int c = (int)W.I(PMX_CMI_ORD + i);
        const uint64_t kp = u_keep(c);
        s_set_key(W, i, ((uint64_t)W.I(PMX_CMI_USC + c) << 32 | (uint32_t)__builtin_popcountll(kp)) ^ tie_of(__builtin_ctzll(kp), 0));
    }
    for (int i = 0; i < n_u; ++i) {
        const int c = (int)W.I(PMX_CMI_ORD + i);
        const uint64_t kp = u_keep(c), key = s_key(W, i);
        const int at = s_rank(W, n_u, i);
        SReg& g = f[at];
        s_clear(g);
        g.id = at;
        g.score = g.score0 = (int32_t)(key >> 32);
        g.hash = (uint32_t)key;
        g.cnt = __builtin_popcountll(kp);
        s_set64(W, PMX_CMI_KEEP + 2 * at, kp);
        const int head = __builtin_ctzll(kp), tail = 63 - __builtin_clzll(kp);
        const int32_t q_lo = (int32_t)(m.Y(head) & 0x3ffu) + 1 - k, q_hi = (int32_t)(m.Y(tail) & 0x3ffu) + 1;
        const int32_t rp_h = (int32_t)MT::pos_of(m.X(head)), rp_t = (int32_t)MT::pos_of(m.X(tail));
        const int32_t rev = (int32_t)MT::rev_of(m.X(head));
        g.rev = rev;
        g.rs = rp_h + 1 > k ? rp_h + 1 - k : 0;
        g.re = rp_t + 1;
        g.qs = rev ? qlen_sum - q_hi : q_lo;
        g.qe = rev ? qlen_sum - q_lo : q_hi;
    }
    int n_f = n_u;
    // ---- chain_post (map.c:206-213)
    s_set_parent(W, o.mask_level, o.mask_len, n_f, f, o.a * 2 + o.b);
    for (int i = 0; i < n_f; ++i) f[i].as = i;   // (the thinning moves records: the member sets follow through the records' `as` field)
    n_f = s_select_sub_multi(W, o.pri_ratio, 0.2f, 0.7f, max_chain_gap_ref, o.k * 2, o.best_n, qlen0, qlen1, n_f, f);

    // ---- mm_seg_gen (hit.c:345-400): per mate, the chains that touch it (fragment order) and their anchors on it
    int n_r[2] = {0, 0}, base[2] = {0, 0}, n_a[2] = {0, 0};
    {
        int wr = 0;
        for (int s = 0; s < 2; ++s) {
            SReg* R = W.regs + (1 + s) * PMX_CM_MAXC;
            const int base_s = wr;
            int nu = 0;
            for (int i = 0; i < n_f; ++i) {
                const uint64_t kp = s_get64(W, PMX_CMI_KEEP + 2 * f[i].as);
                int on = 0, first = -1;
                const int as0 = wr - base_s;
                for (uint64_t wk = kp; wk; wk &= wk - 1) {
                    const int ai = __builtin_ctzll(wk);
                    if ((int)((m.Y(ai) >> 10) & 1u) != s) continue;
                    if (first < 0) first = ai;
                    m.G(wr++) = (c_u16)ai;
                    ++on;
                }
                if (!on) continue;
                // the mate's own query coordinates (mirrored on the reverse strand, hit.c:381)
                const int32_t rev = (int32_t)MT::rev_of(m.X(first));
                const int32_t my_len = s ? qlen1 : qlen0, before = s ? qlen0 : 0;
                const int shift = rev ? qlen_sum - (my_len + before) : before;
                s_set_key(W, nu, ((uint64_t)(uint32_t)f[i].score << 32 | (uint32_t)on) ^ tie_of(first, shift));
                W.I(PMX_CMI_AS + nu) = (uint32_t)as0;
                W.I(PMX_CMI_CNT + nu) = (uint32_t)on;
                ++nu;
            }
            for (int i = 0; i < nu; ++i) {
                const int at = s_rank(W, nu, i);
                const uint64_t key = s_key(W, i);
                SReg& g = R[at];
                s_clear(g);
                g.id = at;
                g.score = g.score0 = (int32_t)(key >> 32);
                g.hash = (uint32_t)key;
                g.cnt = (int32_t)W.I(PMX_CMI_CNT + i);
                g.as = (int32_t)W.I(PMX_CMI_AS + i);
            }
            if (s == 0) { base[0] = base_s; n_a[0] = wr - base_s; n_r[0] = nu; }
            else { base[1] = base_s; n_a[1] = wr - base_s; n_r[1] = nu; }
        }
    }
    if (n_r[0] == 0 || n_r[1] == 0) return want_edits ? PMX_C_BAIL : PMX_C_DONE;   // a mate without a region: unmapped (its partner's edit count: general tier)

    // ---- per mate: coordinates, set_parent, alignment of every region, filter, sort, parents, secondaries, mapq
    for (int s = 0; s < 2; ++s) {
        const int qlen = s ? qlen1 : qlen0;
        const int base_s = s ? base[1] : base[0], n_a_s = s ? n_a[1] : n_a[0];
        SReg* r = W.regs + (1 + s) * PMX_CM_MAXC;
        int nr = s ? n_r[1] : n_r[0];
        if (qlen >= o.rank_min_len) return PMX_C_BAIL;
        const int32_t before = s ? qlen0 : 0;
        for (int i = 0; i < nr; ++i) {
            SReg& g = r[i];
            const int as = g.as;
            const int32_t rev = (int32_t)MT::rev_of(m.X((int)m.G(base_s + as)));
            const CList<MT> L{m, base_s + as, rev ? qlen_sum - (qlen + before) : before};
            CReg c;
            c.cnt = g.cnt; c.rev = rev;
            c_reg_set_coor(L, c, qlen, k);
            g.rev = rev;
            g.rs = c.rs; g.re = c.re; g.qs = c.qs; g.qe = c.qe; g.mlen = c.mlen; g.blen = c.blen;
        }
        s_set_parent(W, o.mask_level, o.mask_len, nr, r, o.a * 2 + o.b);
        for (int i = 0; i < nr; ++i) {
            SReg& g = r[i];
            const int as = g.as;
            CReg c;
            c.cnt = g.cnt; c.score = g.score; c.rev = g.rev; c.qs = g.qs; c.qe = g.qe; c.rs = g.rs; c.re = g.re; c.mlen = g.mlen; c.blen = g.blen;
            c.has_p = 0; c.dp_score = c.dp_max = 0; c.mapq = 0; c.proper_frag = 0; c.m_len = 0;
            const int shift = c.rev ? qlen_sum - (qlen + before) : before;
            const CList<MT> L{m, base_s + as, shift};
            const CList<MT> Lm{m, base_s, shift};   // (the fence scan only visits anchors of the region's strand: same shift)
            if (c_align1<true>(L, o, ri, rd[s], qlen, c, &Lm, as, n_a_s) != PMX_C_DONE) return PMX_C_BAIL;
            g.rs = c.rs; g.re = c.re; g.qs = c.qs; g.qe = c.qe; g.mlen = c.mlen; g.blen = c.blen;
            g.has_p = c.has_p; g.dp_score = c.dp_score; g.dp_max = c.dp_max; g.m_len = c.m_len;
        }
        nr = s_filter_regs(o, qlen, nr, r);
        nr = s_hit_sort(W, nr, r);
        s_set_parent(W, o.mask_level, o.mask_len, nr, r, o.a * 2 + o.b);
        nr = s_select_sub(W, o.pri_ratio, o.k * 2, o.best_n, nr, r);
        s_set_sam_pri(nr, r);
        if (!s_set_mapq(ri, nr, r, o.min_chain_score, o.a)) return PMX_C_BAIL;
        if (s == 0) n_r[0] = nr; else n_r[1] = nr;
    }
    SReg* R0 = W.regs + PMX_CM_MAXC;
    SReg* R1 = W.regs + 2 * PMX_CM_MAXC;
    if (o.pe_ori >= 0 && !s_pair_hits(W, ri, max_chain_gap_ref, o.pe_bonus, o.a * 2 + o.b, o.a, n_r[0], R0, n_r[1], R1)) return PMX_C_BAIL;

    // ---- the record (src/mm_align.c:271-354): the first region of each mate
    if (n_r[0] > 0 && R0[0].has_p && R0[0].blen > 0) out.edit[0] = R0[0].blen - R0[0].mlen;
    if (n_r[1] > 0 && R1[0].has_p && R1[0].blen > 0) out.edit[1] = R1[0].blen - R1[0].mlen;
    if (!(n_r[0] > 0 && n_r[1] > 0 && R0[0].score > 0 && R1[0].score > 0)) return PMX_C_DONE;   // unmapped pair (every region was aligned: the edit counts stand)
    out.mapped = 1;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const SReg& r = s ? R1[0] : R0[0];
        CMate& t = out.m[s];
        t.rs = r.rs; t.re = r.re; t.qs = r.qs; t.qe = r.qe;
        t.dp_max = r.dp_max;
        t.cigar = (uint32_t)r.m_len << 4;
        t.mapq = (uint8_t)r.mapq; t.rev = (uint8_t)r.rev; t.proper_frag = (uint8_t)r.proper_frag; t.has_aln = 1;
    }
    return PMX_C_DONE;
}

}  // namespace aln
}  // namespace pmx
