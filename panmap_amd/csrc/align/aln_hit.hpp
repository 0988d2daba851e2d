// ALIGN stage, part 3: what happens to chains once they exist -- they become regions, regions get ordered, related to
// each other (primary / secondary), thinned, split per mate, and finally carry a mapping quality.
//
// The reference does all of this on arrays of mm_reg1_t with its radix sorter and in-place loops (hit.c:8-466,
// pe.c:6-43).  What has to be identical is every DECISION (which region comes first, which is a secondary of which,
// which survives, the float expressions of the mapping quality) -- not the procedure.  Here the orderings are computed
// as RANKS (a region's place = how many others precede it; no sort scratch, no sort-then-reverse), interval coverage is
// swept by repeated selection instead of sorting clipped intervals, the anchor squeeze walks the regions in increasing
// anchor offset by selection, and the thinning passes are one generic in-place filter with the keep rule as a functor.
// Region lists are short (one to a handful per read, capped by Caps::max_reg <= 64), so the O(n^2) forms are a few
// dozen operations and need none of the per-wave scratch arrays.
//
// Order conventions that leak into results and are therefore kept (hit.c:54-94, 193-225): regions are ordered by
// (score, tie-break hash) DESCENDING; among equal keys the one that came LATER in the input stands first (the reference
// sorts ascending with a stable insertion sort for <= 64 entries and reverses).  The thinning loops read a parent
// THROUGH THE ARRAY AS IT IS at that moment (entries already compacted forward), as the reference's do.
#pragma once
#include "aln_sort.hpp"
#include "aln_types.hpp"

namespace pmx {
namespace aln {

// --------------------------------------------------------------------------------------------- small pieces
PMX_HD uint64_t hit_hash64(uint64_t key) {   // hit.c:42-52 (unmasked variant of the invertible hash)
    key = (~key + (key << 21));
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8));
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4));
    key = key ^ key >> 28;
    key = (key + (key << 31));
    return key;
}

// anchor accessors (mmpriv.h:17-23 packing)
PMX_HD int32_t anc_rpos(const A128& p) { return (int32_t)p.x; }
PMX_HD int32_t anc_qpos(const A128& p) { return (int32_t)p.y; }
PMX_HD int32_t anc_span(const A128& p) { return (int32_t)(p.y >> 32 & 0xff); }
PMX_HD int anc_seg(const A128& p) { return (int)((p.y & PMX_SEED_SEG_MASK) >> PMX_SEED_SEG_SHIFT); }

PMX_HD void reg_clear(Reg& r) {
#define PMX_X(f) r.f = 0;
    PMX_REG_FIELDS(PMX_X)
#undef PMX_X
    r.pad_b[0] = r.pad_b[1] = r.pad_b[2] = 0;
}

// Where a run of anchors [as, as + cnt) puts its region: strand, target interval, query interval (mirrored for the
// reverse strand), and the "fuzzy" matched / block lengths accumulated step by step (hit.c:8-40; is_qstrand == 0).
PMX_HD void reg_set_coor(Reg& r, int32_t qlen, Ptr<const A128> a) {
    PMX_LDS(a);
    const A128 head = a[r.as], tail = a[r.as + r.cnt - 1];
    const int32_t span0 = anc_span(head);
    const int32_t q_lo = anc_qpos(head) + 1 - span0, q_hi = anc_qpos(tail) + 1;   // on the strand the anchors are on
    r.rev = (uint8_t)(head.x >> 63);
    r.rid = (int32_t)(head.x << 1 >> 33);
    r.rs = anc_rpos(head) + 1 > span0 ? anc_rpos(head) + 1 - span0 : 0;
    r.re = anc_rpos(tail) + 1;
    r.qs = r.rev ? qlen - q_hi : q_lo;
    r.qe = r.rev ? qlen - q_lo : q_hi;
    r.mlen = r.blen = 0;
    if (r.cnt <= 0) return;
    int32_t matched = span0, block = span0;
    A128 prev = head;
    for (int i = 1; i < r.cnt; ++i) {
        const A128 cur = a[r.as + i];
        const int32_t dt = anc_rpos(cur) - anc_rpos(prev), dq = anc_qpos(cur) - anc_qpos(prev), sp = anc_span(cur);
        block += dt > dq ? dt : dq;
        matched += (dt > sp && dq > sp) ? sp : (dt < dq ? dt : dq);
        prev = cur;
    }
    r.mlen = matched;
    r.blen = block;
}

// Place of entry i when n keys are listed in descending order, later entries first among equals.
template <class KeyAt>
PMX_HD int rank_desc_later_first(int n, int i, uint64_t key_i, KeyAt key_at) {
    int before = 0;
    for (int j = 0; j < n; ++j) {
        const uint64_t kj = key_at(j);
        before += (kj > key_i || (kj == key_i && j > i)) ? 1 : 0;
    }
    return before;
}

// --------------------------------------------------------------------------------------------- chains -> regions
// mm_gen_regs (hit.c:54-94).  u[i] = chain score << 32 | anchors; the chains' anchors lie back to back in a[].
// A region's order key is its u with a hash of its first anchor (and the read's seed) XORed into the low half, so equal
// scores order pseudo-randomly but reproducibly.
PMX_HDN int gen_regs(Work& W, uint32_t hash, int qlen, int n_u, Ptr<const uint64_t> u, Ptr<const A128> a, Reg* r) {
    PMX_LDS(&W); PMX_LDS(u); PMX_LDS(a); PMX_LDS(r);
    if (n_u == 0) return 0;
    if (n_u > W.caps.max_reg) { W.status |= PMX_ST_OVERFLOW; n_u = W.caps.max_reg; }
    if (n_u > 64) { W.status |= PMX_ST_UNSUPPORTED; n_u = 64; }   // beyond the stable regime of the reference's sorter
    // order keys and anchor offsets, kept in the region slots' own `hash` / `as` / `cnt` / `score` fields until ranked:
    // the final records are written through a second pass so that nothing is overwritten before it is read
    Ptr<A128> key = W.aux128; PMX_LDS(key);   // x = order key, y = first anchor << 32 | anchors
    int first = 0;
    for (int i = 0; i < n_u; ++i) {
        const uint64_t ui = u[i];
        const A128 head = a[first];
        const uint32_t tie = (uint32_t)hit_hash64((hit_hash64(head.x) + hit_hash64(head.y)) ^ hash);
        A128 e;
        e.x = ui ^ tie;
        e.y = (uint64_t)first << 32 | (uint32_t)(int32_t)ui;
        key[i] = e;
        first += (int32_t)ui;
    }
    wave_sync();
    for (int i = 0; i < n_u; ++i) {
        const A128 e = key[i];
        const int at = rank_desc_later_first(n_u, i, e.x, [&](int j) { return key[j].x; });
        Reg& g = r[at];
        reg_clear(g);
        g.id = at;
        g.parent = PMX_PARENT_UNSET;
        g.score = g.score0 = (int32_t)(e.x >> 32);
        g.hash = (uint32_t)e.x;
        g.cnt = (int32_t)e.y;
        g.as = (int32_t)(e.y >> 32);
        g.div = -1.0f;
        reg_set_coor(g, qlen, a);
    }
    wave_sync();
    return n_u;
}

// mm_split_reg (hit.c:112-130): the tail of a region, from its n-th anchor on, becomes a region of its own (z-drop)
PMX_HD void split_reg(Reg& r, Reg& r2, int n, int qlen, Ptr<const A128> a) {
    PMX_LDS(&r); PMX_LDS(a);   // r2 is a caller-local (private) object
    if (n <= 0 || n >= r.cnt) return;
    const int tail_cnt = r.cnt - n;
    const int32_t tail_score = (int32_t)(r.score * ((float)tail_cnt / r.cnt) + .499);
    r2 = r;
    r2.id = -1;
    r2.sam_pri = 0;
    r2.has_p = 0;
    r2.n_cigar = 0; r2.dp_score = r2.dp_max = r2.dp_max2 = 0; r2.n_ambi = 0;
    r2.split_inv = 0;
    r2.cnt = tail_cnt;
    r2.score = tail_score;
    r2.as = r.as + n;
    if (r.parent == r.id) r2.parent = PMX_PARENT_TMP_PRI;
    reg_set_coor(r2, qlen, a);
    r.cnt = n;
    r.score -= tail_score;
    reg_set_coor(r, qlen, a);
    r.split |= 1;
    r2.split |= 2;
}

// --------------------------------------------------------------------------------------------- primary / secondary
// mm_set_parent (hit.c:132-191; hard_mask_level == 0, no ALT contigs).  Regions arrive best first.  A region is a
// SECONDARY of the first earlier primary that it overlaps on the query by more than mask_level of the shorter of the two,
// after discounting (relative to the longer) what no primary covers at all; otherwise it is a new primary.  The part of
// [qs, qe) no primary covers is found by sweeping the primaries' overlaps in increasing start: repeated selection of the
// next start (the reference sorts the clipped intervals; the uncovered length does not depend on how ties are ordered).
PMX_HD int32_t uncovered_by_primaries(const Reg* r, Ptr<const int32_t> pri, int n_pri, int32_t qs, int32_t qe) {
    int32_t reach = qs, uncovered = 0;
    int64_t last = -1;   // (start << 32 | end) of the interval handled last; starts and ends are non-negative
    for (;;) {
        int64_t next = INT64_MAX;
        int same = 0;
        for (int j = 0; j < n_pri; ++j) {
            const Reg& p = r[pri[j]];
            if (p.qe <= qs || p.qs >= qe) continue;
            const int64_t v = (int64_t)(p.qs > qs ? p.qs : qs) << 32 | (uint32_t)(p.qe < qe ? p.qe : qe);
            if (v > last && v < next) next = v;
            same += v == last ? 1 : 0;
        }
        (void)same;   // equal intervals add nothing to the sweep
        if (next == INT64_MAX) break;
        const int32_t s = (int32_t)(next >> 32), e = (int32_t)next;
        if (s > reach) uncovered += s - reach;
        reach = e > reach ? e : reach;
        last = next;
    }
    if (qe > reach) uncovered += qe - reach;
    return uncovered;
}

PMX_HDN void set_parent(Work& W, float mask_level, int mask_len, int n, Reg* r, int sub_diff) {
    PMX_LDS(&W); PMX_LDS(r);
    if (n <= 0) return;
    for (int i = 0; i < n; ++i) r[i].id = i;
    Ptr<int32_t> pri = W.aux32; PMX_LDS(pri);   // indices of the primaries so far
    int n_pri = 1;
    pri[0] = 0;
    r[0].parent = 0;
    for (int i = 1; i < n; ++i) {
        Reg& me = r[i];
        const int32_t qs = me.qs, qe = me.qe, my_len = qe - qs;
        bool touches = false;
        for (int j = 0; j < n_pri && !touches; ++j) touches = !(r[pri[j]].qe <= qs || r[pri[j]].qs >= qe);
        int owner = -1;
        if (touches) {
            const int32_t uncov = uncovered_by_primaries(r, pri, n_pri, qs, qe);
            for (int j = 0; j < n_pri; ++j) {
                Reg& p = r[pri[j]];
                if (p.qe <= qs || p.qs >= qe) continue;
                const int32_t p_len = p.qe - p.qs;
                const int32_t shorter = p_len < my_len ? p_len : my_len, longer = p_len > my_len ? p_len : my_len;
                const int32_t lo = qs > p.qs ? qs : p.qs, hi = qe < p.qe ? qe : p.qe;
                const int32_t ol = hi > lo ? hi - lo : 0;
                if ((float)ol / shorter - (float)uncov / longer > mask_level && uncov <= mask_len) { owner = j; break; }
            }
        }
        if (owner < 0) {
            pri[n_pri++] = i;
            me.parent = i;
            me.n_sub = 0;
            continue;
        }
        // a secondary: it raises its primary's sub-optimal scores and, when it is a real rival, its sub count
        Reg& p = r[pri[owner]];
        const int32_t p_len = p.qe - p.qs;
        const int32_t shorter = p_len < my_len ? p_len : my_len;
        const int32_t lo = qs > p.qs ? qs : p.qs, hi = qe < p.qe ? qe : p.qe;
        const int32_t ol = hi > lo ? hi - lo : 0;
        bool rival = me.cnt >= p.cnt;
        me.parent = p.parent;
        if (p.subsc < me.score) p.subsc = me.score;
        if (p.has_p && me.has_p && (p.rid != me.rid || p.rs != me.rs || p.re != me.re || ol != shorter)) {
            if (p.dp_max2 < me.dp_max) p.dp_max2 = me.dp_max;
            rival = rival || p.dp_max - me.dp_max <= sub_diff;
        }
        if (rival) ++p.n_sub;
    }
}

// mm_set_sam_pri (hit.c:227-237): the first primary of the list is THE primary line of the SAM output
PMX_HD int set_sam_pri(int n, Reg* r) {
    PMX_LDS(r);
    int n_pri = 0;
    for (int i = 0; i < n; ++i) {
        const bool is_pri = r[i].id == r[i].parent;
        n_pri += is_pri ? 1 : 0;
        r[i].sam_pri = (uint8_t)(is_pri && n_pri == 1);
    }
    return n_pri;
}

// mm_sync_regs (hit.c:239-262): after entries were dropped, ids are positions again and parents follow their primaries
PMX_HDN void sync_regs(Work& W, int n_regs, Reg* regs) {
    PMX_LDS(&W); PMX_LDS(regs);
    if (n_regs <= 0) return;
    int id_end = 0;
    for (int i = 0; i < n_regs; ++i) id_end = regs[i].id + 1 > id_end ? regs[i].id + 1 : id_end;
    Ptr<int32_t> now_at = W.aux32; PMX_LDS(now_at);   // old id -> position, -1 = gone
    if (id_end > W.caps.max_reg * 4) { W.status |= PMX_ST_OVERFLOW; return; }
    for (int i = 0; i < id_end; ++i) now_at[i] = -1;
    for (int i = 0; i < n_regs; ++i)
        if (regs[i].id >= 0) now_at[regs[i].id] = i;
    for (int i = 0; i < n_regs; ++i) {
        Reg& g = regs[i];
        const int32_t par = g.parent;
        g.id = i;
        g.parent = par == PMX_PARENT_TMP_PRI ? i : (par >= 0 && par < id_end && now_at[par] >= 0) ? now_at[par] : PMX_PARENT_UNSET;
    }
    set_sam_pri(n_regs, regs);
}

// One in-place thinning pass: entry i survives iff keep(i) -- evaluated on the array AS IT IS when i is reached, i.e. with
// the survivors before it already moved forward (the reference's loops read their parents that way) -- and survivors close
// ranks.  Returns the new length; ids / parents are re-synchronised when something was dropped.
template <class Keep>
PMX_HD int thin_regs(Work& W, int n, Reg* r, Keep keep) {
    int k = 0;
    for (int i = 0; i < n; ++i) {
        if (!keep(i)) continue;
        if (k != i) r[k] = r[i];
        ++k;
    }
    if (k != n) sync_regs(W, k, r);
    return k;
}

// mm_select_sub (hit.c:264-285): which secondaries of a single-segment read are worth keeping
PMX_HDN void select_sub(Work& W, float pri_ratio, int min_diff, int best_n, int check_strand, int min_strand_sc, int* n_, Reg* r) {
    PMX_LDS(&W); PMX_LDS(r);
    if (!(pri_ratio > 0.0f) || *n_ <= 0) return;
    int n_2nd = 0;
    *n_ = thin_regs(W, *n_, r, [&](int i) {
        Reg& me = r[i];
        const int p = me.parent;
        if (p == i || me.inv) return true;
        const Reg& par = r[p];
        if ((me.score >= par.score * pri_ratio || me.score + min_diff >= par.score) && n_2nd < best_n) {
            const bool same_place = me.qs == par.qs && me.qe == par.qe && me.rid == par.rid && me.rs == par.rs && me.re == par.re;
            if (same_place) return false;
            ++n_2nd;
            return true;
        }
        if (check_strand && n_2nd < best_n && me.score > min_strand_sc && me.rev != par.rev) {
            me.strand_retained = 1;
            ++n_2nd;
            return true;
        }
        return false;
    });
}

// mm_select_sub_multi (pe.c:6-43): the same for fragment chains of a read pair; a secondary close to its primary on the
// same strand needs pri1 of its score, one that differs from it in covering both mates or not needs pri2, others pri_ratio
PMX_HDN void select_sub_multi(Work& W, float pri_ratio, float pri1, float pri2, int max_gap_ref, int min_diff, int best_n, int n_segs,
                             const int* qlens, int* n_, Reg* r) {
    PMX_LDS(&W); PMX_LDS(r); PMX_LDS(qlens); PMX_LDS(n_);
    if (!(pri_ratio > 0.0f) || *n_ <= 0) return;
    const int max_dist = n_segs == 2 ? qlens[0] + qlens[1] + max_gap_ref : 0;
    const int32_t mate_border = n_segs == 2 ? qlens[0] : 0;
    int n_2nd = 0;
    *n_ = thin_regs(W, *n_, r, [&](int i) {
        const Reg& q = r[i];
        if (q.parent == i) return true;
        const Reg& p = r[q.parent];
        bool worth;
        if (q.score + min_diff >= p.score) worth = true;
        else if (p.rev == q.rev && p.rid == q.rid && q.re - p.rs < max_dist && p.re - q.rs < max_dist) worth = q.score >= p.score * pri1;
        else {
            const bool p_both = n_segs == 2 && p.qs < mate_border && p.qe > mate_border;
            const bool q_both = n_segs == 2 && q.qs < mate_border && q.qe > mate_border;
            worth = q.score >= p.score * ((q_both || q_both == p_both) ? pri_ratio : pri2);
        }
        if (worth && n_2nd++ >= best_n) worth = false;
        return worth;
    });
}

// mm_filter_strand_retained (hit.c:277-290).  The divergences come from the device's pow() (aln_map.hpp est_err): a
// comparison within a few float steps of equality is flagged, not decided.
PMX_HD int filter_strand_retained(int n_regs, Reg* r, uint32_t* status) {
    PMX_LDS(r);
    int k = 0;
    auto close = [](float x, float y) { const float d = x > y ? x - y : y - x, m = (x < 0 ? -x : x) > (y < 0 ? -y : y) ? (x < 0 ? -x : x) : (y < 0 ? -y : y); return d <= 1e-6f * m; };
    for (int i = 0; i < n_regs; ++i) {
        if (r[i].strand_retained && r[i].div > 0.0f && (close(r[i].div, r[r[i].parent].div * 5.0f) || close(r[i].div, 0.01f))) *status |= PMX_ST_UNSUPPORTED;
        const bool drop = r[i].strand_retained && !(r[i].div < r[r[i].parent].div * 5.0f || r[i].div < 0.01f);
        if (drop) continue;
        if (k != i) r[k] = r[i];
        ++k;
    }
    return k;
}

// mm_filter_regs (hit.c:301-322): regions too thin to report
PMX_HD void filter_regs(const Opt& o, int qlen, int* n_regs, Reg* regs) {
    PMX_LDS(regs);
    const float clip = qlen * o.max_clip_ratio;
    int k = 0;
    for (int i = 0; i < *n_regs; ++i) {
        const Reg& g = regs[i];
        bool drop = !g.inv && !g.seg_split && g.cnt < o.min_cnt;
        if (g.has_p) drop = drop || g.mlen < o.min_chain_score || g.dp_max < o.min_dp_max || (g.qs > clip && qlen - g.qe > clip);
        if (drop) continue;
        if (k != i) regs[k] = regs[i];
        ++k;
    }
    *n_regs = k;
}

// mm_hit_sort (hit.c:193-225): best alignment first -- by DP score once aligned, else chain score -- empty shells dropped
PMX_HDN void hit_sort(Work& W, int* n_regs, Reg* r) {
    PMX_LDS(&W); PMX_LDS(r);
    const int n = *n_regs;
    if (n <= 1) return;
    if (n > 64) { W.status |= PMX_ST_UNSUPPORTED; return; }
    Reg* t = W.reg_tmp; PMX_LDS(t);
    auto listed = [&](int i) { return r[i].inv || r[i].cnt > 0; };
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
    wave_sync();
    for (int i = 0; i < n_out; ++i) r[i] = t[i];
    *n_regs = n_out;
}

// mm_squeeze_a (hit.c:324-343): the anchors of the surviving regions move to the front of a[], in the order of their
// current offsets; each region's `as` follows.  Regions are visited by increasing offset through repeated selection.
PMX_HDN int squeeze_a(Work& W, int n_regs, Reg* regs, Ptr<A128> a) {
    PMX_LDS(&W); PMX_LDS(regs); PMX_LDS(a);
    int write = 0;
    int64_t last = -1;   // (as << 32 | index) of the region placed last
    for (int done = 0; done < n_regs; ++done) {
        int64_t next = INT64_MAX;
        for (int i = 0; i < n_regs; ++i) {
            const int64_t v = (int64_t)(uint32_t)regs[i].as << 32 | (uint32_t)i;
            if (v > last && v < next) next = v;
        }
        Reg& g = regs[(int32_t)next];
        if (g.as != write) {
            for (int j = 0; j < g.cnt; ++j) a[write + j] = a[g.as + j];   // towards lower addresses: no overlap hazard
            g.as = write;
        }
        write += g.cnt;
        last = next;
    }
    return write;
}

// --------------------------------------------------------------------------------------------- per-mate split
// mm_seg_gen (hit.c:345-400) for a read pair: every fragment chain contributes to each mate the anchors that lie on it
// (query positions rebased to the mate), as a chain with the FRAGMENT chain's score; then the per-mate regions are
// generated like the fragment ones and marked as segment splits.
PMX_HDN void seg_gen(Work& W, uint32_t hash, const int* qlens, int n_regs0, const Reg* regs0, Ptr<const A128> a) {
    PMX_LDS(&W); PMX_LDS(qlens); PMX_LDS(regs0); PMX_LDS(a);
    const int n_segs = W.n_segs;
    Ptr<uint64_t> su0 = W.seg_u[0], su1 = W.seg_u[1]; PMX_LDS(su0); PMX_LDS(su1);
    Ptr<A128> sa0 = W.seg_a[0]; PMX_LDS(sa0);
    const int32_t len0 = qlens[0], len1 = n_segs > 1 ? qlens[1] : 0, total = len0 + len1;
    // pass 1: per chain and mate, how many anchors (the chain list of a mate skips chains that do not touch it)
    int n_u0 = 0, n_u1 = 0, n_a0 = 0;
    for (int i = 0; i < n_regs0; ++i) {
        const Reg& f = regs0[i];
        uint32_t on0 = 0, on1 = 0;
        for (int j = 0; j < f.cnt; ++j) {
            const int sg = anc_seg(a[f.as + j]);
            on0 += sg == 0; on1 += sg != 0;
        }
        const uint64_t sc = (uint64_t)(uint32_t)f.score << 32;
        if (on0) su0[n_u0++] = sc + on0;
        if (on1 && n_segs > 1) su1[n_u1++] = sc + on1;
        n_a0 += (int)on0;
    }
    W.seg_n_u[0] = n_u0;
    if (n_segs > 1) W.seg_n_u[1] = n_u1;
    // pass 2: the anchors, mate 0's list first, mate 1's right behind it (both share one max_anchor block)
    Ptr<A128> sa1 = sa0 + n_a0;
    W.seg_a[1] = sa1;
    int w0 = 0, w1 = 0;
    for (int i = 0; i < n_regs0; ++i) {
        const Reg& f = regs0[i];
        for (int j = 0; j < f.cnt; ++j) {
            A128 p = a[f.as + j];
            const bool second = anc_seg(p) != 0;
            const int32_t my_len = second ? len1 : len0, before = second ? len0 : 0;
            p.y -= (uint64_t)(int64_t)((p.x >> 63) ? total - (my_len + before) : before);   // the mate's own coordinates (mirrored on the reverse strand)
            if (second) sa1[w1++] = p;
            else sa0[w0++] = p;
        }
    }
    W.seg_n_a[0] = w0;
    if (n_segs > 1) W.seg_n_a[1] = w1;
    for (int s = 0; s < n_segs; ++s) {
        Reg* mine = W.regs[s]; PMX_LDS(mine);
        W.n_regs[s] = gen_regs(W, hash, qlens[s], W.seg_n_u[s], s ? su1 : su0, s ? sa1 : sa0, mine);
        for (int i = 0; i < W.n_regs[s]; ++i) {
            mine[i].seg_split = 1;
            mine[i].seg_id = (uint8_t)s;
        }
    }
}

// --------------------------------------------------------------------------------------------- mapping quality
// mm_set_mapq (hit.c:421-466) + mm_set_inv_mapq (:395-419).  The float expressions are the reference's, operand for operand
// (they are compiled without contraction); logf comes from tables the host filled with ITS logf (see RefIndex).
PMX_HDN void set_mapq(const RefIndex& ri, int n_regs, Reg* regs, int min_chain_sc, int match_sc, int rep_len, int is_sr, uint32_t* status) {
    PMX_LDS(regs); PMX_LDS(status);
    if (n_regs == 0) return;
    int64_t pri_score_sum = 0;
    for (int i = 0; i < n_regs; ++i) pri_score_sum += regs[i].parent == regs[i].id ? regs[i].score : 0;
    const float uniq_ratio = (float)pri_score_sum / (float)(pri_score_sum + rep_len);
    const float q_coef = 40.0f;
    for (int i = 0; i < n_regs; ++i) {
        Reg& r = regs[i];
        if (r.inv || r.parent != r.id) { r.mapq = 0; continue; }
        const bool tables_ok = !(r.has_p && (r.dp_max < 0 || r.dp_max >= ri.n_logf)) && r.score >= 0 && r.score < ri.n_logf && r.n_sub + 1 < ri.n_logf;
        if (!tables_ok) {   // outside the host logf tables
            *status |= PMX_ST_UNSUPPORTED;
            r.mapq = 0;
            continue;
        }
        const float pen_s1 = (r.score > 100 ? 1.0f : 0.01f * r.score) * uniq_ratio;
        float pen_cm = r.cnt > 10 ? 1.0f : 0.1f * r.cnt;
        pen_cm = pen_s1 < pen_cm ? pen_s1 : pen_cm;
        const int subsc = r.subsc > min_chain_sc ? r.subsc : min_chain_sc;
        int mapq;
        if (r.has_p && r.dp_max2 > 0 && r.dp_max > 0) {   // a competing alignment exists
            const float identity = (float)r.mlen / r.blen;
            const float x = (float)r.dp_max2 * subsc / r.dp_max / r.score0;
            mapq = (int)(identity * pen_cm * q_coef * (1.0f - x * x) * ri.logf_ratio[r.dp_max]);
            if (!is_sr) {
                const int mapq_alt = (int)(6.02f * identity * identity * (r.dp_max - r.dp_max2) / match_sc + .499f);
                mapq = mapq < mapq_alt ? mapq : mapq_alt;
            }
        } else {
            const float x = (float)subsc / r.score0;
            if (r.has_p) {
                const float identity = (float)r.mlen / r.blen;
                mapq = (int)(identity * pen_cm * q_coef * (1.0f - x) * ri.logf_ratio[r.dp_max]);
            } else mapq = (int)(pen_cm * q_coef * (1.0f - x) * ri.logf_int[r.score]);
        }
        mapq -= (int)(4.343f * ri.logf_int[r.n_sub + 1] + .499f);
        mapq = mapq > 0 ? mapq : 0;
        r.mapq = (uint8_t)(mapq < 60 ? mapq : 60);
        if (r.has_p && r.dp_max > r.dp_max2 && r.mapq == 0) r.mapq = 1;
    }
    // an inversion hit takes the lower mapq of its neighbours along the reference (among the primary / parentless hits)
    if (n_regs < 3) return;
    bool any_inv = false;
    for (int i = 0; i < n_regs; ++i) any_inv = any_inv || regs[i].inv;
    if (!any_inv) return;
    for (int i = 0; i < n_regs; ++i) {
        const Reg& g = regs[i];
        if (!g.inv || !(g.parent == i || g.parent < 0)) continue;
        // predecessor / successor of g in the order radix_sort_128x leaves (key rs; equal keys: the order of the reference's
        // sorter is not restated -> flagged)
        int before = -1, after = -1;
        for (int j = 0; j < n_regs; ++j) {
            const Reg& h = regs[j];
            if (!(h.parent == j || h.parent < 0) || j == i) continue;
            if (h.rs == g.rs) { *status |= PMX_ST_UNSUPPORTED; continue; }
            if (h.rs < g.rs) {
                if (before >= 0 && regs[before].rs == h.rs) *status |= PMX_ST_UNSUPPORTED;
                if (before < 0 || regs[before].rs < h.rs) before = j;
            } else {
                if (after >= 0 && regs[after].rs == h.rs) *status |= PMX_ST_UNSUPPORTED;
                if (after < 0 || regs[after].rs > h.rs) after = j;
            }
        }
        if (before < 0 || after < 0) continue;   // first or last in the order: untouched (the loop runs over 1 .. n_aux - 2)
        if (regs[before].inv || regs[after].inv) *status |= PMX_ST_UNSUPPORTED;   // chained inversion hits: update order matters
        regs[i].mapq = regs[before].mapq < regs[after].mapq ? regs[before].mapq : regs[after].mapq;
    }
}

}  // namespace aln
}  // namespace pmx
