// ALIGN stage, part 6: pairing of the two mates' hits (pe.c:45-177) and the per-fragment driver that
// strings the parts together the way mm_map_frag does (map.c:236-390), ending with the record the
// reference's boundary extracts (extract_align_result / align_worker_func, src/mm_align.c:271-354).
#pragma once
#include "aln_align.hpp"
#include "aln_chain.hpp"
#include "aln_hit.hpp"
#include "aln_rmq.hpp"
#include "aln_seed.hpp"
#include "aln_types.hpp"

namespace pmx {
namespace aln {

PMX_HD uint32_t wang_hash(uint32_t key) {   // __ac_Wang_hash (khash.h)
    key += ~(key << 15);
    key ^= (key >> 10);
    key += (key << 3);
    key ^= (key >> 6);
    key += ~(key << 11);
    key ^= (key >> 16);
    return key;
}

// mm_set_pe_thru (pe.c:45-64)
PMX_HD void set_pe_thru(const int* qlens, const int* n_regs, Reg* const* regs) {
    Reg* r0 = regs[0]; Reg* r1 = regs[1];
    PMX_LDS(r0); PMX_LDS(r1); PMX_LDS(qlens); PMX_LDS(n_regs);
    int n_pri[2] = {0, 0}, pri[2] = {-1, -1};
    for (int s = 0; s < 2; ++s)
        for (int i = 0; i < n_regs[s]; ++i)
            if ((s ? r1 : r0)[i].id == (s ? r1 : r0)[i].parent) { ++n_pri[s]; pri[s] = i; }
    if (n_pri[0] == 1 && n_pri[1] == 1) {
        Reg& p = r0[pri[0]];
        Reg& q = r1[pri[1]];
        const int d1 = p.rs - q.rs < 0 ? q.rs - p.rs : p.rs - q.rs, d2 = p.re - q.re < 0 ? q.re - p.re : p.re - q.re;
        if (p.rid == q.rid && p.rev == q.rev && d1 < 3 && d2 < 3 &&
            ((p.qs == 0 && qlens[1] - q.qe == 0) || (q.qs == 0 && qlens[0] - p.qe == 0)))
            p.pe_thru = q.pe_thru = 1;
    }
}

// Pairing of the two mates' regions (what mm_pair decides, pe.c:76-177).
//
// Every region of either mate is an END of a possible fragment: a forward region of mate 1 or a reverse region of mate 2
// OPENS a fragment on its strand, the other two kinds CLOSE one.  The reference sorts the ends by (reference start, kind)
// and sweeps the sorted array; with a handful of regions per mate no array is built here: the ends are visited in that
// order by selection (the order of equal keys is the enumeration order -- mate 1's regions, then mate 2's -- which is what
// the reference's stable insertion sort leaves; more than 64 ends would take its unstable radix sort and are refused).
// For a closing end the candidates are the opening ends of the OTHER mate on the same strand that precede it, nearest
// first, until one lies further back than the maximal fragment length; before that, the nearest opening end of the strand
// from EITHER mate must pass the same distance test.  A pair scores the sum of the two dp_max (the regions' hashes break
// ties); the first pair found with the best score wins.
// The mapq of the winner needs the runner-up's score and the number of pairs within sub_diff of the best: a running
// top two replaces the reference's sort of all scores, the count is taken from the stored upper halves.
PMX_HDN void pair_hits(Work& W, const RefIndex& ri, int max_gap_ref, int pe_bonus, int sub_diff, int match_sc, const int* qlens, int* n_regs,
                      Reg* const* regs) {
    PMX_LDS(&W); PMX_LDS(qlens); PMX_LDS(n_regs);
    Reg* m0 = regs[0]; Reg* m1 = regs[1];
    PMX_LDS(m0); PMX_LDS(m1);
    const int n0 = n_regs[0], n_ends = n0 + n_regs[1];
    if (n0 == 0 || n_regs[1] == 0) return;   // only one end is mapped
    auto mate_of = [&](int e) { return e >= n0 ? 1 : 0; };
    auto reg_of = [&](int e) -> Reg& { return e >= n0 ? m1[e - n0] : m0[e]; };
    auto key_of = [&](int e) {
        const Reg& r = reg_of(e);
        return (uint64_t)(uint32_t)r.rid << 32 | (uint32_t)(r.rs << 1) | (uint32_t)(mate_of(e) ^ r.rev);
    };
    if (n_ends > 64) { W.status |= PMX_ST_UNSUPPORTED; return; }
    // a pair must reach the two best single scores minus the bonus
    int floor_dp = -pe_bonus;
    for (int m = 0; m < 2; ++m) {
        int top = 0;
        for (int i = 0; i < n_regs[m]; ++i) { const int d = (m ? m1 : m0)[i].dp_max; top = top > d ? top : d; }
        floor_dp += top;
    }
    if (floor_dp < 0) floor_dp = 0;

    Ptr<uint64_t> hi_scores = ptr_cast<uint64_t>(W.z); PMX_LDS(hi_scores);   // dp sums of every pair found (for the count)
    const int hi_cap = W.caps.max_anchor * 2;                                 // z holds max_anchor 16-byte cells
    int n_pairs = 0;
    int64_t best = -1, second = -1;       // the two largest pair scores (equal scores count twice)
    int best_end[2] = {-1, -1};

    // the end after (k_prev, e_prev) in the sweep order whose kind bit equals `closing`; -1 when there is none
    auto next_end = [&](bool started, uint64_t k_prev, int e_prev, uint32_t closing) {
        int pick = -1;
        uint64_t k_pick = 0;
        for (int e = 0; e < n_ends; ++e) {
            const uint64_t k = key_of(e);
            if ((uint32_t)(k & 1) != closing) continue;
            if (started && (k < k_prev || (k == k_prev && e <= e_prev))) continue;
            if (pick < 0 || k < k_pick) { pick = e; k_pick = k; }   // (ascending e: the first of equal keys is kept)
        }
        return pick;
    };
    // the opening end of strand `rev` just before (k_lim, e_lim) in the sweep order, from mate `want` (-1: either mate)
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
        const Reg& rc = reg_of(c);
        const int near = opening_before(k_cur, c, rc.rev, -1);
        if (near < 0) continue;
        { const Reg& q = reg_of(near); if (rc.rid != q.rid || rc.rs - q.re > max_gap_ref) continue; }
        uint64_t k_lim = k_cur;
        int e_lim = c;
        for (;;) {
            const int p = opening_before(k_lim, e_lim, rc.rev, 1 - mate_of(c));
            if (p < 0) break;
            k_lim = key_of(p); e_lim = p;
            const Reg& q = reg_of(p);
            if (rc.rid != q.rid || rc.rs - q.re > max_gap_ref) break;
            if (rc.dp_max + q.dp_max < floor_dp) continue;
            const int64_t score = (int64_t)(rc.dp_max + q.dp_max) << 32 | (uint32_t)(rc.hash + q.hash);
            if (score > best) { second = best; best = score; best_end[mate_of(p)] = p; best_end[mate_of(c)] = c; }
            else if (score > second) second = score;
            if (n_pairs < hi_cap) hi_scores[n_pairs] = (uint64_t)(uint32_t)(rc.dp_max + q.dp_max);
            else W.status |= PMX_ST_OVERFLOW;
            ++n_pairs;
        }
    }
    if (n_pairs > hi_cap) n_pairs = hi_cap;

    if (n_pairs > 0 && best > 0) {
        Reg* won[2] = {&reg_of(best_end[0]), &reg_of(best_end[1])};
        won[0]->proper_frag = won[1]->proper_frag = 1;
        for (int m = 0; m < 2; ++m) {
            Reg* all = m ? m1 : m0;
            Reg* w = won[m];
            if (w->id != w->parent) {   // a secondary won: it takes its primary's place, the former primary keeps no mapq
                Reg& old_pri = all[w->parent];
                const int old_id = old_pri.id;
                for (int i = 0; i < n_regs[m]; ++i)
                    if (all[i].parent == old_id) all[i].parent = w->id;
                old_pri.mapq = 0;
            }
            if (!w->sam_pri) {
                for (int i = 0; i < n_regs[m]; ++i) all[i].sam_pri = 0;
                w->sam_pri = 1;
            }
        }
        const uint32_t best_hi = (uint32_t)((uint64_t)best >> 32);
        int pe_q = won[0]->mapq > won[1]->mapq ? won[0]->mapq : won[1]->mapq;
        if (n_pairs > 1) {
            int close = 0;                // pairs within sub_diff of the best (the best included)
            for (int i = 0; i < n_pairs; ++i) close += hi_scores[i] + (uint64_t)sub_diff >= (uint64_t)best_hi;
            if (close >= ri.n_logf) { W.status |= PMX_ST_UNSUPPORTED; close = ri.n_logf - 1; }
            const int by_margin = (int)(6.02f * (float)((best >> 32) - (second >> 32)) / match_sc - 4.343f * ri.logf_int[close]);
            pe_q = pe_q < by_margin ? pe_q : by_margin;
        }
        const int at_least = n_pairs == 1 ? 2 : (best_hi > (uint32_t)((uint64_t)second >> 32) ? 1 : 0);
        for (int m = 0; m < 2; ++m) {
            Reg* w = won[m];
            if (w->mapq < pe_q) w->mapq = (uint8_t)(int)(.2f * w->mapq + .8f * pe_q + .499f);
            if (w->mapq < at_least) w->mapq = (uint8_t)at_least;
        }
    }
    set_pe_thru(qlens, n_regs, regs);
}

// mm_est_err (esterr.c:30-64): per region, the fraction of the read's minimizers between its first and last anchor that
// are anchors of the region -> a per-base divergence estimate.  mini_pos[] of the reference (query span << 32 | query
// position of every minimizer that was kept as a seed, map.c:157) is read from the seed arrays, which still hold exactly
// those minimizers in query order.  pow(): the device's, not glibc's -- the one reader of the value
// (filter_strand_retained) flags a comparison that a last-bit difference could turn.
PMX_HDN void est_err(Work& W, int l_ref, int qlen, int n_regs, Reg* regs, Ptr<A128> a) {
    PMX_LDS(&W); PMX_LDS(regs);
    const int n_mini = W.n_seeds;
    if (n_mini == 0) return;
    Ptr<SeedA> mini = W.seeds; PMX_LDS(mini);
    Ptr<SeedB> mini_b = W.seeds_b; PMX_LDS(mini_b);
    float mean_span;
    {
        uint64_t spans = 0;
        for (int i = 0; i < n_mini; ++i) spans += mini_b[i].q_span & 0xff;
        mean_span = (float)spans / n_mini;
    }
    auto mini_at = [&](int m) { return (int32_t)(mini[m].q_pos >> 1); };
    for (int i = 0; i < n_regs; ++i) {
        Reg& r = regs[i];
        r.div = -1.0f;
        if (r.cnt == 0) continue;
        // the region's anchors in the order of the read as sequenced (a reverse-strand chain runs backwards through it), each
        // as the position of its minimizer's last base on that strand
        auto anchor_at = [&](int k) {
            const A128 p = r.rev ? a[r.as + r.cnt - 1 - k] : a[r.as + k];
            const int32_t last = (int32_t)p.y, span = (int32_t)(p.y >> 32 & 0xff);
            return (p.x >> 63) ? qlen - 1 - (last + 1 - span) : last;
        };
        // the first anchor among the minimizers (they are in read order): bisection; an anchor that is no minimizer of the
        // list (cannot happen for a chain of this read) leaves the estimate unset
        int first = -1;
        {
            const int32_t want = anchor_at(0);
            int32_t lo = 0, hi = n_mini - 1;
            while (lo <= hi && first < 0) {
                const int32_t mid = (int32_t)(((uint64_t)lo + (uint64_t)hi) >> 1), at = mini_at(mid);
                if (at == want) first = mid;
                else if (at < want) lo = mid + 1;
                else hi = mid - 1;
            }
        }
        if (first < 0) continue;
        // walk the minimizers from there and tick off the anchors in turn: `hits` of the minimizers up to the last one
        // ticked off are anchors of the chain
        int32_t last_hit = first, hits = 1;
        for (int32_t m = first + 1; m < n_mini && hits < r.cnt; ++m)
            if (mini_at(m) == anchor_at(hits)) { last_hit = m; ++hits; }
        int32_t trials = last_hit - first + 1;
        // a chain that stops short of the read's (reference's) ends missed one more minimizer on that side
        trials += (r.qs > mean_span && r.rs > mean_span) ? 1 : 0;
        trials += (qlen - r.qs > mean_span && l_ref - r.re > mean_span) ? 1 : 0;
        r.div = hits >= trials ? 0.0f : (float)(1.0 - pow((double)hits / trials, 1.0 / mean_span));
    }
}

// mm_map_frag (map.c:236-390) for n_segs in {1,2}.  Regions end up in W.regs[s] / W.n_regs[s].
PMX_HDN void map_frag(Work& W, const Opt& o, const RefIndex& ri) {
    PMX_LDS(&W);
    Ptr<A128> a_ = W.a; PMX_LDS(a_);
    Ptr<uint64_t> u_ = W.u; PMX_LDS(u_);
    Reg* regs0_ = W.regs0; PMX_LDS(regs0_);
    const int n_segs = W.n_segs;
    int qlen_sum = 0;
    for (int i = 0; i < n_segs; ++i) { qlen_sum += W.qlen[i]; W.n_regs[i] = 0; }
    W.n_regs0 = 0;
    W.cig_next = 0;
    if (qlen_sum == 0) return;
    // qname is NULL at the boundary (src/mm_align.c:316): hash depends on lengths and the seed only
    uint32_t hash = 0;
    hash ^= wang_hash((uint32_t)qlen_sum) + wang_hash((uint32_t)o.seed);
    hash = wang_hash(hash);

    if (!W.mv_ready) collect_minimizers(W, o);
    PMX_STAMP(W, 1);
    if (o.q_occ_frac > 0.0f) seed_mz_flt(W, o.mid_occ, o.q_occ_frac);   // map.c:251
    // MM_F_HEAP_SORT is set for the short-read branch only (src/mm_align.c:140-166)
    if (o.is_sr_like) collect_seed_hits_heap(W, o, ri, qlen_sum, o.mid_occ);
    else collect_seed_hits_sorted(W, o, ri, qlen_sum, o.mid_occ);
    PMX_STAMP(W, 2);

    const int max_chain_gap_qry = o.max_gap;   // not MM_F_SR
    int max_chain_gap_ref;
    if (o.max_gap_ref > 0) max_chain_gap_ref = o.max_gap_ref;
    else if (o.max_frag_len > 0) {
        max_chain_gap_ref = o.max_frag_len - qlen_sum;
        if (max_chain_gap_ref < o.max_gap) max_chain_gap_ref = o.max_gap;
    } else max_chain_gap_ref = o.max_gap;

    chain_dp(W, o, max_chain_gap_ref, max_chain_gap_qry, n_segs);
    PMX_STAMP(W, 3);

    if (o.bw_long > o.bw && n_segs == 1 && W.n_u > 1) {   // long-join re-chaining of long reads (map.c:296-305)
        const int32_t st = (int32_t)a_[0].y, en = (int32_t)a_[(int32_t)u_[0] - 1].y;
        if (qlen_sum - (en - st) > o.rmq_rescue_size || en - st > qlen_sum * o.rmq_rescue_ratio) {
#if defined(PMX_ALL_LDS)
            W.status |= PMX_ST_OVERFLOW;   // no room for the trees in this layout (and bw_long == bw for its reads): general tier
#else
            radix_sort_128x(a_, a_ + W.n_a, &W.status);   // W.n_a: the anchors that are in chains
            chain_rmq(W, o.max_gap, o.rmq_inner_dist, o.bw_long, o.max_chain_skip, o.rmq_size_cap, o.min_cnt, o.min_chain_score, o.chn_pen_gap,
                      o.chn_pen_skip);
#endif
        }
    } else if (o.max_occ > o.mid_occ && W.rep_len > 0) {   // re-chain with a higher occurrence cap (map.c:306-330)
        int rechain = 0;
        if (W.n_u > 0) {
            int n_chained_segs = 1, mx = 0, max_i = -1, max_off = -1, off = 0;
            for (int i = 0; i < W.n_u; ++i) {
                if (mx < (int)(u_[i] >> 32)) { mx = (int)(u_[i] >> 32); max_i = i; max_off = off; }
                off += (int32_t)u_[i];
            }
            for (int i = 1; i < (int32_t)u_[max_i]; ++i)
                if ((a_[max_off + i].y & PMX_SEED_SEG_MASK) != (a_[max_off + i - 1].y & PMX_SEED_SEG_MASK)) ++n_chained_segs;
            if (n_chained_segs < n_segs) rechain = 1;
        } else rechain = 1;
        if (rechain) {
            if (o.is_sr_like) collect_seed_hits_heap(W, o, ri, qlen_sum, o.max_occ);
            else collect_seed_hits_sorted(W, o, ri, qlen_sum, o.max_occ);
            chain_dp(W, o, max_chain_gap_ref, max_chain_gap_qry, n_segs);
        }
    }
    W.frag_gap = max_chain_gap_ref;

    W.n_regs0 = gen_regs(W, hash, qlen_sum, W.n_u, u_, a_, regs0_);

    // chain_post (map.c:206-213)
    set_parent(W, o.mask_level, o.mask_len, W.n_regs0, regs0_, o.a * 2 + o.b);
    if (n_segs <= 1) select_sub(W, o.pri_ratio, o.k * 2, o.best_n, 1, (int)(o.max_gap * 0.8), &W.n_regs0, regs0_);
    else select_sub_multi(W, o.pri_ratio, 0.2f, 0.7f, max_chain_gap_ref, o.k * 2, o.best_n, n_segs, W.qlen, &W.n_regs0, regs0_);
    // mm_est_err + mm_filter_strand_retained (map.c:334-335; single-segment mode only).  The divergence estimate only
    // decides which strand_retained hits stay, so it is evaluated when there is one
    if (n_segs == 1) {
        bool any = false;
        for (int i = 0; i < W.n_regs0; ++i) any = any || regs0_[i].strand_retained;
        if (any) {
#if defined(PMX_ALL_LDS)
            W.status |= PMX_ST_OVERFLOW;   // this layout's seed arrays are overlaid by now: the general tier takes the read
#else
            est_err(W, ri.len, qlen_sum, W.n_regs0, regs0_, a_);
            W.n_regs0 = filter_strand_retained(W.n_regs0, regs0_, &W.status);
#endif
        }
    }

    if (n_segs == 1) {
        Reg* rs0 = W.regs[0]; PMX_LDS(rs0);
        for (int i = 0; i < W.n_regs0; ++i) rs0[i] = regs0_[i];
        W.n_regs[0] = W.n_regs0;
        align_regs(W, o, ri, 0, &W.n_regs[0], rs0, a_);
        if (W.status & PMX_ST_ABORT) return;
        set_mapq(ri, W.n_regs[0], rs0, o.min_chain_score, o.a, W.rep_len, 0, &W.status);
    } else {
        PMX_STAMP(W, 4);
        seg_gen(W, hash, W.qlen, W.n_regs0, regs0_, a_);
        PMX_STAMP(W, 5);
        for (int s = 0; s < n_segs; ++s) {
            set_parent(W, o.mask_level, o.mask_len, W.n_regs[s], W.regs[s], o.a * 2 + o.b);
            align_regs(W, o, ri, s, &W.n_regs[s], W.regs[s], W.seg_a[s]);
            if (W.status & PMX_ST_NEED_WAVE) return;
            if (W.status & PMX_ST_NEED_DP) continue;   // thread-per-pair kernel: the other mate posts its DP requests in the same pass
            set_mapq(ri, W.n_regs[s], W.regs[s], o.min_chain_score, o.a, W.rep_len, 0, &W.status);
            PMX_STAMP(W, 9);
        }
        if (W.status & PMX_ST_ABORT) return;
        if (n_segs == 2 && o.pe_ori >= 0) {
            Reg* rr[2] = {W.regs[0], W.regs[1]};
            pair_hits(W, ri, max_chain_gap_ref, o.pe_bonus, o.a * 2 + o.b, o.a, W.qlen, W.n_regs, rr);
            PMX_STAMP(W, 10);
        }
    }
}

}  // namespace aln
}  // namespace pmx
