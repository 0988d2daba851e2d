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

// mm_pair (pe.c:76-177)
PMX_HDN void pair_hits(Work& W, const RefIndex& ri, int max_gap_ref, int pe_bonus, int sub_diff, int match_sc, const int* qlens, int* n_regs,
                      Reg* const* regs) {
    // pair array (s, rev, key, reg index) kept as parallel arrays in the idle chaining scratch
    PMX_LDS(&W); PMX_LDS(qlens); PMX_LDS(n_regs);
    Reg* r0_ = regs[0]; Reg* r1_ = regs[1];
    PMX_LDS(r0_); PMX_LDS(r1_);
#define PMX_REGS(s) ((s) ? r1_ : r0_)
    const int cap = W.caps.max_reg * 2;
    Ptr<uint64_t> key = W.aux64; PMX_LDS(key);                 // [cap]
    Ptr<int32_t> ps = W.aux32; PMX_LDS(ps);                    // [cap] segment
    Ptr<int32_t> pi = ps + cap;                                // [cap] index in regs[s]
    Ptr<uint64_t> sc = ptr_cast<uint64_t>(W.z); PMX_LDS(sc);   // pair scores
    const int sc_cap = W.caps.max_anchor * 2;      // z holds max_anchor A128
    int n = 0, segs = 0, dp_thres = 0;
    for (int s = 0; s < 2; ++s) {
        int mx = 0;
        for (int i = 0; i < n_regs[s]; ++i) {
            const Reg& r = PMX_REGS(s)[i];
            ps[n] = s;
            pi[n] = i;
            key[n] = (uint64_t)(uint32_t)r.rid << 32 | (uint32_t)(r.rs << 1) | (uint32_t)(s ^ r.rev);
            mx = mx > r.dp_max ? mx : r.dp_max;
            ++n;
            segs |= 1 << s;
        }
        dp_thres += mx;
    }
    if (segs != 3) return;   // only one end is mapped
    dp_thres -= pe_bonus;
    if (dp_thres < 0) dp_thres = 0;
    if (n > 64) { W.status |= PMX_ST_UNSUPPORTED; return; }   // beyond the stable insertion-sort regime of radix_sort_pair
    for (int i = 1; i < n; ++i) {   // rs_insertsort on key
        if (key[i] < key[i - 1]) {
            const uint64_t tk = key[i];
            const int32_t ts = ps[i], ti = pi[i];
            int j;
            for (j = i; j > 0 && tk < key[j - 1]; --j) { key[j] = key[j - 1]; ps[j] = ps[j - 1]; pi[j] = pi[j - 1]; }
            key[j] = tk; ps[j] = ts; pi[j] = ti;
        }
    }
    int64_t mx = -1;
    int max_idx[2] = {-1, -1}, last[2] = {-1, -1};
    int n_sc = 0;
    for (int i = 0; i < n; ++i) {
        const Reg& ri_ = PMX_REGS(ps[i])[pi[i]];
        const int rev_i = ri_.rev;
        if (key[i] & 1) {   // reverse first read or forward second read
            if (last[rev_i] < 0) continue;
            const Reg* q = &PMX_REGS(ps[last[rev_i]])[pi[last[rev_i]]];
            if (ri_.rid != q->rid || ri_.rs - q->re > max_gap_ref) continue;
            for (int j = last[rev_i]; j >= 0; --j) {
                q = &PMX_REGS(ps[j])[pi[j]];
                if (q->rev != rev_i || ps[j] == ps[i]) continue;
                if (ri_.rid != q->rid || ri_.rs - q->re > max_gap_ref) break;
                if (ri_.dp_max + q->dp_max < dp_thres) continue;
                const int64_t score = (int64_t)(ri_.dp_max + q->dp_max) << 32 | (uint32_t)(ri_.hash + q->hash);
                if (score > mx) { mx = score; max_idx[ps[j]] = j; max_idx[ps[i]] = i; }
                if (n_sc < sc_cap) sc[n_sc++] = (uint64_t)score;
                else W.status |= PMX_ST_OVERFLOW;
            }
        } else last[rev_i] = i;
    }
    if (n_sc > 1) radix_sort_64(sc, sc + n_sc, &W.status);
    if (n_sc > 0 && mx > 0) {
        int n_sub = 0, mapq_pe;
        Reg* r[2];
        r[0] = &r0_[pi[max_idx[0]]];
        r[1] = &r1_[pi[max_idx[1]]];
        r[0]->proper_frag = r[1]->proper_frag = 1;
        for (int s = 0; s < 2; ++s) {
            if (r[s]->id != r[s]->parent) {   // lift to primary and update parent
                Reg* p = &PMX_REGS(s)[r[s]->parent];
                const int pid = p->id;
                for (int i = 0; i < n_regs[s]; ++i)
                    if (PMX_REGS(s)[i].parent == pid) PMX_REGS(s)[i].parent = r[s]->id;
                p->mapq = 0;
            }
            if (!r[s]->sam_pri) {
                for (int i = 0; i < n_regs[s]; ++i) PMX_REGS(s)[i].sam_pri = 0;
                r[s]->sam_pri = 1;
            }
        }
        mapq_pe = r[0]->mapq > r[1]->mapq ? r[0]->mapq : r[1]->mapq;
        for (int i = 0; i < n_sc; ++i)
            if ((sc[i] >> 32) + (uint64_t)sub_diff >= (uint64_t)mx >> 32) ++n_sub;
        if (n_sc > 1) {
            if (n_sub >= ri.n_logf) { W.status |= PMX_ST_UNSUPPORTED; n_sub = ri.n_logf - 1; }
            const int mapq_pe_alt = (int)(6.02f * (float)((mx >> 32) - (int64_t)(sc[n_sc - 2] >> 32)) / match_sc - 4.343f * ri.logf_int[n_sub]);
            mapq_pe = mapq_pe < mapq_pe_alt ? mapq_pe : mapq_pe_alt;
        }
        if (r[0]->mapq < mapq_pe) r[0]->mapq = (uint8_t)(int)(.2f * r[0]->mapq + .8f * mapq_pe + .499f);
        if (r[1]->mapq < mapq_pe) r[1]->mapq = (uint8_t)(int)(.2f * r[1]->mapq + .8f * mapq_pe + .499f);
        if (n_sc == 1) {
            if (r[0]->mapq < 2) r[0]->mapq = 2;
            if (r[1]->mapq < 2) r[1]->mapq = 2;
        } else if ((uint64_t)mx >> 32 > sc[n_sc - 2] >> 32) {
            if (r[0]->mapq < 1) r[0]->mapq = 1;
            if (r[1]->mapq < 1) r[1]->mapq = 1;
        }
    }
    set_pe_thru(qlens, n_regs, regs);
#undef PMX_REGS
}

// mm_est_err (esterr.c:30-64): per region, the fraction of the read's minimizers between its first and last anchor that
// are anchors of the region -> a per-base divergence estimate.  mini_pos[] of the reference (query span << 32 | query
// position of every minimizer that was kept as a seed, map.c:157) is read from the seed arrays, which still hold exactly
// those minimizers in query order.  pow(): the device's, not glibc's -- the one reader of the value
// (filter_strand_retained) flags a comparison that a last-bit difference could turn.
PMX_HDN void est_err(Work& W, int l_ref, int qlen, int n_regs, Reg* regs, Ptr<A128> a) {
    PMX_LDS(&W); PMX_LDS(regs);
    const int n = W.n_seeds;
    if (n == 0) return;
    Ptr<SeedA> sa = W.seeds; PMX_LDS(sa);
    Ptr<SeedB> sb = W.seeds_b; PMX_LDS(sb);
    uint64_t sum_k = 0;
    for (int i = 0; i < n; ++i) sum_k += sb[i].q_span & 0xff;
    const float avg_k = (float)sum_k / n;
    auto for_qpos = [&](const A128 p) {   // get_for_qpos
        int32_t x = (int32_t)p.y;
        const int32_t q_span = (int32_t)(p.y >> 32 & 0xff);
        if (p.x >> 63) x = qlen - 1 - (x + 1 - q_span);
        return x;
    };
    for (int i = 0; i < n_regs; ++i) {
        Reg& r = regs[i];
        r.div = -1.0f;
        if (r.cnt == 0) continue;
        int32_t st = -1;
        {   // get_mini_idx: binary search of the first anchor's query position
            const int32_t x = for_qpos(r.rev ? a[r.as + r.cnt - 1] : a[r.as]);
            int32_t L = 0, R = n - 1;
            while (L <= R) {
                const int32_t m = (int32_t)(((uint64_t)L + (uint64_t)R) >> 1);
                const int32_t y = (int32_t)(sa[m].q_pos >> 1);
                if (y < x) L = m + 1;
                else if (y > x) R = m - 1;
                else { st = m; break; }
            }
        }
        if (st < 0) continue;
        int32_t en = st, k = 1, n_match = 1;
        for (int32_t j = st + 1; j < n && k < r.cnt; ++j) {
            const int32_t x = for_qpos(r.rev ? a[r.as + r.cnt - 1 - k] : a[r.as + k]);
            if (x == (int32_t)(sa[j].q_pos >> 1)) { ++k; en = j; ++n_match; }
        }
        int32_t n_tot = en - st + 1;
        if (r.qs > avg_k && r.rs > avg_k) ++n_tot;
        if (qlen - r.qs > avg_k && l_ref - r.re > avg_k) ++n_tot;
        r.div = n_match >= n_tot ? 0.0f : (float)(1.0 - pow((double)n_match / n_tot, 1.0 / avg_k));
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
