// ALIGN stage, part 3: region ("hit") bookkeeping -- chains -> regions, primary/secondary relations,
// per-segment split for read pairs, sorting, filtering, mapping quality.
// Reference behaviour: hit.c:8-466, pe.c:6-43 (mm_select_sub_multi), esterr.c:30-64.
#pragma once
#include "aln_sort.hpp"
#include "aln_types.hpp"

namespace pmx {
namespace aln {

// mm_cal_fuzzy_len + mm_reg_set_coor (hit.c:8-40); is_qstrand == 0
PMX_HD void reg_set_coor(Reg& r, int32_t qlen, Ptr<const A128> a) {
    PMX_LDS(a);
    const int32_t k = r.as, q_span = (int32_t)(a[k].y >> 32 & 0xff);
    r.rev = (uint8_t)(a[k].x >> 63);
    r.rid = (int32_t)(a[k].x << 1 >> 33);
    r.rs = (int32_t)a[k].x + 1 > q_span ? (int32_t)a[k].x + 1 - q_span : 0;
    r.re = (int32_t)a[k + r.cnt - 1].x + 1;
    if (!r.rev) {
        r.qs = (int32_t)a[k].y + 1 - q_span;
        r.qe = (int32_t)a[k + r.cnt - 1].y + 1;
    } else {
        r.qs = qlen - ((int32_t)a[k + r.cnt - 1].y + 1);
        r.qe = qlen - ((int32_t)a[k].y + 1 - q_span);
    }
    r.mlen = r.blen = 0;
    if (r.cnt <= 0) return;
    r.mlen = r.blen = q_span;
    for (int i = r.as + 1; i < r.as + r.cnt; ++i) {
        const int span = (int)(a[i].y >> 32 & 0xff);
        const int tl = (int32_t)a[i].x - (int32_t)a[i - 1].x;
        const int ql = (int32_t)a[i].y - (int32_t)a[i - 1].y;
        r.blen += tl > ql ? tl : ql;
        r.mlen += tl > span && ql > span ? span : tl < ql ? tl : ql;
    }
}

PMX_HD uint64_t hit_hash64(uint64_t key) {   // hit.c:42-52 (unmasked variant)
    key = (~key + (key << 21));
    key = key ^ key >> 24;
    key = ((key + (key << 3)) + (key << 8));
    key = key ^ key >> 14;
    key = ((key + (key << 2)) + (key << 4));
    key = key ^ key >> 28;
    key = (key + (key << 31));
    return key;
}

PMX_HD void reg_clear(Reg& r) {
#define PMX_X(f) r.f = 0;
    PMX_REG_FIELDS(PMX_X)
#undef PMX_X
    r.pad_b[0] = r.pad_b[1] = r.pad_b[2] = 0;
}

// mm_gen_regs (hit.c:54-94): chains sorted by (score, hash) descending
PMX_HDN int gen_regs(Work& W, uint32_t hash, int qlen, int n_u, Ptr<const uint64_t> u, Ptr<const A128> a, Reg* r) {
    PMX_LDS(&W); PMX_LDS(u); PMX_LDS(a); PMX_LDS(r);
    if (n_u == 0) return 0;
    if (n_u > W.caps.max_reg) { W.status |= PMX_ST_OVERFLOW; n_u = W.caps.max_reg; }
    Ptr<A128> z = W.aux128; PMX_LDS(z);
    int k = 0;
    for (int i = 0; i < n_u; ++i) {
        const uint32_t h = (uint32_t)hit_hash64((hit_hash64(a[k].x) + hit_hash64(a[k].y)) ^ hash);
        z[i].x = u[i] ^ h;
        z[i].y = (uint64_t)k << 32 | (uint32_t)(int32_t)u[i];
        k += (int32_t)u[i];
    }
    radix_sort_128x(z, z + n_u, &W.status);
    for (int i = 0; i < n_u >> 1; ++i) { const A128 tmp = z[i]; z[i] = z[n_u - 1 - i]; z[n_u - 1 - i] = tmp; }
    for (int i = 0; i < n_u; ++i) {
        Reg& ri = r[i];
        reg_clear(ri);
        ri.id = i;
        ri.parent = PMX_PARENT_UNSET;
        ri.score = ri.score0 = (int32_t)(z[i].x >> 32);
        ri.hash = (uint32_t)z[i].x;
        ri.cnt = (int32_t)z[i].y;
        ri.as = (int32_t)(z[i].y >> 32);
        ri.div = -1.0f;
        reg_set_coor(ri, qlen, a);
    }
    return n_u;
}

// mm_split_reg (hit.c:112-130)
PMX_HD void split_reg(Reg& r, Reg& r2, int n, int qlen, Ptr<const A128> a) {
    PMX_LDS(&r); PMX_LDS(a);   // r2 is a caller-local (private) object
    if (n <= 0 || n >= r.cnt) return;
    r2 = r;
    r2.id = -1;
    r2.sam_pri = 0;
    r2.has_p = 0;
    r2.n_cigar = 0; r2.dp_score = r2.dp_max = r2.dp_max2 = 0; r2.n_ambi = 0;
    r2.split_inv = 0;
    r2.cnt = r.cnt - n;
    r2.score = (int32_t)(r.score * ((float)r2.cnt / r.cnt) + .499);
    r2.as = r.as + n;
    if (r.parent == r.id) r2.parent = PMX_PARENT_TMP_PRI;
    reg_set_coor(r2, qlen, a);
    r.cnt -= r2.cnt;
    r.score -= r2.score;
    reg_set_coor(r, qlen, a);
    r.split |= 1;
    r2.split |= 2;
}

// mm_set_parent (hit.c:132-191); hard_mask_level == 0, no ALT contigs
PMX_HDN void set_parent(Work& W, float mask_level, int mask_len, int n, Reg* r, int sub_diff) {
    PMX_LDS(&W); PMX_LDS(r);
    if (n <= 0) return;
    for (int i = 0; i < n; ++i) r[i].id = i;
    Ptr<uint64_t> cov = W.aux64; PMX_LDS(cov);
    Ptr<int32_t> w = W.aux32; PMX_LDS(w);
    w[0] = 0;
    r[0].parent = 0;
    int k = 1;
    for (int i = 1; i < n; ++i) {
        Reg& ri = r[i];
        const int si = ri.qs, ei = ri.qe;
        int n_cov = 0, uncov_len = 0, j;
        for (j = 0; j < k; ++j) {
            const Reg& rp = r[w[j]];
            int sj = rp.qs, ej = rp.qe;
            if (ej <= si || sj >= ei) continue;
            if (sj < si) sj = si;
            if (ej > ei) ej = ei;
            cov[n_cov++] = (uint64_t)(uint32_t)sj << 32 | (uint32_t)ej;
        }
        bool is_new_primary = n_cov == 0;
        if (!is_new_primary) {
            int x = si;
            radix_sort_64(cov, cov + n_cov, &W.status);
            for (int q = 0; q < n_cov; ++q) {
                if ((int)(cov[q] >> 32) > x) uncov_len += (int)(cov[q] >> 32) - x;
                x = (int32_t)cov[q] > x ? (int32_t)cov[q] : x;
            }
            if (ei > x) uncov_len += ei - x;
            for (j = 0; j < k; ++j) {
                Reg& rp = r[w[j]];
                const int sj = rp.qs, ej = rp.qe;
                if (ej <= si || sj >= ei) continue;
                const int mn = ej - sj < ei - si ? ej - sj : ei - si;
                const int mx = ej - sj > ei - si ? ej - sj : ei - si;
                const int ol = si < sj ? (ei < sj ? 0 : ei < ej ? ei - sj : ej - sj) : (ej < si ? 0 : ej < ei ? ej - si : ei - si);
                if ((float)ol / mn - (float)uncov_len / mx > mask_level && uncov_len <= mask_len) {
                    int cnt_sub = 0, sci = ri.score;
                    ri.parent = rp.parent;
                    rp.subsc = rp.subsc > sci ? rp.subsc : sci;
                    if (ri.cnt >= rp.cnt) cnt_sub = 1;
                    if (rp.has_p && ri.has_p && (rp.rid != ri.rid || rp.rs != ri.rs || rp.re != ri.re || ol != mn)) {
                        sci = ri.dp_max;
                        rp.dp_max2 = rp.dp_max2 > sci ? rp.dp_max2 : sci;
                        if (rp.dp_max - ri.dp_max <= sub_diff) cnt_sub = 1;
                    }
                    if (cnt_sub) ++rp.n_sub;
                    break;
                }
            }
            is_new_primary = j == k;
        }
        if (is_new_primary) { w[k++] = i; ri.parent = i; ri.n_sub = 0; }
    }
}

// mm_set_sam_pri (hit.c:227-237)
PMX_HD int set_sam_pri(int n, Reg* r) {
    PMX_LDS(r);
    int n_pri = 0;
    for (int i = 0; i < n; ++i) {
        if (r[i].id == r[i].parent) { ++n_pri; r[i].sam_pri = (n_pri == 1); }
        else r[i].sam_pri = 0;
    }
    return n_pri;
}

// mm_sync_regs (hit.c:239-262)
PMX_HDN void sync_regs(Work& W, int n_regs, Reg* regs) {
    PMX_LDS(&W); PMX_LDS(regs);
    if (n_regs <= 0) return;
    int max_id = -1;
    for (int i = 0; i < n_regs; ++i) max_id = max_id > regs[i].id ? max_id : regs[i].id;
    const int n_tmp = max_id + 1;
    Ptr<int32_t> tmp = W.aux32; PMX_LDS(tmp);
    if (n_tmp > W.caps.max_reg * 4) { W.status |= PMX_ST_OVERFLOW; return; }
    for (int i = 0; i < n_tmp; ++i) tmp[i] = -1;
    for (int i = 0; i < n_regs; ++i)
        if (regs[i].id >= 0) tmp[regs[i].id] = i;
    for (int i = 0; i < n_regs; ++i) {
        Reg& r = regs[i];
        r.id = i;
        if (r.parent == PMX_PARENT_TMP_PRI) r.parent = i;
        else if (r.parent >= 0 && tmp[r.parent] >= 0) r.parent = tmp[r.parent];
        else r.parent = PMX_PARENT_UNSET;
    }
    set_sam_pri(n_regs, regs);
}

// mm_select_sub (hit.c:264-285)
PMX_HDN void select_sub(Work& W, float pri_ratio, int min_diff, int best_n, int check_strand, int min_strand_sc, int* n_, Reg* r) {
    PMX_LDS(&W); PMX_LDS(r);
    if (pri_ratio > 0.0f && *n_ > 0) {
        const int n = *n_;
        int k = 0, n_2nd = 0;
        for (int i = 0; i < n; ++i) {
            const int p = r[i].parent;
            if (p == i || r[i].inv) {
                r[k++] = r[i];
            } else if ((r[i].score >= r[p].score * pri_ratio || r[i].score + min_diff >= r[p].score) && n_2nd < best_n) {
                if (!(r[i].qs == r[p].qs && r[i].qe == r[p].qe && r[i].rid == r[p].rid && r[i].rs == r[p].rs && r[i].re == r[p].re)) {
                    r[k++] = r[i];
                    ++n_2nd;
                }
            } else if (check_strand && n_2nd < best_n && r[i].score > min_strand_sc && r[i].rev != r[p].rev) {
                r[i].strand_retained = 1;
                r[k++] = r[i];
                ++n_2nd;
            }
        }
        if (k != n) sync_regs(W, k, r);
        *n_ = k;
    }
}

// mm_select_sub_multi (pe.c:6-43)
PMX_HDN void select_sub_multi(Work& W, float pri_ratio, float pri1, float pri2, int max_gap_ref, int min_diff, int best_n, int n_segs,
                             const int* qlens, int* n_, Reg* r) {
    PMX_LDS(&W); PMX_LDS(r); PMX_LDS(qlens); PMX_LDS(n_);
    if (pri_ratio > 0.0f && *n_ > 0) {
        const int n = *n_;
        int k = 0, n_2nd = 0;
        const int max_dist = n_segs == 2 ? qlens[0] + qlens[1] + max_gap_ref : 0;
        for (int i = 0; i < n; ++i) {
            int to_keep = 0;
            if (r[i].parent == i) to_keep = 1;
            else if (r[i].score + min_diff >= r[r[i].parent].score) to_keep = 1;
            else {
                const Reg &p = r[r[i].parent], &q = r[i];
                if (p.rev == q.rev && p.rid == q.rid && q.re - p.rs < max_dist && p.re - q.rs < max_dist) {
                    if (q.score >= p.score * pri1) to_keep = 1;
                } else {
                    const int is_par_both = (n_segs == 2 && p.qs < qlens[0] && p.qe > qlens[0]);
                    const int is_chi_both = (n_segs == 2 && q.qs < qlens[0] && q.qe > qlens[0]);
                    if (is_chi_both || is_chi_both == is_par_both) {
                        if (q.score >= p.score * pri_ratio) to_keep = 1;
                    } else {
                        if (q.score >= p.score * pri2) to_keep = 1;
                    }
                }
            }
            if (to_keep && r[i].parent != i) {
                if (n_2nd++ >= best_n) to_keep = 0;
            }
            if (to_keep) r[k++] = r[i];
        }
        if (k != n) sync_regs(W, k, r);
        *n_ = k;
    }
}

// mm_filter_strand_retained (hit.c:287-299)
PMX_HD int filter_strand_retained(int n_regs, Reg* r) {
    PMX_LDS(r);
    int k = 0;
    for (int i = 0; i < n_regs; ++i) {
        const int p = r[i].parent;
        if (!r[i].strand_retained || r[i].div < r[p].div * 5.0f || r[i].div < 0.01f) {
            if (k < i) r[k++] = r[i];
            else ++k;
        }
    }
    return k;
}

// mm_filter_regs (hit.c:301-322)
PMX_HD void filter_regs(const Opt& o, int qlen, int* n_regs, Reg* regs) {
    PMX_LDS(regs);
    int k = 0;
    for (int i = 0; i < *n_regs; ++i) {
        Reg& r = regs[i];
        int flt = 0;
        if (!r.inv && !r.seg_split && r.cnt < o.min_cnt) flt = 1;
        if (r.has_p) {
            if (r.mlen < o.min_chain_score) flt = 1;
            else if (r.dp_max < o.min_dp_max) flt = 1;
            else if (r.qs > qlen * o.max_clip_ratio && qlen - r.qe > qlen * o.max_clip_ratio) flt = 1;
        }
        if (!flt) {
            if (k < i) regs[k++] = regs[i];
            else ++k;
        }
    }
    *n_regs = k;
}

// mm_hit_sort (hit.c:193-225): by (dp_max or score, hash) descending; cnt==0 regions squeezed out
PMX_HDN void hit_sort(Work& W, int* n_regs, Reg* r) {
    PMX_LDS(&W); PMX_LDS(r);
    const int n = *n_regs;
    if (n <= 1) return;
    Ptr<A128> aux = W.aux128; PMX_LDS(aux);
    Reg* t = W.reg_tmp; PMX_LDS(t);
    int n_aux = 0;
    for (int i = 0; i < n; ++i) {
        if (r[i].inv || r[i].cnt > 0) {
            const int score = r[i].has_p ? r[i].dp_max : r[i].score;
            aux[n_aux].x = (uint64_t)(int64_t)score << 32 | r[i].hash;
            aux[n_aux++].y = (uint64_t)i;
        }
    }
    radix_sort_128x(aux, aux + n_aux, &W.status);
    for (int i = n_aux - 1; i >= 0; --i) t[n_aux - 1 - i] = r[aux[i].y];
    for (int i = 0; i < n_aux; ++i) r[i] = t[i];
    *n_regs = n_aux;
}

// mm_squeeze_a (hit.c:324-343)
PMX_HDN int squeeze_a(Work& W, int n_regs, Reg* regs, Ptr<A128> a) {
    PMX_LDS(&W); PMX_LDS(regs); PMX_LDS(a);
    Ptr<uint64_t> aux = W.aux64; PMX_LDS(aux);
    int as = 0;
    for (int i = 0; i < n_regs; ++i) aux[i] = (uint64_t)(uint32_t)regs[i].as << 32 | (uint32_t)i;
    radix_sort_64(aux, aux + n_regs, &W.status);
    for (int i = 0; i < n_regs; ++i) {
        Reg& r = regs[(int32_t)aux[i]];
        if (r.as != as) {
            for (int j = 0; j < r.cnt; ++j) a[as + j] = a[r.as + j];   // memmove to a lower address
            r.as = as;
        }
        as += r.cnt;
    }
    return as;
}

// mm_seg_gen (hit.c:345-400) for n_segs == 2: split fragment chains into per-mate chains
PMX_HDN void seg_gen(Work& W, uint32_t hash, const int* qlens, int n_regs0, const Reg* regs0, Ptr<const A128> a) {
    PMX_LDS(&W); PMX_LDS(qlens); PMX_LDS(regs0); PMX_LDS(a);
    Ptr<uint64_t> su[2] = {W.seg_u[0], W.seg_u[1]};
    PMX_LDS(su[0]); PMX_LDS(su[1]);
    Ptr<A128> sa0 = W.seg_a[0]; PMX_LDS(sa0);
    const int n_segs = W.n_segs;
    int acc_qlen[3];
    acc_qlen[0] = 0;
    for (int s = 1; s < n_segs; ++s) acc_qlen[s] = acc_qlen[s - 1] + qlens[s - 1];
    const int qlen_sum = acc_qlen[n_segs - 1] + qlens[n_segs - 1];
    // u[s][i] = score << 32 | anchors of segment s in chain i (counted in registers, one store per entry)
    int n_seg_anchors[2] = {0, 0};
    for (int i = 0; i < n_regs0; ++i) {
        const Reg& r = regs0[i];
        const int r_as = r.as, r_cnt = r.cnt;
        const uint64_t sc_hi = (uint64_t)(uint32_t)r.score << 32;
        uint32_t c0 = 0, c1 = 0;
        for (int j = 0; j < r_cnt; ++j) {
            const int sid = (int)((a[r_as + j].y & PMX_SEED_SEG_MASK) >> PMX_SEED_SEG_SHIFT);
            c0 += sid ? 0u : 1u;
            c1 += sid ? 1u : 0u;
        }
        su[0][i] = sc_hi + c0;
        if (n_segs > 1) su[1][i] = sc_hi + c1;
        n_seg_anchors[0] += (int)c0;
        n_seg_anchors[1] += (int)c1;
    }
    W.seg_a[1] = sa0 + n_seg_anchors[0];   // both mates' anchor lists share one max_anchor block
    Ptr<A128> sa1 = sa0 + n_seg_anchors[0];
    for (int s = 0; s < n_segs; ++s) {
        int n_u = 0;
        for (int i = 0; i < n_regs0; ++i)
            if ((int32_t)(s ? su[1] : su[0])[i] != 0) { (s ? su[1] : su[0])[n_u] = (s ? su[1] : su[0])[i]; ++n_u; }
        W.seg_n_u[s] = n_u;
    }
    {
        int na0 = 0, na1 = 0;   // output cursors of the two mates
        const int ql0 = qlens[0], ql1 = n_segs > 1 ? qlens[1] : 0;
        for (int i = 0; i < n_regs0; ++i) {
            const Reg& r = regs0[i];
            const int r_as = r.as, r_cnt = r.cnt;
            for (int j = 0; j < r_cnt; ++j) {
                A128 a1 = a[r_as + j];
                const int sid = (int)((a1.y & PMX_SEED_SEG_MASK) >> PMX_SEED_SEG_SHIFT);
                const int ql = sid ? ql1 : ql0, acc = sid ? acc_qlen[1] : acc_qlen[0];
                a1.y -= (uint64_t)(int64_t)(a1.x >> 63 ? qlen_sum - (ql + acc) : acc);
                if (sid) sa1[na1++] = a1;
                else sa0[na0++] = a1;
            }
        }
        W.seg_n_a[0] = na0;
        if (n_segs > 1) W.seg_n_a[1] = na1;
    }
    for (int s = 0; s < n_segs; ++s) {
        Reg* rs_ = W.regs[s]; PMX_LDS(rs_);
        W.n_regs[s] = gen_regs(W, hash, qlens[s], W.seg_n_u[s], s ? su[1] : su[0], s ? sa1 : sa0, rs_);
        for (int i = 0; i < W.n_regs[s]; ++i) {
            rs_[i].seg_split = 1;
            rs_[i].seg_id = (uint8_t)s;
        }
    }
}

// mm_set_mapq (hit.c:421-466) without inversion hits; logf values come from host-computed tables
PMX_HDN void set_mapq(const RefIndex& ri, int n_regs, Reg* regs, int min_chain_sc, int match_sc, int rep_len, int is_sr, uint32_t* status) {
    PMX_LDS(regs); PMX_LDS(status);
    const float q_coef = 40.0f;
    int64_t sum_sc = 0;
    if (n_regs == 0) return;
    for (int i = 0; i < n_regs; ++i)
        if (regs[i].parent == regs[i].id) sum_sc += regs[i].score;
    const float uniq_ratio = (float)sum_sc / (float)(sum_sc + rep_len);
    for (int i = 0; i < n_regs; ++i) {
        Reg& r = regs[i];
        if (r.inv) {
            r.mapq = 0;
        } else if (r.parent == r.id) {
            int mapq;
            const float pen_s1 = (r.score > 100 ? 1.0f : 0.01f * r.score) * uniq_ratio;
            float pen_cm = r.cnt > 10 ? 1.0f : 0.1f * r.cnt;
            pen_cm = pen_s1 < pen_cm ? pen_s1 : pen_cm;
            const int subsc = r.subsc > min_chain_sc ? r.subsc : min_chain_sc;
            if ((r.has_p && (r.dp_max < 0 || r.dp_max >= ri.n_logf)) || r.score < 0 || r.score >= ri.n_logf || r.n_sub + 1 >= ri.n_logf) {
                *status |= PMX_ST_UNSUPPORTED;   // outside the host logf tables
                r.mapq = 0;
                continue;
            }
            if (r.has_p && r.dp_max2 > 0 && r.dp_max > 0) {
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
                } else {
                    mapq = (int)(pen_cm * q_coef * (1.0f - x) * ri.logf_int[r.score]);
                }
            }
            mapq -= (int)(4.343f * ri.logf_int[r.n_sub + 1] + .499f);
            mapq = mapq > 0 ? mapq : 0;
            r.mapq = (uint8_t)(mapq < 60 ? mapq : 60);
            if (r.has_p && r.dp_max > r.dp_max2 && r.mapq == 0) r.mapq = 1;
        } else r.mapq = 0;
    }
}

}  // namespace aln
}  // namespace pmx
