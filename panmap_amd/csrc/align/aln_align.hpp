// ALIGN stage, part 5: base-level alignment of one region through its anchors -- end fixing, bad-seed
// filtering, left extension / gap filling / right extension with ksw_extd2, z-drop handling, CIGAR
// clean-up and the per-alignment statistics.
// Reference behaviour: align.c:9-167 (scoring matrix, mm_test_zdrop, mm_fix_cigar), :240-343
// (mm_update_extra, mm_append_cigar, mm_align_pair), :355-526 (anchor adjust / filters / end fixing),
// :575-833 (mm_align1), :884-1027 (mm_update_dp_max, mm_align_skeleton).  Non-splice, non-SR,
// non-qstrand path (the option block of src/mm_align.c:118-188 never sets those flags).
#pragma once
#include "aln_chain.hpp"
#include "aln_hit.hpp"
#include "aln_ksw.hpp"
#include "aln_swll.hpp"
#include "aln_types.hpp"

namespace pmx {
namespace aln {

PMX_HD void gen_simple_mat(int8_t* mat, int8_t a, int8_t b, int8_t sc_ambi) {   // ksw_gen_simple_mat, m = 5
    a = a < 0 ? -a : a;
    b = b > 0 ? -b : b;
    sc_ambi = sc_ambi > 0 ? -sc_ambi : sc_ambi;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 4; ++j) mat[i * 5 + j] = i == j ? a : b;
        mat[i * 5 + 4] = sc_ambi;
    }
    for (int j = 0; j < 5; ++j) mat[4 * 5 + j] = sc_ambi;
}

// o.mat is always ksw_gen_simple_mat(a, b, sc_ambi) (make_opt): the per-base loops evaluate it arithmetically
// instead of loading a table entry per base
PMX_HD int simple_score(const Opt& o, uint32_t t, uint32_t q) {
    const int amb = o.sc_ambi > 0 ? -o.sc_ambi : o.sc_ambi;
    const int mis = o.b > 0 ? -o.b : o.b;
    const int mch = o.a < 0 ? -o.a : o.a;
    return (t > 3 || q > 3) ? amb : (t == q ? mch : mis);
}

PMX_HD void ref_getseq(const RefIndex& ri, int st, int en, Ptr<uint8_t> out) {   // mm_idx_getseq (index.c:152-162)
    PMX_LDS(out);
    if (en > ri.len) en = ri.len;
#if PMX_W == 1
    // scalar copy: aligned 32-bit loads from the reference, 32-bit stores while four bases are available
    GlobalByteReader src(ri.seq);
    Ptr<uint32_t> out4 = ptr_cast<uint32_t>(out);
    int i = st;
    for (; i + 4 <= en; i += 4) {
        const uint32_t wd = src[i] | src[i + 1] << 8 | src[i + 2] << 16 | src[i + 3] << 24;
        out4[(i - st) >> 2] = wd;
    }
    for (; i < en; ++i) out[i - st] = (uint8_t)src[i];
#else
    const int lane = lane_id();
    for (int i = st + lane; i < en; i += PMX_W) out[i - st] = ri.seq[i];
#endif
    wave_sync();
}

PMX_HD void seq_rev(int len, Ptr<uint8_t> seq) {   // mm_seq_rev
    PMX_LDS(seq);
    wave_sync();
    if (lane_id() == 0 || PMX_W == 1)
        for (int i = 0; i < len >> 1; ++i) { const uint8_t t = seq[i]; seq[i] = seq[len - 1 - i]; seq[len - 1 - i] = t; }
    wave_sync();
}

// CIGAR storage of a region: a slot in the per-wave pool
PMX_HD Ptr<uint32_t> reg_cigar(Work& W, const Reg& r) { return W.cig_pool + (size_t)r.cig_slot * W.caps.max_cigar; }

PMX_HD void reg_alloc_p(Work& W, Reg& r) {
    if (r.has_p) return;
    r.has_p = 1;
    r.dp_score = r.dp_max = r.dp_max2 = 0;
    r.n_ambi = 0;
    r.n_cigar = 0;
    if (W.cig_next < W.caps.n_cig_slots) r.cig_slot = (uint32_t)W.cig_next++;
    else { W.status |= PMX_ST_OVERFLOW; r.cig_slot = 0; }
}

// mm_append_cigar (align.c:291-314)
PMX_HD void append_cigar(Work& W, Reg& r, int n_cigar, Ptr<const uint32_t> cigar) {
    PMX_LDS(&W); PMX_LDS(&r); PMX_LDS(cigar);
    if (n_cigar == 0) return;
    reg_alloc_p(W, r);
    Ptr<uint32_t> c = reg_cigar(W, r); PMX_LDS(c);
    if ((int)r.n_cigar + n_cigar > W.caps.max_cigar) { W.status |= PMX_ST_OVERFLOW; return; }
    if (r.n_cigar > 0 && (c[r.n_cigar - 1] & 0xf) == (cigar[0] & 0xf)) {
        c[r.n_cigar - 1] += cigar[0] >> 4 << 4;
        for (int i = 1; i < n_cigar; ++i) c[r.n_cigar + i - 1] = cigar[i];
        r.n_cigar += n_cigar - 1;
    } else {
        for (int i = 0; i < n_cigar; ++i) c[r.n_cigar + i] = cigar[i];
        r.n_cigar += n_cigar;
    }
}

// mm_align_pair (align.c:316-343) without the splice / single-affine branches (q != q2 on this path)
PMX_HD void align_pair(Work& W, const Opt& o, int qlen, Ptr<const uint8_t> qseq, int tlen, Ptr<const uint8_t> tseq, int w, int end_bonus, int zdrop,
                       int flag, Ez& ez) {
    if (o.max_sw_mat > 0 && (int64_t)tlen * qlen > o.max_sw_mat) {
        ez_reset(ez);
        ez.zdropped = 1;
    } else if (o.q == o.q2 && o.e == o.e2) {
        W.status |= PMX_ST_UNSUPPORTED;   // ksw_extz2 (single affine) is not on this path's presets
        ez_reset(ez);
        ez.zdropped = 1;
    } else {
        ksw_extd2_auto(W, qlen, qseq, tlen, tseq, o.mat, (int8_t)o.q, (int8_t)o.e, (int8_t)o.q2, (int8_t)o.e2, w, zdrop, end_bonus, flag, ez);
    }
}

// Scalar execution models only (thread per pair, host): try the DP shortcuts with the target bases read straight
// from the reference (no copy into W.tseq, no in-place reversal for a left extension).  Same pre-checks as
// align_pair; false = nothing decided, take the regular path.
template <class QF, class TF>
PMX_HD bool try_shortcut_direct(Work& W, const Opt& o, int qlen, QF& qf, int tlen, TF& tf, int w, int end_bonus, int zdrop, int flag, Ez& ez) {
    if (o.max_sw_mat > 0 && (int64_t)tlen * qlen > o.max_sw_mat) return false;
    if (o.q == o.q2 && o.e == o.e2) return false;
    W.last_dp_shortcut = 0;
    if (!ksw_shortcut_f(W, qlen, qf, tlen, tf, o.mat, (int8_t)o.q, (int8_t)o.e, (int8_t)o.q2, (int8_t)o.e2, w, zdrop, end_bonus, flag, ez)) return false;
    W.last_dp_shortcut = 1;
    return true;
}

// update_max_zdrop + mm_test_zdrop (align.c:32-89).  The inversion probe -- a local alignment of the reverse complement
// of the query part of the largest score drop against its target part (ksw_ll_i16 -> sw_ll, align/aln_swll.hpp) -- decides
// between return codes 1 and 2; the thread-per-pair kernel has no DP scratch for it and hands the pair to the wave tiers.
PMX_HDN int test_zdrop(Work& W, const Opt& o, Ptr<const uint8_t> qseq, Ptr<const uint8_t> tseq, int n_cigar, Ptr<const uint32_t> cigar) {
    PMX_LDS(&W); PMX_LDS(qseq); PMX_LDS(tseq); PMX_LDS(cigar);
    int32_t score = 0, mx = INT32_MIN, max_i = -1, max_j = -1, i = 0, j = 0, max_zdrop = 0;
    int pos[2][2] = {{-1, -1}, {-1, -1}};
    auto upd = [&](int32_t sc, int ii, int jj) {
        if (sc < mx) {
            const int li = ii - max_i, lj = jj - max_j;
            const int diff = li > lj ? li - lj : lj - li;
            const int z = mx - sc - diff * o.e;
            if (z > max_zdrop) {
                max_zdrop = z;
                pos[0][0] = max_i; pos[0][1] = ii;
                pos[1][0] = max_j; pos[1][1] = jj;
            }
        } else { mx = sc; max_i = ii; max_j = jj; }
    };
    ByteReader q_r(qseq), t_r(tseq);
    for (int k = 0; k < n_cigar; ++k) {
        const uint32_t op = cigar[k] & 0xf, len = cigar[k] >> 4;
        if (op == 0) {
            for (uint32_t l = 0; l < len; ++l) {
                score += simple_score(o, t_r[i + (int)l], q_r[j + (int)l]);
                upd(score, i + (int)l, j + (int)l);
            }
            i += len; j += len;
        } else if (op == 1 || op == 2 || op == 3) {
            score -= o.q + o.e * (int)len;
            if (op == 1) j += len;
            else i += len;
            upd(score, i, j);
        }
    }
    const int q_len = pos[1][1] - pos[1][0], t_len = pos[0][1] - pos[0][0];
    if (max_zdrop > o.zdrop_inv && q_len < o.max_gap && t_len < o.max_gap) {
#if defined(PMX_THREAD_PER_PAIR)
        W.status |= PMX_ST_NEED_WAVE;
#else
        const int q_end = pos[1][1], t_beg = pos[0][0];
        auto qf = [&](int k) { const uint32_t c = q_r[q_end - k - 1]; return (int)(c >= 4 ? 4u : 3u - c); };
        auto tf = [&](int k) { return (int)t_r[t_beg + k]; };
        int q_off, t_off;
        bool ok;
        const int sc = sw_ll(W, o, q_len, qf, t_len, tf, &q_off, &t_off, &ok);
        if (!ok) W.status |= PMX_ST_OVERFLOW;   // DP scratch of this layout too small: the next tier's is not
        else if (sc >= o.min_chain_score * o.a && sc >= o.min_dp_max) return 2;   // there is a potential inversion
#endif
    }
    return max_zdrop > o.zdrop ? 1 : 0;
}

// mm_fix_cigar (align.c:91-167)
PMX_HDN void fix_cigar(Work& W, Reg& r, Ptr<const uint8_t> qseq, Ptr<const uint8_t> tseq, int* qshift, int* tshift) {
    PMX_LDS(&W); PMX_LDS(&r); PMX_LDS(qseq); PMX_LDS(tseq);
    Ptr<uint32_t> cg = reg_cigar(W, r); PMX_LDS(cg);
    int32_t toff = 0, qoff = 0, to_shrink = 0;
    *qshift = *tshift = 0;
    if (r.n_cigar <= 1) return;
    for (uint32_t k = 0; k < r.n_cigar; ++k) {   // indel left alignment
        const uint32_t op = cg[k] & 0xf, len = cg[k] >> 4;
        if (len == 0) to_shrink = 1;
        if (op == 0) { toff += len; qoff += len; }
        else if (op == 1 || op == 2) {
            if (k > 0 && k < r.n_cigar - 1 && (cg[k - 1] & 0xf) == 0 && (cg[k + 1] & 0xf) == 0) {
                int l;
                const int prev_len = (int)(cg[k - 1] >> 4);
                if (op == 1) {
                    for (l = 0; l < prev_len; ++l)
                        if (qseq[qoff - 1 - l] != qseq[qoff + len - 1 - l]) break;
                } else {
                    for (l = 0; l < prev_len; ++l)
                        if (tseq[toff - 1 - l] != tseq[toff + len - 1 - l]) break;
                }
                if (l > 0) { cg[k - 1] -= (uint32_t)l << 4; cg[k + 1] += (uint32_t)l << 4; qoff -= l; toff -= l; }
                if (l == prev_len) to_shrink = 1;
            }
            if (op == 1) qoff += len;
            else toff += len;
        } else if (op == 3) toff += len;
    }
    for (uint32_t k = 0; k + 2 < r.n_cigar; ++k) {   // fix CIGAR like 5I6D7I
        if ((cg[k] & 0xf) > 0 && (cg[k] & 0xf) + (cg[k + 1] & 0xf) == 3) {
            uint32_t l, s[3] = {0, 0, 0};
            for (l = k; l < r.n_cigar; ++l) {
                const uint32_t op = cg[l] & 0xf;
                if (op == 1 || op == 2 || cg[l] >> 4 == 0) s[op] += cg[l] >> 4;
                else break;
            }
            if (s[1] > 0 && s[2] > 0 && l - k > 2) {
                cg[k] = s[1] << 4 | 1;
                cg[k + 1] = s[2] << 4 | 2;
                for (k += 2; k < l; ++k) cg[k] &= 0xf;
                to_shrink = 1;
            }
            k = l;
        }
    }
    if (to_shrink) {
        int32_t l = 0;
        for (uint32_t k = 0; k < r.n_cigar; ++k)
            if (cg[k] >> 4 != 0) cg[l++] = cg[k];
        r.n_cigar = (uint32_t)l;
        l = 0;
        for (uint32_t k = 0; k < r.n_cigar; ++k) {
            if (k == r.n_cigar - 1 || (cg[k] & 0xf) != (cg[k + 1] & 0xf)) cg[l++] = cg[k];
            else cg[k + 1] += cg[k] >> 4 << 4;
        }
        r.n_cigar = (uint32_t)l;
    }
    if ((cg[0] & 0xf) == 1 || (cg[0] & 0xf) == 2) {   // leading I or D
        const int32_t l = (int32_t)(cg[0] >> 4);
        if ((cg[0] & 0xf) == 1) {
            if (r.rev) r.qe -= l;
            else r.qs += l;
            *qshift = l;
        } else { r.rs += l; *tshift = l; }
        --r.n_cigar;
        for (uint32_t k = 0; k < r.n_cigar; ++k) cg[k] = cg[k + 1];
    }
}

// mm_update_extra (align.c:240-289), log_gap = 1, is_eqx = 0
template <class TR>
PMX_HD void update_extra_core(Work& W, const Opt& o, Reg& r, Ptr<const uint8_t> qseq, TR& t_r, int8_t q, int8_t e) {
    PMX_LDS(&W); PMX_LDS(&r); PMX_LDS(qseq);
    int32_t toff = 0, qoff = 0;
    double s = 0.0, mx = 0.0;
    Ptr<const uint32_t> cg = reg_cigar(W, r); PMX_LDS(cg);
    ByteReader q_r(qseq);
    r.blen = r.mlen = 0;
    for (uint32_t k = 0; k < r.n_cigar; ++k) {
        const uint32_t op = cg[k] & 0xf, len = cg[k] >> 4;
        if (op == 0) {
            int n_ambi = 0, n_diff = 0;
            for (uint32_t l = 0; l < len; ++l) {
                const int cq = (int)q_r[qoff + (int)l], ct = (int)t_r[toff + (int)l];
                if (ct > 3 || cq > 3) ++n_ambi;
                else if (ct != cq) ++n_diff;
                s += simple_score(o, (uint32_t)ct, (uint32_t)cq);
                if (s < 0) s = 0;
                else mx = mx > s ? mx : s;
            }
            r.blen += len - n_ambi;
            r.mlen += len - (n_ambi + n_diff);
            r.n_ambi += n_ambi;
            toff += len; qoff += len;
        } else if (op == 1) {
            int n_ambi = 0;
            for (uint32_t l = 0; l < len; ++l)
                if (q_r[qoff + (int)l] > 3) ++n_ambi;
            r.blen += len - n_ambi;
            r.n_ambi += n_ambi;
            s -= q + (double)e * mg_log2f((float)(1.0 + len));
            if (s < 0) s = 0;
            qoff += len;
        } else if (op == 2) {
            int n_ambi = 0;
            for (uint32_t l = 0; l < len; ++l)
                if (t_r[toff + (int)l] > 3) ++n_ambi;
            r.blen += len - n_ambi;
            r.n_ambi += n_ambi;
            s -= q + (double)e * mg_log2f((float)(1.0 + len));
            if (s < 0) s = 0;
            toff += len;
        } else if (op == 3) toff += len;
    }
    r.dp_max = (int32_t)(mx + .499);
}

PMX_HDN void update_extra(Work& W, const Opt& o, Reg& r, Ptr<const uint8_t> qseq, Ptr<const uint8_t> tseq, int8_t q, int8_t e) {
    PMX_LDS(&W); PMX_LDS(&r); PMX_LDS(qseq); PMX_LDS(tseq);
    if (!r.has_p) return;
    int32_t qshift, tshift;
    fix_cigar(W, r, qseq, tseq, &qshift, &tshift);
    qseq += qshift;
    tseq += tshift;
    ByteReader t_r(tseq);
    update_extra_core(W, o, r, qseq, t_r, q, e);
}

// mm_adjust_minier (align.c:355-372), non-HPC
PMX_HD void adjust_minier(const Opt& o, const A128& a, int32_t* r, int32_t* q) {
    *r = (int32_t)a.x - (o.k >> 1);
    *q = (int32_t)a.y - (o.k >> 1);
}

PMX_HD int anchor_gap(Ptr<const A128> a, int i) {   // query advance minus reference advance between anchors i-1 and i
    return ((int32_t)a[i].y - (int32_t)a[i - 1].y) - ((int32_t)a[i].x - (int32_t)a[i - 1].x);
}

// collect_long_gaps + mm_filter_bad_seeds (align.c:374-427).  K[] lives in W.kidx.
PMX_HD int collect_long_gaps(Work& W, int as1, int cnt1, Ptr<const A128> a, int min_gap, Ptr<int32_t> K, int cap) {
    PMX_LDS(&W); PMX_LDS(a); PMX_LDS(K);
    int n = 0;
    for (int i = 1; i < cnt1; ++i) {
        const int gap = anchor_gap(a + as1, i);
        if (gap < -min_gap || gap > min_gap) ++n;
    }
    if (n <= 1) return 0;
    if (n > cap) { W.status |= PMX_ST_OVERFLOW; return 0; }
    n = 0;
    for (int i = 1; i < cnt1; ++i) {
        const int gap = anchor_gap(a + as1, i);
        if (gap < -min_gap || gap > min_gap) K[n++] = i;
    }
    return n;
}

PMX_HDN void filter_bad_seeds(Work& W, int as1, int cnt1, Ptr<A128> a, int min_gap, int diff_thres, int max_ext_len, int max_ext_cnt) {
    PMX_LDS(&W); PMX_LDS(a);
    Ptr<int32_t> K = W.kidx; PMX_LDS(K);
    const int n = collect_long_gaps(W, as1, cnt1, a, min_gap, K, W.caps.max_anchor);
    if (n == 0) return;
    int mx = 0, max_st = -1, max_en = -1;
    for (int k = 0;; ++k) {
        int gap, l, n_ins = 0, n_del = 0, qs, rs, max_diff = 0, max_diff_l = -1;
        if (k == n || k >= max_en) {
            if (max_en > 0)
                for (int i = K[max_st]; i < K[max_en]; ++i) a[as1 + i].y |= PMX_SEED_IGNORE;
            mx = 0; max_st = max_en = -1;
            if (k == n) break;
        }
        const int i = K[k];
        gap = ((int32_t)a[as1 + i].y - (int32_t)a[as1 + i - 1].y) - (int32_t)(a[as1 + i].x - a[as1 + i - 1].x);
        if (gap > 0) n_ins += gap;
        else n_del += -gap;
        qs = (int32_t)a[as1 + i - 1].y;
        rs = (int32_t)a[as1 + i - 1].x;
        for (l = k + 1; l < n && l <= k + max_ext_cnt; ++l) {
            const int j = K[l];
            if ((int32_t)a[as1 + j].y - qs > max_ext_len || (int32_t)a[as1 + j].x - rs > max_ext_len) break;
            gap = ((int32_t)a[as1 + j].y - (int32_t)a[as1 + j - 1].y) - (int32_t)(a[as1 + j].x - a[as1 + j - 1].x);
            if (gap > 0) n_ins += gap;
            else n_del += -gap;
            const int ad = n_ins - n_del < 0 ? n_del - n_ins : n_ins - n_del;
            const int diff = n_ins + n_del - ad;
            if (max_diff < diff) { max_diff = diff; max_diff_l = l; }
        }
        if (max_diff > diff_thres && max_diff > mx) { mx = max_diff; max_st = k; max_en = max_diff_l; }
    }
}

// mm_filter_bad_seeds_alt (align.c:429-462)
PMX_HDN void filter_bad_seeds_alt(Work& W, int as1, int cnt1, Ptr<A128> a, int min_gap, int max_ext) {
    PMX_LDS(&W); PMX_LDS(a);
    Ptr<int32_t> K = W.kidx; PMX_LDS(K);
    const int n = collect_long_gaps(W, as1, cnt1, a, min_gap, K, W.caps.max_anchor);
    if (n == 0) return;
    for (int k = 0; k < n;) {
        const int i = K[k];
        int l;
        int gap1 = ((int32_t)a[as1 + i].y - (int32_t)a[as1 + i - 1].y) - ((int32_t)a[as1 + i].x - (int32_t)a[as1 + i - 1].x);
        int re1 = (int32_t)a[as1 + i].x;
        int qe1 = (int32_t)a[as1 + i].y;
        gap1 = gap1 > 0 ? gap1 : -gap1;
        for (l = k + 1; l < n; ++l) {
            const int j = K[l];
            if ((int32_t)a[as1 + j].y - qe1 > max_ext || (int32_t)a[as1 + j].x - re1 > max_ext) break;
            int gap2 = ((int32_t)a[as1 + j].y - (int32_t)a[as1 + j - 1].y) - (int32_t)(a[as1 + j].x - a[as1 + j - 1].x);
            const int q_span_pre = (int)(a[as1 + j - 1].y >> 32 & 0xff);
            const int rs2 = (int32_t)a[as1 + j - 1].x + q_span_pre;
            const int qs2 = (int32_t)a[as1 + j - 1].y + q_span_pre;
            const int m = rs2 - re1 < qs2 - qe1 ? rs2 - re1 : qs2 - qe1;
            gap2 = gap2 > 0 ? gap2 : -gap2;
            if (m > gap1 + gap2) break;
            re1 = (int32_t)a[as1 + j].x;
            qe1 = (int32_t)a[as1 + j].y;
            gap1 = gap2;
        }
        if (l > k + 1) {
            const int end = K[l - 1];
            for (int j = K[k]; j < end; ++j) a[as1 + j].y |= PMX_SEED_IGNORE;
            a[as1 + end].y |= PMX_SEED_LONG_JOIN;
        }
        k = l;
    }
}

// mm_fix_bad_ends (align.c:464-502)
PMX_HDN void fix_bad_ends(const Reg& r, Ptr<const A128> a, int bw, int min_match, int32_t* as, int32_t* cnt) {
    PMX_LDS(&r); PMX_LDS(a);
    *as = r.as;
    *cnt = r.cnt;
    if (r.cnt < 3) return;
    int32_t m, l;
    m = l = (int32_t)(a[r.as].y >> 32 & 0xff);
    for (int32_t i = r.as + 1; i < r.as + r.cnt - 1; ++i) {
        const int32_t q_span = (int32_t)(a[i].y >> 32 & 0xff);
        if (a[i].y & PMX_SEED_LONG_JOIN) break;
        const int32_t lr = (int32_t)a[i].x - (int32_t)a[i - 1].x;
        const int32_t lq = (int32_t)a[i].y - (int32_t)a[i - 1].y;
        const int32_t mn = lr < lq ? lr : lq, mxv = lr > lq ? lr : lq;
        if (mxv - mn > l >> 1) *as = i;
        l += mn;
        m += mn < q_span ? mn : q_span;
        if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r.mlen >> 1) break;
    }
    *cnt = r.as + r.cnt - *as;
    m = l = (int32_t)(a[r.as + r.cnt - 1].y >> 32 & 0xff);
    for (int32_t i = r.as + r.cnt - 2; i > *as; --i) {
        const int32_t q_span = (int32_t)(a[i + 1].y >> 32 & 0xff);
        if (a[i + 1].y & PMX_SEED_LONG_JOIN) break;
        const int32_t lr = (int32_t)a[i + 1].x - (int32_t)a[i].x;
        const int32_t lq = (int32_t)a[i + 1].y - (int32_t)a[i].y;
        const int32_t mn = lr < lq ? lr : lq, mxv = lr > lq ? lr : lq;
        if (mxv - mn > l >> 1) *cnt = i + 1 - *as;
        l += mn;
        m += mn < q_span ? mn : q_span;
        if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r.mlen >> 1) break;
    }
}

// mm_align1 (align.c:575-833)
PMX_HDN void align1(Work& W, const Opt& o, const RefIndex& ri, int qlen, const Ptr<uint8_t>* qseq0, Reg& r, Reg& r2, int n_a, Ptr<A128> a, Ez& ez) {
    PMX_LDS(&W); PMX_LDS(&r); PMX_LDS(a);   // r2 and ez are caller-locals (private)
    Ptr<uint32_t> cig_tmp = W.cig_tmp; PMX_LDS(cig_tmp);
    const int32_t rev = (int32_t)(a[r.as].x >> 63);
    int32_t as1, cnt1;
    Ptr<uint8_t> tseq = W.tseq; PMX_LDS(tseq);
    int32_t l, dropped = 0, rs0, re0, qs0, qe0;
    int32_t rs, re, qs, qe;
    int32_t rs1, qs1, re1, qe1;
    const int32_t ref_len = ri.len;

    r2.cnt = 0;
    if (r.cnt == 0) return;
    const int bw = (int)(o.bw * 1.5 + 1.);
    int bw_long = (int)(o.bw_long * 1.5 + 1.);
    if (bw_long < bw) bw_long = bw;

    fix_bad_ends(r, a, o.bw, o.min_chain_score * 2, &as1, &cnt1);
    filter_bad_seeds(W, as1, cnt1, a, 10, 40, o.max_gap >> 1, 10);
    filter_bad_seeds_alt(W, as1, cnt1, a, 30, o.max_gap >> 1);
    adjust_minier(o, a[as1], &rs, &qs);
    adjust_minier(o, a[as1 + cnt1 - 1], &re, &qe);

    // region to align (align.c:636-691)
    rs0 = (int32_t)a[r.as].x + 1 - (int32_t)(a[r.as].y >> 32 & 0xff);
    qs0 = (int32_t)a[r.as].y + 1 - (int32_t)(a[r.as].y >> 32 & 0xff);
    if (rs0 < 0) rs0 = 0;
    rs1 = qs1 = 0;
    l = 0;
    for (int32_t i = r.as - 1; i >= 0 && a[i].x >> 32 == a[r.as].x >> 32; --i) {
        const int32_t x = (int32_t)a[i].x + 1 - (int32_t)(a[i].y >> 32 & 0xff);
        const int32_t y = (int32_t)a[i].y + 1 - (int32_t)(a[i].y >> 32 & 0xff);
        if (x < rs0 && y < qs0) {
            if (++l > o.min_cnt) {
                l = rs0 - x > qs0 - y ? rs0 - x : qs0 - y;
                rs1 = rs0 - l; qs1 = qs0 - l;
                if (rs1 < 0) rs1 = 0;
                break;
            }
        }
    }
    if (qs > 0 && rs > 0) {
        l = qs < o.max_gap ? qs : o.max_gap;
        qs1 = qs1 > qs - l ? qs1 : qs - l;
        qs0 = qs0 < qs1 ? qs0 : qs1;
        l += l * o.a > o.q ? (l * o.a - o.q) / o.e : 0;
        l = l < o.max_gap ? l : o.max_gap;
        l = l < rs ? l : rs;
        rs1 = rs1 > rs - l ? rs1 : rs - l;
        rs0 = rs0 < rs1 ? rs0 : rs1;
        rs0 = rs0 < rs ? rs0 : rs;
    } else { rs0 = rs; qs0 = qs; }
    re0 = (int32_t)a[r.as + r.cnt - 1].x + 1;
    qe0 = (int32_t)a[r.as + r.cnt - 1].y + 1;
    re1 = ref_len; qe1 = qlen;
    l = 0;
    for (int32_t i = r.as + r.cnt; i < n_a && a[i].x >> 32 == a[r.as].x >> 32; ++i) {
        const int32_t x = (int32_t)a[i].x + 1;
        const int32_t y = (int32_t)a[i].y + 1;
        if (x > re0 && y > qe0) {
            if (++l > o.min_cnt) {
                l = x - re0 > y - qe0 ? x - re0 : y - qe0;
                re1 = re0 + l; qe1 = qe0 + l;
                break;
            }
        }
    }
    if (qe < qlen && re < ref_len) {
        l = qlen - qe < o.max_gap ? qlen - qe : o.max_gap;
        qe1 = qe1 < qe + l ? qe1 : qe + l;
        qe0 = qe0 > qe1 ? qe0 : qe1;
        l += l * o.a > o.q ? (l * o.a - o.q) / o.e : 0;
        l = l < o.max_gap ? l : o.max_gap;
        l = l < ref_len - re ? l : ref_len - re;
        re1 = re1 < re + l ? re1 : re + l;
        re0 = re0 > re1 ? re0 : re1;
    } else { re0 = re; qe0 = qe; }
    if (re0 - rs0 > W.caps.max_tlen || re0 <= rs0) { W.status |= PMX_ST_OVERFLOW; return; }

    PMX_STAMP(W, 20);
    if (qs > 0 && rs > 0) {   // left extension (align.c:704-722)
        Ptr<uint8_t> qseq = qseq0[rev] + qs0; PMX_LDS(qseq);
        bool decided = false;
#if PMX_W == 1
        {
            RevBases<ByteReader> qf(ByteReader(Ptr<const uint8_t>(qseq)), qs - qs0);
            RevBases<GlobalByteReader> tf(GlobalByteReader(ri.seq + rs0), rs - rs0);
            decided = try_shortcut_direct(W, o, qs - qs0, qf, rs - rs0, tf, bw, o.end_bonus, r.split_inv ? o.zdrop_inv : o.zdrop,
                                          PMX_EZ_EXTZ_ONLY | PMX_EZ_RIGHT | PMX_EZ_REV_CIGAR, ez);
        }
#endif
        if (!decided) {
            ref_getseq(ri, rs0, rs, tseq);
            seq_rev(qs - qs0, qseq);
            seq_rev(rs - rs0, tseq);
            W.skip_shortcut = PMX_W == 1;
            align_pair(W, o, qs - qs0, qseq, rs - rs0, tseq, bw, o.end_bonus, r.split_inv ? o.zdrop_inv : o.zdrop,
                       PMX_EZ_EXTZ_ONLY | PMX_EZ_RIGHT | PMX_EZ_REV_CIGAR, ez);
            W.skip_shortcut = 0;
        }
        if (ez.n_cigar > 0) {
            append_cigar(W, r, ez.n_cigar, cig_tmp);
            r.dp_score += (int32_t)ez.max;
        }
        rs1 = rs - (ez.reach_end ? ez.mqe_t + 1 : ez.max_t + 1);
        qs1 = qs - (ez.reach_end ? qs - qs0 : ez.max_q + 1);
        if (!decided) seq_rev(qs - qs0, qseq);
    } else { rs1 = rs; qs1 = qs; }
    re1 = rs; qe1 = qs;
    PMX_STAMP(W, 21);

    // gap filling (align.c:727-797).  The reference walks the anchors one by one and fills whenever the stretch
    // since the last fill is long enough; here the walk is split into a cheap scan to the next anchor that closes a
    // fill and the fill itself, so that in the thread-per-pair kernel the lanes of a wave meet at the fill (with one
    // flat loop a wave would run the fill branch at nearly every anchor, for whichever lanes happen to fill there).
    for (int32_t i = 1; i < cnt1; ++i) {
        bool fill = false;
        uint64_t ay = 0;
        for (; i < cnt1; ++i) {
            const A128 ai = a[as1 + i];
            ay = ai.y;
            if ((ay & (PMX_SEED_IGNORE | PMX_SEED_TANDEM)) && i != cnt1 - 1) continue;
            adjust_minier(o, ai, &re, &qe);
            re1 = re; qe1 = qe;
            if (i == cnt1 - 1 || (ay & PMX_SEED_LONG_JOIN) || (qe - qs >= o.min_ksw_len && re - rs >= o.min_ksw_len)) { fill = true; break; }
        }
        if (!fill) break;
        {
            int bw1 = bw_long;
            if (ay & PMX_SEED_LONG_JOIN) bw1 = qe - qs > re - rs ? qe - qs : re - rs;
            Ptr<uint8_t> qseq = qseq0[rev] + qs; PMX_LDS(qseq);
            bool decided = false;
#if PMX_W == 1
            {
                FwdBases<ByteReader> qf{ByteReader(Ptr<const uint8_t>(qseq))};
                FwdBases<GlobalByteReader> tf{GlobalByteReader(ri.seq + rs)};
                decided = try_shortcut_direct(W, o, qe - qs, qf, re - rs, tf, bw1, -1, o.zdrop, PMX_EZ_APPROX_MAX, ez);
            }
#endif
            if (!decided) {
                ref_getseq(ri, rs, re, tseq);
                W.skip_shortcut = PMX_W == 1;
                align_pair(W, o, qe - qs, qseq, re - rs, tseq, bw1, -1, o.zdrop, PMX_EZ_APPROX_MAX, ez);   // first pass: approximate Z-drop
                W.skip_shortcut = 0;
            }
            // a gap fill answered by shortcut (2) is gap-free with at most three mismatches (d(a+b) <= a + 2*gmin): its
            // largest score drop is below 4(a+b) <= zdrop, so mm_test_zdrop returns 0 without looking
            const bool tz_skip = W.last_dp_shortcut && 4 * (o.a + o.b) <= o.zdrop && 4 * (o.a + o.b) <= o.zdrop_inv;
            if (!tz_skip && decided) ref_getseq(ri, rs, re, tseq);   // the shortcut read the reference directly
            const int zdrop_code = tz_skip ? 0 : test_zdrop(W, o, qseq, tseq, ez.n_cigar, cig_tmp);
            if (zdrop_code != 0) align_pair(W, o, qe - qs, qseq, re - rs, tseq, bw1, -1, zdrop_code == 2 ? o.zdrop_inv : o.zdrop, 0, ez);
            if (ez.n_cigar > 0) append_cigar(W, r, ez.n_cigar, cig_tmp);
            if (ez.zdropped) {   // truncated by Z-drop
                int32_t j;
                reg_alloc_p(W, r);
                for (j = i - 1; j >= 0; --j)
                    if ((int32_t)a[as1 + j].x <= rs + ez.max_t) break;
                dropped = 1;
                if (j < 0) j = 0;
                r.dp_score += (int32_t)ez.max;
                re1 = rs + (ez.max_t + 1);
                qe1 = qs + (ez.max_q + 1);
                if (cnt1 - (j + 1) >= o.min_cnt) {
                    split_reg(r, r2, as1 + j + 1 - r.as, qlen, a);
                    if (zdrop_code == 2) r2.split_inv = 1;
                }
                break;
            } else if (r.has_p) r.dp_score += ez.score;
            rs = re; qs = qe;
        }
    }

    PMX_STAMP(W, 22);
    if (!dropped && qe < qe0 && re < re0) {   // right extension (align.c:799-815)
        Ptr<uint8_t> qseq = qseq0[rev] + qe; PMX_LDS(qseq);
        bool decided = false;
#if PMX_W == 1
        {
            FwdBases<ByteReader> qf{ByteReader(Ptr<const uint8_t>(qseq))};
            FwdBases<GlobalByteReader> tf{GlobalByteReader(ri.seq + re)};
            decided = try_shortcut_direct(W, o, qe0 - qe, qf, re0 - re, tf, bw, o.end_bonus, o.zdrop, PMX_EZ_EXTZ_ONLY, ez);
        }
#endif
        if (!decided) {
            ref_getseq(ri, re, re0, tseq);
            W.skip_shortcut = PMX_W == 1;
            align_pair(W, o, qe0 - qe, qseq, re0 - re, tseq, bw, o.end_bonus, o.zdrop, PMX_EZ_EXTZ_ONLY, ez);
            W.skip_shortcut = 0;
        }
        if (ez.n_cigar > 0) {
            append_cigar(W, r, ez.n_cigar, cig_tmp);
            r.dp_score += (int32_t)ez.max;
        }
        re1 = re + (ez.reach_end ? ez.mqe_t + 1 : ez.max_t + 1);
        qe1 = qe + (ez.reach_end ? qe0 - qe : ez.max_q + 1);
    }

    PMX_STAMP(W, 23);
    r.rs = rs1; r.re = re1;
    if (!rev) { r.qs = qs1; r.qe = qe1; }
    else { r.qs = qlen - qe1; r.qe = qlen - qs1; }

    if (r.has_p) {
        Ptr<const uint8_t> qseq = qseq0[r.rev] + qs1; PMX_LDS(qseq);
#if PMX_W == 1
        if (r.n_cigar <= 1) {   // mm_fix_cigar is a no-op then: the statistics can scan the reference directly
            GlobalByteReader t_r(ri.seq + rs1);
            update_extra_core(W, o, r, qseq, t_r, (int8_t)o.q, (int8_t)o.e);
        } else
#endif
        {
            ref_getseq(ri, rs1, re1, tseq);
            update_extra(W, o, r, qseq, tseq, (int8_t)o.q, (int8_t)o.e);
        }
    }
}

// mm_event_identity / mm_recal_max_dp / mm_update_dp_max (align.c:918-965); only reached for qlen >= rank_min_len
PMX_HDN void update_dp_max(Work& W, int qlen, int n_regs, Reg* regs, float frac, int a, int b) {
    PMX_LDS(&W); PMX_LDS(regs);
    int32_t mx = -1, max2 = -1, max_i = -1;
    if (n_regs < 2) return;
    for (int i = 0; i < n_regs; ++i) {
        const Reg& r = regs[i];
        if (!r.has_p) continue;
        if (r.dp_max > mx) { max2 = mx; mx = r.dp_max; max_i = i; }
        else if (r.dp_max > max2) max2 = r.dp_max;
    }
    if (max_i < 0 || mx < 0 || max2 < 0) return;
    if (regs[max_i].qe - regs[max_i].qs < (double)qlen * frac) return;
    if (max2 < (double)mx * frac) return;
    auto count_gaps = [&](const Reg& r, int32_t* n_gap, int32_t* n_gapo) {
        Ptr<const uint32_t> cg = reg_cigar(W, r); PMX_LDS(cg);
        *n_gap = *n_gapo = 0;
        for (uint32_t i = 0; i < r.n_cigar; ++i) {
            const int32_t op = cg[i] & 0xf, len = cg[i] >> 4;
            if (op == 1 || op == 2) { ++*n_gapo; *n_gap += len; }
        }
    };
    int32_t n_gap, n_gapo;
    count_gaps(regs[max_i], &n_gap, &n_gapo);
    double div = 1. - (double)regs[max_i].mlen / (regs[max_i].blen + (int32_t)regs[max_i].n_ambi - n_gap + n_gapo);
    if (div < 0.02) div = 0.02;
    double b2 = 0.5 / div;
    if (b2 * a < b) b2 = (double)a / b;
    for (int i = 0; i < n_regs; ++i) {
        Reg& r = regs[i];
        if (!r.has_p) continue;
        Ptr<const uint32_t> cg = reg_cigar(W, r); PMX_LDS(cg);
        int32_t ng = 0;
        double gap_cost = 0.0;
        for (uint32_t q = 0; q < r.n_cigar; ++q) {
            const int32_t op = cg[q] & 0xf, len = cg[q] >> 4;
            if (op == 1 || op == 2) {
                gap_cost += b2 + (double)mg_log2f((float)(1.0 + len));
                ng += len;
            }
        }
        const int32_t n_mis = r.blen + (int32_t)r.n_ambi - r.mlen - ng;
        r.dp_max = (int32_t)(a * (r.mlen - b2 * n_mis - gap_cost) + .499);
        if (r.dp_max < 0) r.dp_max = 0;
    }
}

// mm_align1_inv (align.c:835-885): between the two parts of a region that was split at a suspected inversion, align the
// reverse strand of the query gap to the target gap -- the end of a local alignment of the reversed sequences fixes the
// start, an extension from there gives the alignment.  1 = r_inv holds an inversion hit.
PMX_HDN int align1_inv(Work& W, const Opt& o, const RefIndex& ri, int qlen, const Ptr<uint8_t>* qseq0, const Reg& r1, const Reg& r2, Reg& r_inv, Ez& ez) {
    PMX_LDS(&W); PMX_LDS(&r1); PMX_LDS(&r2); PMX_LDS(&r_inv);
    reg_clear(r_inv);
    if (!(r1.split & 1) || !(r2.split & 2)) return 0;
    if (r1.id != r1.parent && r1.parent != PMX_PARENT_TMP_PRI) return 0;
    if (r2.id != r2.parent && r2.parent != PMX_PARENT_TMP_PRI) return 0;
    if (r1.rev != r2.rev) return 0;   // (one reference sequence: the rid test always passes)
    const int ql = r1.rev ? r1.qs - r2.qe : r2.qs - r1.qe;
    const int tl = r2.rs - r1.re;
    if (ql < o.min_chain_score || ql > o.max_gap) return 0;
    if (tl < o.min_chain_score || tl > o.max_gap) return 0;
#if defined(PMX_THREAD_PER_PAIR)
    W.status |= PMX_ST_NEED_WAVE;
    return 0;
#else
    Ptr<uint8_t> tseq = W.tseq; PMX_LDS(tseq);
    ref_getseq(ri, r1.re, r2.rs, tseq);
    Ptr<uint8_t> qseq = r1.rev ? qseq0[0] + r2.qe : qseq0[1] + (qlen - r2.qs); PMX_LDS(qseq);
    int q_off, t_off;
    {
        ByteReader q_r{Ptr<const uint8_t>(qseq)}, t_r{Ptr<const uint8_t>(tseq)};
        auto qf = [&](int k) { return (int)q_r[ql - 1 - k]; };   // mm_seq_rev on both, undone afterwards
        auto tf = [&](int k) { return (int)t_r[tl - 1 - k]; };
        bool ok;
        const int score = sw_ll(W, o, ql, qf, tl, tf, &q_off, &t_off, &ok);
        if (!ok) { W.status |= PMX_ST_OVERFLOW; return 0; }
        if (score < o.min_dp_max) return 0;
    }
    q_off = ql - (q_off + 1);
    t_off = tl - (t_off + 1);
    // The local alignment may end on the padding behind the reversed query (aln_swll.hpp); q_off is then negative (down
    // to -7) and the reference's extension starts that many bases BEFORE the query gap, in the neighbouring region's
    // bases: the same here, as long as that stays inside this strand's copy of the read (it does unless a region ends
    // within 7 bases of the read's end, where the reference reads past its array)
    if (t_off < 0 || (r1.rev ? r2.qe : qlen - r2.qs) + q_off < 0) { W.status |= PMX_ST_UNSUPPORTED; return 0; }
    if (t_off > 0) ref_getseq(ri, r1.re + t_off, r2.rs, tseq);   // the DP reads its target from the start of W.tseq
    Ptr<uint32_t> cig_tmp = W.cig_tmp; PMX_LDS(cig_tmp);
    align_pair(W, o, ql - q_off, qseq + q_off, tl - t_off, tseq, (int)(o.bw * 1.5), -1, o.zdrop, PMX_EZ_EXTZ_ONLY, ez);
    if (W.status & PMX_ST_ABORT) return 0;
    if (ez.n_cigar == 0) return 0;
    append_cigar(W, r_inv, ez.n_cigar, cig_tmp);
    r_inv.dp_score = (int32_t)ez.max;
    r_inv.id = -1;
    r_inv.parent = PMX_PARENT_UNSET;
    r_inv.inv = 1;
    r_inv.rev = !r1.rev;
    r_inv.div = -1.0f;
    if (r_inv.rev == 0) {
        r_inv.qs = r2.qe + q_off;
        r_inv.qe = r_inv.qs + ez.max_q + 1;
    } else {
        r_inv.qe = r2.qs - q_off;
        r_inv.qs = r_inv.qe - (ez.max_q + 1);
    }
    r_inv.rs = r1.re + t_off;
    r_inv.re = r_inv.rs + ez.max_t + 1;
    update_extra(W, o, r_inv, qseq + q_off, tseq, (int8_t)o.q, (int8_t)o.e);
    return 1;
#endif
}

// mm_align_skeleton (align.c:967-1027) for one segment; then the tail of align_regs (map.c:225-234)
PMX_HDN void align_regs(Work& W, const Opt& o, const RefIndex& ri, int seg, int* n_regs_, Reg* regs, Ptr<A128> a) {
    PMX_LDS(&W); PMX_LDS(n_regs_); PMX_LDS(regs); PMX_LDS(a);
    const int qlen = W.qlen[seg];
    int n_regs = *n_regs_;
    const Ptr<uint8_t> qseq0[2] = {W.qseq[seg][0], W.qseq[seg][1]};
    Ez ez;
    const int n_a = squeeze_a(W, n_regs, regs, a);
    PMX_STAMP(W, 6);
    for (int i = 0; i < n_regs; ++i) {
#ifdef PMX_INTERLEAVED
        Reg& r2 = W.reg_tmp[0];   // (strided struct: lives in the arena, not on the stack; hit_sort uses reg_tmp later)
#else
        Reg r2;
#endif
        reg_clear(r2);
        align1(W, o, ri, qlen, qseq0, regs[i], r2, n_a, a, ez);
        if (r2.cnt > 0) {   // mm_insert_reg
            if (n_regs + 1 > W.caps.max_reg) { W.status |= PMX_ST_OVERFLOW; }
            else {
                for (int j = n_regs - 1; j > i; --j) regs[j + 1] = regs[j];
                regs[i + 1] = r2;
                ++n_regs;
            }
        }
        if (i > 0 && regs[i].split_inv && !(W.status & PMX_ST_ABORT)) {
            if (align1_inv(W, o, ri, qlen, qseq0, regs[i - 1], regs[i], r2, ez)) {
                if (n_regs + 1 > W.caps.max_reg) { W.status |= PMX_ST_OVERFLOW; }
                else {
                    for (int j = n_regs - 1; j > i; --j) regs[j + 1] = regs[j];
                    regs[i + 1] = r2;
                    ++n_regs;
                    ++i;   // skip the inserted inversion hit
                }
            }
        }
        if (W.status & PMX_ST_NEED_WAVE) return;   // thread-per-pair kernel: this pair is re-run by the wave kernel
        // (after a posted DP request the loop goes on: the other regions post theirs in the same pass, ksw_extd2_auto)
    }
    if (W.status & PMX_ST_ABORT) return;   // nothing below is meaningful on neutral DP results
    PMX_STAMP(W, 7);
    filter_regs(o, qlen, &n_regs, regs);
    if (qlen >= o.rank_min_len) {
        update_dp_max(W, qlen, n_regs, regs, o.rank_frac, o.a, o.b);
        filter_regs(o, qlen, &n_regs, regs);
    }
    hit_sort(W, &n_regs, regs);
    // map.c:229-233
    set_parent(W, o.mask_level, o.mask_len, n_regs, regs, o.a * 2 + o.b);
    select_sub(W, o.pri_ratio, o.k * 2, o.best_n, 0, (int)(o.max_gap * 0.8), &n_regs, regs);
    set_sam_pri(n_regs, regs);
    *n_regs_ = n_regs;
    PMX_STAMP(W, 8);
}

}  // namespace aln
}  // namespace pmx
