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

// ---- CIGAR operations: BAM encoding, length << 4 | kind (M = 0, I = 1, D = 2) ---------------------------------------------
enum : uint32_t { CG_M = 0, CG_I = 1, CG_D = 2 };
PMX_HD uint32_t cg_kind(uint32_t c) { return c & 0xfu; }
PMX_HD uint32_t cg_len(uint32_t c) { return c >> 4; }
PMX_HD bool cg_is_indel(uint32_t kind) { return kind == CG_I || kind == CG_D; }

// Appends one operation to a list under construction: nothing for an empty operation, an operation of the kind the list
// ends with lengthens that entry.  `n` never exceeds the number of operations pushed, so a list can be rewritten in place.
PMX_HD void cg_push(Ptr<uint32_t> c, uint32_t& n, uint32_t kind, uint32_t len) {
    if (len == 0) return;
    if (n > 0 && cg_kind(c[n - 1]) == kind) c[n - 1] += len << 4;
    else c[n++] = len << 4 | kind;
}

// A region takes the operations of one more DP (what mm_append_cigar does, align.c:291-314).  Every DP's list comes out of
// the backtrack without empty operations and without two neighbours of one kind, so pushing every operation joins exactly
// the one pair the reference joins: the region's last operation with the DP's first.
PMX_HD void append_cigar(Work& W, Reg& r, int n_new, Ptr<const uint32_t> ops) {
    PMX_LDS(&W); PMX_LDS(&r); PMX_LDS(ops);
    if (n_new == 0) return;
    reg_alloc_p(W, r);
    if ((int)r.n_cigar + n_new > W.caps.max_cigar) { W.status |= PMX_ST_OVERFLOW; return; }
    Ptr<uint32_t> c = reg_cigar(W, r); PMX_LDS(c);
    uint32_t n = r.n_cigar;
#if PMX_W == 64 && defined(__HIP_DEVICE_COMPILE__)
    // (wave kernels: the list of a long read's region lives in the wave's HBM slab -- the one join by hand, the rest 64
    //  operations per round trip instead of a dependent read-modify-write each)
    wave_sync();
    const uint32_t first = ops[0];
    int from = 0;
    if (cg_len(first) == 0) from = 1;   // (never out of the backtrack; kept for the scalar form's sake)
    else if (n > 0 && cg_kind(c[n - 1]) == cg_kind(first)) {
        if (lane_id() == 0) c[n - 1] += cg_len(first) << 4;
        from = 1;
    }
    for (int i = from + lane_id(); i < n_new; i += 64) c[n + (uint32_t)(i - from)] = ops[i];
    n += (uint32_t)(n_new - from);
    wave_sync();
#else
    for (int i = 0; i < n_new; ++i) cg_push(c, n, cg_kind(ops[i]), cg_len(ops[i]));
#endif
    r.n_cigar = n;
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

// The largest score drop along an alignment, and whether it hides an inversion (what update_max_zdrop + mm_test_zdrop
// decide, align.c:32-89): 0 = fine, 1 = the drop exceeds zdrop, 2 = a local alignment of the reverse complement of the
// query part of the drop against its target part scores like a chain (ksw_ll_i16 -> sw_ll, align/aln_swll.hpp).
//
// The reference re-scores the alignment base by base and tests the drop after every base.  Stated over RUNS here: along a
// run of matching bases the running score only rises, so
//   * while it is still below the best score so far, every drop it could report is smaller than the one reported at the
//     event that ended the previous run (same diagonal offset, lower score there);
//   * once it reaches the best score it stays the best to the end of the run: the run moves the best-score mark to its
//     last base, or (when it never gets there) changes nothing.
// Only the bases that do not match -- and the gaps -- are events that are evaluated one by one.
struct DropScan {
    int32_t score = 0, best = INT32_MIN, best_t = -1, best_q = -1, worst = 0;
    int32_t t_from = -1, t_to = -1, q_from = -1, q_to = -1;   // the stretch of the largest drop
    PMX_HD void event(int32_t t, int32_t q, int gap_ext) {     // the running score was changed at (t, q)
        if (score >= best) { best = score; best_t = t; best_q = q; return; }
        const int32_t dt = t - best_t, dq = q - best_q;
        const int32_t drop = best - score - (dt > dq ? dt - dq : dq - dt) * gap_ext;
        if (drop > worst) { worst = drop; t_from = best_t; t_to = t; q_from = best_q; q_to = q; }
    }
    PMX_HD void match_run(int32_t t_last, int32_t q_last, int32_t gain) {   // `gain` > 0 over a run that ends at (t_last, q_last)
        score += gain;
        if (score >= best) { best = score; best_t = t_last; best_q = q_last; }
    }
};

// Length of the run of equal, unambiguous base pairs at the start of t[t0 .. t0+len) / q[q0 .. q0+len): the position of the
// first pair that differs or whose target base is ambiguous (`len` if there is none); ct / cq receive the bases there.
// Wave-per-read kernels: 64 pairs per round trip, the first bad one from a ballot -- the walks over a long read's alignment
// (mm_test_zdrop per gap fill, mm_update_extra over the whole CIGAR: ~20,000 bases per 10 kb read, each a dependent byte load
// from the wave's HBM slab in the scalar form) then cost a round trip per mismatch instead of one per base.
template <class TRD, class QRD>
PMX_HD int32_t equal_run(TRD& t_r, QRD& q_r, int32_t t0, int32_t q0, int32_t len, uint32_t& ct, uint32_t& cq) {
#if PMX_W == 64 && defined(__HIP_DEVICE_COMPILE__)
    const int lane = lane_id();
    for (int32_t base = 0; base < len; base += 64) {
        const int32_t i = base + lane;
        uint32_t a = 0, b = 0;
        bool bad = false;
        if (i < len) { a = t_r[t0 + i]; b = q_r[q0 + i]; bad = a != b || a > 3; }
        const unsigned long long m = __ballot(bad);
        if (m != 0ULL) {
            const int p = __builtin_ctzll(m);
            ct = (uint32_t)__builtin_amdgcn_readlane((int)a, p);
            cq = (uint32_t)__builtin_amdgcn_readlane((int)b, p);
            return base + p;
        }
    }
    return len;
#else
    int32_t run = 0;
    for (; run < len; ++run) {
        ct = t_r[t0 + run]; cq = q_r[q0 + run];
        if (ct != cq || ct > 3) break;
    }
    return run;
#endif
}

PMX_HDN int test_zdrop(Work& W, const Opt& o, Ptr<const uint8_t> qseq, Ptr<const uint8_t> tseq, int n_ops, Ptr<const uint32_t> ops) {
    PMX_LDS(&W); PMX_LDS(qseq); PMX_LDS(tseq); PMX_LDS(ops);
    DropScan ds;
    ByteReader q_r(qseq), t_r(tseq);
    const int match = o.a < 0 ? -o.a : o.a;
    int32_t t = 0, q = 0;   // target / query bases consumed
    for (int k = 0; k < n_ops; ++k) {
        const uint32_t kind = cg_kind(ops[k]);
        const int32_t len = (int32_t)cg_len(ops[k]);
        if (kind == CG_M) {
            int32_t at = 0;
            while (at < len) {
                uint32_t ct = 0, cq = 0;
                const int32_t run = equal_run(t_r, q_r, t + at, q + at, len - at, ct, cq);
                if (run > 0) ds.match_run(t + at + run - 1, q + at + run - 1, run * match);
                at += run;
                if (at < len) {   // a mismatch or an ambiguous base
                    ds.score += simple_score(o, ct, cq);
                    ds.event(t + at, q + at, o.e);
                    ++at;
                }
            }
            t += len; q += len;
        } else if (kind == CG_I || kind == CG_D || kind == 3) {
            ds.score -= o.q + o.e * len;
            if (kind == CG_I) q += len;
            else t += len;
            ds.event(t, q, o.e);
        }
    }
    const int q_len = ds.q_to - ds.q_from, t_len = ds.t_to - ds.t_from;
    if (ds.worst > o.zdrop_inv && q_len < o.max_gap && t_len < o.max_gap) {
#if defined(PMX_THREAD_PER_PAIR)
        W.status |= PMX_ST_NEED_WAVE;   // no DP scratch for the inversion probe in this kernel: the wave tiers take the pair
#else
        const int q_end = ds.q_to, t_beg = ds.t_from;
        auto qf = [&](int k) { const uint32_t c = q_r[q_end - k - 1]; return (int)(c >= 4 ? 4u : 3u - c); };
        auto tf = [&](int k) { return (int)t_r[t_beg + k]; };
        int q_off, t_off;
        bool ok;
        const int sc = sw_ll(W, o, q_len, qf, t_len, tf, &q_off, &t_off, &ok);
        if (!ok) W.status |= PMX_ST_OVERFLOW;   // DP scratch of this layout too small: the next tier's is not
        else if (sc >= o.min_chain_score * o.a && sc >= o.min_dp_max) return 2;   // there is a potential inversion
#endif
    }
    return ds.worst > o.zdrop ? 1 : 0;
}

// How far a gap of `gap` bases that starts at `at` can slide to the left over `room` aligned bases: one step for every base
// before the gap that equals the base `gap` positions later (the alignment stays the same alignment).
template <class SEQ>
PMX_HD int32_t gap_slide_left(SEQ& s, int32_t at, int32_t gap, int32_t room) {
    int32_t n = 0;
    while (n < room && s[at - 1 - n] == s[at + gap - 1 - n]) ++n;
    return n;
}

// The CIGAR of a finished region in its final form (what mm_fix_cigar produces, align.c:91-167):
//  1. every insertion / deletion between two match blocks moves as far left as the sequences allow;
//  2. a stretch of insertions and deletions that follow each other directly (three operations or more, empty ones in
//     between do not interrupt it once it has begun) becomes one insertion + one deletion with the summed lengths;
//  3. empty operations go, neighbours of one kind join;
//  4. a leading insertion / deletion is cut off and the region's start moves instead (*qshift / *tshift say by how much).
// Steps 2 and 3 are one in-place rewrite here: the list is re-emitted through cg_push, which drops empty operations and
// joins equal neighbours as it goes (the reference decides per list whether to compact; a list it leaves alone has
// nothing to drop or join, so compacting always gives the same list).
PMX_HDN void fix_cigar(Work& W, Reg& r, Ptr<const uint8_t> qseq, Ptr<const uint8_t> tseq, int* qshift, int* tshift) {
    PMX_LDS(&W); PMX_LDS(&r); PMX_LDS(qseq); PMX_LDS(tseq);
    Ptr<uint32_t> c = reg_cigar(W, r); PMX_LDS(c);
    *qshift = *tshift = 0;
    const uint32_t n_in = r.n_cigar;
    if (n_in <= 1) return;
    {   // 1. left alignment; (tpos, qpos) = bases consumed before operation k
        int32_t tpos = 0, qpos = 0;
        for (uint32_t k = 0; k < n_in; ++k) {
            const uint32_t kind = cg_kind(c[k]);
            const int32_t len = (int32_t)cg_len(c[k]);
            if (kind == CG_M) { tpos += len; qpos += len; continue; }
            if (!cg_is_indel(kind)) { if (kind == 3) tpos += len; continue; }
            if (k > 0 && k + 1 < n_in && cg_kind(c[k - 1]) == CG_M && cg_kind(c[k + 1]) == CG_M) {
                const int32_t room = (int32_t)cg_len(c[k - 1]);
                const int32_t by = kind == CG_I ? gap_slide_left(qseq, qpos, len, room) : gap_slide_left(tseq, tpos, len, room);
                c[k - 1] -= (uint32_t)by << 4;     // (may become empty: step 3 removes it)
                c[k + 1] += (uint32_t)by << 4;
                qpos -= by; tpos -= by;
            }
            if (kind == CG_I) qpos += len;
            else tpos += len;
        }
    }
    uint32_t n_out = 0;
    for (uint32_t k = 0; k < n_in;) {   // 2. + 3.
        const uint32_t kind = cg_kind(c[k]);
        // a stretch begins where an insertion and a deletion touch (and at least one more operation follows)
        if (k + 2 < n_in && kind != CG_M && kind + cg_kind(c[k + 1]) == CG_I + CG_D) {
            uint32_t end = k, sum[3] = {0, 0, 0};
            for (; end < n_in; ++end) {
                const uint32_t kd = cg_kind(c[end]), ln = cg_len(c[end]);
                if (!cg_is_indel(kd) && ln != 0) break;
                if (kd < 3) sum[kd] += ln;          // (an empty operation of any kind adds nothing)
            }
            if (sum[CG_I] > 0 && sum[CG_D] > 0 && end - k > 2) {
                cg_push(c, n_out, CG_I, sum[CG_I]);
                cg_push(c, n_out, CG_D, sum[CG_D]);
            } else {
                for (uint32_t j = k; j < end; ++j) cg_push(c, n_out, cg_kind(c[j]), cg_len(c[j]));
            }
            if (end < n_in) cg_push(c, n_out, cg_kind(c[end]), cg_len(c[end]));   // the operation that ended the stretch starts none
            k = end + 1;
        } else {
            cg_push(c, n_out, kind, cg_len(c[k]));
            ++k;
        }
    }
    if (n_out > 0 && cg_is_indel(cg_kind(c[0]))) {   // 4.
        const int32_t cut = (int32_t)cg_len(c[0]);
        if (cg_kind(c[0]) == CG_I) {
            if (r.rev) r.qe -= cut;
            else r.qs += cut;
            *qshift = cut;
        } else {
            r.rs += cut;
            *tshift = cut;
        }
        --n_out;
        for (uint32_t k = 0; k < n_out; ++k) c[k] = c[k + 1];
    }
    r.n_cigar = n_out;
}

// The statistics of a finished alignment (what mm_update_extra computes, align.c:240-289; log_gap = 1, is_eqx = 0):
// aligned / matching / ambiguous bases and dp_max, the largest value of a running score that cannot fall below zero.
// Over RUNS as in test_zdrop: along a run of matching bases the score rises from a value >= 0, so its maximum over the
// run is its value at the end of the run; a base that does not match, and a gap, are single events.  (The running score
// is a double like the reference's; every value it takes is a multiple of 2^-24 below 2^20 -- integers and
// q + e * (a float) -- so all its additions are exact and a run added at once equals its bases added one by one.)
template <class TR>
PMX_HD void update_extra_core(Work& W, const Opt& o, Reg& r, Ptr<const uint8_t> qseq, TR& t_r, int8_t q, int8_t e) {
    PMX_LDS(&W); PMX_LDS(&r); PMX_LDS(qseq);
    Ptr<const uint32_t> c = reg_cigar(W, r); PMX_LDS(c);
    ByteReader q_r(qseq);
    const int match = o.a < 0 ? -o.a : o.a;
    double run_score = 0.0, peak = 0.0;
    int32_t t = 0, qp = 0, aligned = 0, same = 0, ambiguous = 0;
    auto ambiguous_in = [&](auto& rd, int32_t from, int32_t len) { int n = 0; for (int32_t i = 0; i < len; ++i) n += rd[from + i] > 3; return n; };
    for (uint32_t k = 0; k < r.n_cigar; ++k) {
        const uint32_t kind = cg_kind(c[k]);
        const int32_t len = (int32_t)cg_len(c[k]);
        if (kind == CG_M) {
            int32_t at = 0, n_amb = 0, n_sub = 0;
            while (at < len) {
                uint32_t ct = 0, cq = 0;
                const int32_t run = equal_run(t_r, q_r, t + at, qp + at, len - at, ct, cq);
                if (run > 0) {
                    run_score += (double)(run * match);
                    peak = peak > run_score ? peak : run_score;
                    at += run;
                }
                if (at < len) {
                    if (ct > 3 || cq > 3) ++n_amb; else ++n_sub;
                    run_score += simple_score(o, ct, cq);
                    if (run_score < 0) run_score = 0;
                    else peak = peak > run_score ? peak : run_score;
                    ++at;
                }
            }
            aligned += len - n_amb;
            same += len - n_amb - n_sub;
            ambiguous += n_amb;
            t += len; qp += len;
        } else if (cg_is_indel(kind)) {
            const int n_amb = kind == CG_I ? ambiguous_in(q_r, qp, len) : ambiguous_in(t_r, t, len);
            aligned += len - n_amb;
            ambiguous += n_amb;
            run_score -= q + (double)e * mg_log2f((float)(1.0 + len));
            if (run_score < 0) run_score = 0;
            if (kind == CG_I) qp += len;
            else t += len;
        } else if (kind == 3) t += len;
    }
    r.blen = aligned;
    r.mlen = same;
    r.n_ambi += ambiguous;
    r.dp_max = (int32_t)(peak + .499);
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

// ---- the anchors of one region as the clean-up passes see them -------------------------------------------------------------
// Anchor i of the view ends at reference base tpos(i) and query base qpos(i) and spans span(i) bases; skew(i) is by how much
// the step from anchor i-1 to anchor i leaves the diagonal (> 0: the query advances further than the reference).
template <class AP>
struct AnchorViewT {
    AP a;
    int first;
    PMX_HD int32_t tpos(int i) const { return (int32_t)a[first + i].x; }
    PMX_HD int32_t qpos(int i) const { return (int32_t)a[first + i].y; }
    PMX_HD int32_t span(int i) const { return (int32_t)(a[first + i].y >> 32 & 0xff); }
    PMX_HD int32_t skew(int i) const { return (qpos(i) - qpos(i - 1)) - (tpos(i) - tpos(i - 1)); }
    PMX_HD bool joined(int i) const { return (a[first + i].y & PMX_SEED_LONG_JOIN) != 0; }
    PMX_HD void flag(int i, uint64_t bits) { a[first + i].y |= bits; }
};
typedef AnchorViewT<Ptr<A128>> AnchorView;
typedef AnchorViewT<Ptr<const A128>> ConstAnchorView;

// mm_adjust_minier (align.c:355-372), non-HPC: the middle of the anchor's k-mer
PMX_HD void adjust_minier(const Opt& o, const A128& a, int32_t* r, int32_t* q) {
    *r = (int32_t)a.x - (o.k >> 1);
    *q = (int32_t)a.y - (o.k >> 1);
}

// The steps of the chain that leave the diagonal by more than `tol` bases, as indices into the view (W.kidx); fewer than two
// of them: nothing for the filters below to do (0 is returned).
PMX_HD int skewed_steps(Work& W, const AnchorView& v, int n_anchors, int tol, Ptr<int32_t> out, int cap) {
    PMX_LDS(&W); PMX_LDS(out);
    int n = 0;
    bool full = false;
    for (int i = 1; i < n_anchors; ++i) {
        const int s = v.skew(i);
        if (s >= -tol && s <= tol) continue;
        if (n < cap) out[n] = i;
        else full = true;
        ++n;
    }
    if (n <= 1) return 0;
    if (full) { W.status |= PMX_ST_OVERFLOW; return 0; }
    return n;
}

// Anchors inside a stretch where insertions and deletions cancel each other are unreliable (what mm_filter_bad_seeds drops,
// align.c:391-427).  From every skewed step a window of at most `max_steps` further skewed steps (and `max_reach` bases) is
// scanned; its weight is the largest amount of insertion that is matched by deletion, 2 * min(inserted, deleted), over
// the prefixes of the window, and the window is cut where that maximum is first reached.  Windows heavier than
// `min_weight` compete: of those that begin before the current winner ends, the heaviest one wins (the earlier one on a
// tie); the anchors of a winner, from its first skewed step up to (not including) its last, are marked IGNORE.
PMX_HDN void filter_bad_seeds(Work& W, int as1, int cnt1, Ptr<A128> a, int tol, int min_weight, int max_reach, int max_steps) {
    PMX_LDS(&W); PMX_LDS(a);
    Ptr<int32_t> step = W.kidx; PMX_LDS(step);
    AnchorView v{a, as1};
    const int n = skewed_steps(W, v, cnt1, tol, step, W.caps.max_anchor);
    if (n == 0) return;
    int win_from = -1, win_to = -1, win_weight = 0;     // the current winner, as positions in step[]
    auto settle = [&]() {
        if (win_to > 0)
            for (int i = step[win_from]; i < step[win_to]; ++i) v.flag(i, PMX_SEED_IGNORE);
        win_from = win_to = -1;
        win_weight = 0;
    };
    for (int k = 0; k < n; ++k) {
        if (k >= win_to) settle();
        const int i0 = step[k];
        int32_t ins = 0, del = 0;
        { const int s = v.skew(i0); if (s > 0) ins = s; else del = -s; }
        const int32_t q_from = v.qpos(i0 - 1), t_from = v.tpos(i0 - 1);
        int weight = 0, cut = -1;
        for (int m = k + 1; m < n && m <= k + max_steps; ++m) {
            const int j = step[m];
            if (v.qpos(j) - q_from > max_reach || v.tpos(j) - t_from > max_reach) break;
            const int s = v.skew(j);
            if (s > 0) ins += s; else del -= s;
            const int cancelled = 2 * (ins < del ? ins : del);
            if (cancelled > weight) { weight = cancelled; cut = m; }
        }
        if (weight > min_weight && weight > win_weight) { win_weight = weight; win_from = k; win_to = cut; }
    }
    settle();
}

// Skewed steps that lie close together are bridged (what mm_filter_bad_seeds_alt does, align.c:429-462): starting from a
// skewed step, the next one joins the bridge while it begins within `max_reach` bases and the diagonal stretch between the
// two is no longer than the two skews together.  The anchors under a bridge are marked IGNORE and its last anchor
// LONG_JOIN: the gap filler aligns the whole bridge in one DP whose band is the bridge's length.
PMX_HDN void filter_bad_seeds_alt(Work& W, int as1, int cnt1, Ptr<A128> a, int tol, int max_reach) {
    PMX_LDS(&W); PMX_LDS(a);
    Ptr<int32_t> step = W.kidx; PMX_LDS(step);
    AnchorView v{a, as1};
    const int n = skewed_steps(W, v, cnt1, tol, step, W.caps.max_anchor);
    int k = 0;
    while (k < n) {
        int last = k;                                   // the bridge covers step[k..last]
        int32_t t_end = v.tpos(step[k]), q_end = v.qpos(step[k]);
        int32_t skew_here = v.skew(step[k]);
        if (skew_here < 0) skew_here = -skew_here;
        for (int m = k + 1; m < n; ++m) {
            const int j = step[m];
            if (v.qpos(j) - q_end > max_reach || v.tpos(j) - t_end > max_reach) break;
            int32_t skew_next = v.skew(j);
            if (skew_next < 0) skew_next = -skew_next;
            // the diagonal stretch between the bridge's end and the anchor before j (its span belongs to the stretch's end)
            const int32_t t_gap = v.tpos(j - 1) + v.span(j - 1) - t_end, q_gap = v.qpos(j - 1) + v.span(j - 1) - q_end;
            if ((t_gap < q_gap ? t_gap : q_gap) > skew_here + skew_next) break;
            t_end = v.tpos(j); q_end = v.qpos(j);
            skew_here = skew_next;
            last = m;
        }
        if (last > k) {
            for (int i = step[k]; i < step[last]; ++i) v.flag(i, PMX_SEED_IGNORE);
            v.flag(step[last], PMX_SEED_LONG_JOIN);
        }
        k = last + 1;
    }
}

// The ends of a chain are trimmed where they run off the diagonal (what mm_fix_bad_ends does, align.c:464-502).  Both ends
// are walked inwards over neighbouring anchors lo < hi with one rule: a step whose reference and query advance differ by
// more than half of the bases covered so far moves the end to the step's inner anchor; the walk stops at a LONG_JOIN mark
// and once enough bases were covered or matched (two band widths covered, `min_match` and a band width matched, or half of
// the region's matching bases).
struct EndWalk {
    int32_t covered, matched;
    PMX_HD explicit EndWalk(int32_t span0) : covered(span0), matched(span0) {}
    // the step lo -> hi (= lo + 1); the span that counts is the later anchor's.  true: this step runs off the diagonal
    template <class V> PMX_HD bool off_diagonal(const V& v, int lo, int hi) const {
        const int32_t dt = v.tpos(hi) - v.tpos(lo), dq = v.qpos(hi) - v.qpos(lo);
        return (dt > dq ? dt - dq : dq - dt) > covered >> 1;
    }
    template <class V> PMX_HD void take(const V& v, int lo, int hi) {
        const int32_t dt = v.tpos(hi) - v.tpos(lo), dq = v.qpos(hi) - v.qpos(lo);
        const int32_t adv = dt < dq ? dt : dq, sp = v.span(hi);
        covered += adv;
        matched += adv < sp ? adv : sp;
    }
    PMX_HD bool enough(int bw, int min_match, int32_t region_matches) const {
        return covered >= bw << 1 || (matched >= min_match && matched >= bw) || matched >= region_matches >> 1;
    }
};

PMX_HDN void fix_bad_ends(const Reg& r, Ptr<const A128> a, int bw, int min_match, int32_t* as, int32_t* cnt) {
    PMX_LDS(&r); PMX_LDS(a);
    *as = r.as;
    *cnt = r.cnt;
    if (r.cnt < 3) return;
    const ConstAnchorView v{a, r.as};
    const int n = r.cnt;
    int first = 0;                            // the kept anchors are [first, first + kept) of the view
    {
        EndWalk w(v.span(0));
        for (int hi = 1; hi < n - 1; ++hi) {
            if (v.joined(hi)) break;
            if (w.off_diagonal(v, hi - 1, hi)) first = hi;
            w.take(v, hi - 1, hi);
            if (w.enough(bw, min_match, r.mlen)) break;
        }
    }
    int kept = n - first;
    {
        EndWalk w(v.span(n - 1));
        for (int lo = n - 2; lo > first; --lo) {
            if (v.joined(lo + 1)) break;
            if (w.off_diagonal(v, lo, lo + 1)) kept = lo + 1 - first;
            w.take(v, lo, lo + 1);
            if (w.enough(bw, min_match, r.mlen)) break;
        }
    }
    *as = r.as + first;
    *cnt = kept;
}

// How far beyond the chain's outermost (adjusted) anchor an end extension may reach, on the query and on the reference
// (the window computation of mm_align1, align.c:636-691, written once for both ends in DISTANCES from that anchor):
//   room_q / room_t   bases left on the read / the reference beyond the anchor
//   fence_q / fence_t where neighbouring chains on the same strand begin to own the bases (the room if there are none)
//   own_q / own_t     where the region's own outermost anchor ends (it may lie beyond the adjusted one)
// The query reach is max_gap at most; the reference reach adds the bases a gap-free extension of that length could pay for
// with gap extensions, capped by max_gap again.  Neither crosses the fence; the region's own anchor is always inside.
struct Reach { int32_t q, t; };
PMX_HD Reach extension_reach(const Opt& o, int32_t room_q, int32_t room_t, Reach fence, Reach own) {
    int32_t len = room_q < o.max_gap ? room_q : o.max_gap;
    if (fence.q > len) fence.q = len;
    if (own.q < fence.q) own.q = fence.q;
    if (len * o.a > o.q) len += (len * o.a - o.q) / o.e;
    if (len > o.max_gap) len = o.max_gap;
    if (len > room_t) len = room_t;
    if (fence.t > len) fence.t = len;
    if (own.t < fence.t) own.t = fence.t;
    return own;
}

// mm_align1 (align.c:575-833)
PMX_HDN void align1(Work& W, const Opt& o, const RefIndex& ri, int qlen, const Ptr<uint8_t>* qseq0, Reg& r, Reg& r2, int n_a, Ptr<A128> a, Ez& ez) {
    PMX_LDS(&W); PMX_LDS(&r); PMX_LDS(a);   // r2 and ez are caller-locals (private)
    Ptr<uint32_t> cig_tmp = W.cig_tmp; PMX_LDS(cig_tmp);
    const int32_t rev = (int32_t)(a[r.as].x >> 63);
    int32_t as1, cnt1;
    Ptr<uint8_t> tseq = W.tseq; PMX_LDS(tseq);
    int32_t dropped = 0, rs0, re0, qs0, qe0;
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

    // The window the end extensions may use (align.c:636-691), in distances from the adjusted outer anchors (rs, qs) and
    // (re, qe): extension_reach.  A chain of the same strand fences an extension off once more than min_cnt of its anchors lie
    // wholly beyond the region's own first (last) anchor: the fence stands as far out as the nearest such anchor's far corner.
    {
        const uint64_t strand_rid = a[r.as].x >> 32;
        const int32_t own_t0 = (int32_t)a[r.as].x + 1 - (int32_t)(a[r.as].y >> 32 & 0xff);
        const int32_t own_q0 = (int32_t)a[r.as].y + 1 - (int32_t)(a[r.as].y >> 32 & 0xff);
        const int32_t own_t = own_t0 < 0 ? 0 : own_t0;
        if (qs > 0 && rs > 0) {
            Reach fence{qs, rs};                                  // no neighbour: the start of the read / the reference
            int seen = 0;
            for (int32_t i = r.as - 1; i >= 0 && a[i].x >> 32 == strand_rid; --i) {
                const int32_t sp = (int32_t)(a[i].y >> 32 & 0xff), t_i = (int32_t)a[i].x + 1 - sp, q_i = (int32_t)a[i].y + 1 - sp;
                if (t_i >= own_t || q_i >= own_q0) continue;
                if (++seen > o.min_cnt) {
                    const int32_t back = own_t - t_i > own_q0 - q_i ? own_t - t_i : own_q0 - q_i;
                    fence.q = qs - (own_q0 - back);
                    fence.t = rs - (own_t - back);
                    break;
                }
            }
            const Reach rch = extension_reach(o, qs, rs, fence, Reach{qs - own_q0, rs - own_t});
            qs0 = qs - rch.q;
            rs0 = rs - (rch.t < 0 ? 0 : rch.t);                  // (never inside the adjusted anchor)
        } else { rs0 = rs; qs0 = qs; }
        const int32_t own_t1 = (int32_t)a[r.as + r.cnt - 1].x + 1, own_q1 = (int32_t)a[r.as + r.cnt - 1].y + 1;
        if (qe < qlen && re < ref_len) {
            Reach fence{qlen - qe, ref_len - re};
            int seen = 0;
            for (int32_t i = r.as + r.cnt; i < n_a && a[i].x >> 32 == strand_rid; ++i) {
                const int32_t t_i = (int32_t)a[i].x + 1, q_i = (int32_t)a[i].y + 1;
                if (t_i <= own_t1 || q_i <= own_q1) continue;
                if (++seen > o.min_cnt) {
                    const int32_t ahead = t_i - own_t1 > q_i - own_q1 ? t_i - own_t1 : q_i - own_q1;
                    fence.q = own_q1 + ahead - qe;
                    fence.t = own_t1 + ahead - re;
                    break;
                }
            }
            const Reach rch = extension_reach(o, qlen - qe, ref_len - re, fence, Reach{own_q1 - qe, own_t1 - re});
            qe0 = qe + rch.q;
            re0 = re + rch.t;
        } else { re0 = re; qe0 = qe; }
    }
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
