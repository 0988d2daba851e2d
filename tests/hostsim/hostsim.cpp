// CPU unit-test build of the ALIGN pipeline sources (panmap_amd/csrc/align/*.hpp) with PMX_W = 1.
// TEST INFRASTRUCTURE: lets the host logic of the kernels be checked against the compiled reference
// aligner in a container without a GPU.  Never linked into libpanmap_amd.so; the product path is the
// HIP kernel in panmap_amd/csrc/align_kernel.hip.
#include <cstdio>
#include <cstdlib>
#include <vector>

#ifdef PMX_HOSTSIM_TPP
// emulation of the thread-per-pair kernel's DP service (align_kernel_tpp.hip + k_align_dp_serve): the pipeline
// posts a request instead of running a DP, the request is served by ksw_extd2, the pair is replayed.
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { const unsigned long long o = *p; *p += v; return o; }
#endif
#include "align/aln_host.hpp"
#ifndef PMX_HOSTSIM_TPP
// work counters of the compact tier (0: anchors, 1: anchor pairs the chain fill evaluated, 2: run skips, 3: pairs whose seeds the two-way merge ordered)
static long long pmx_c_cnt[4];
#define PMX_C_DUMP 1
#define PMX_C_COUNT(k, v) (pmx_c_cnt[k] += (v))
// per-anchor trip counts of the chain fill (inner loop, max_ii rescan) of the pair being processed: hs_compact_trace
static int pmx_c_trace_on = 0, pmx_c_trace_n = 0;
static unsigned char pmx_c_trace_trips[64], pmx_c_trace_rescan[64];
#define PMX_C_TRACE(i, trips, rescan) do { if (pmx_c_trace_on && (i) < 64) { pmx_c_trace_trips[i] = (unsigned char)((trips) > 255 ? 255 : (trips)); pmx_c_trace_rescan[i] = (unsigned char)((rescan) > 255 ? 255 : (rescan)); pmx_c_trace_n = (i) + 1; } } while (0)
#include "align/aln_compact.hpp"
// trace[pair][0] = anchors, [1..64] trips, [65..128] rescans (set the buffer before hs_align_compact; NULL = off)
static unsigned char* pmx_c_trace_out = nullptr;
extern "C" void hs_compact_trace(unsigned char* out) { pmx_c_trace_out = out; pmx_c_trace_on = out != nullptr; }
extern "C" void hs_compact_counts(long long* out, int reset) { for (int i = 0; i < 4; ++i) { out[i] = pmx_c_cnt[i]; if (reset) pmx_c_cnt[i] = 0; } }
#endif

using namespace pmx::aln;

extern "C" int hs_align(const char* ref, int64_t ref_len, int n_reads, const char** reads, const int* lens, int paired, AlnRecord* recs,
                        uint32_t* cigars, int64_t cig_cap, int64_t* cig_used, int verbose) {
    const bool edits_in_score = getenv("PMX_HS_EDITS") != nullptr;
    int64_t total = 0;
    int max_len = 0;
    for (int i = 0; i < n_reads; ++i) { total += lens[i]; max_len = std::max(max_len, lens[i]); }
    const int avg_len = n_reads > 0 ? (int)(total / n_reads) : 150;
    Opt o = make_opt(avg_len);
    HostRefIndex hri;
    build_ref_index(ref, ref_len, o, std::max(4096, max_len * (o.a + 1) * 2 + 64), hri);
    const RefIndex ri = hri.view();
    const int n_segs = paired ? 2 : 1;
#ifdef PMX_HOSTSIM_TPP
    Layout L = plan_layout_compact(max_len, n_segs, o);
    L.slow_bytes += 64;
    const Layout Lserve = L;
    L.tb_cap = 0;   // like api_align.hip: no in-lane DPs, every DP is a request
    std::vector<uint8_t> dp_req(sizeof(DpReq) * PMX_DP_REQ_PER_PASS);
    std::vector<DpRes> dp_res(PMX_DP_MAX_CALLS);
    int64_t n_requests = 0, n_wave = 0;
#else
    // long reads: like api_align.hip, DPs of up to PMX_DP_FAST_TLEN bases run on the small copy of the DP arrays (PMX_HS_DP_FAST overrides)
    const int dp_fast = getenv("PMX_HS_DP_FAST") ? atoi(getenv("PMX_HS_DP_FAST")) : (o.is_sr_like ? 0 : PMX_DP_FAST_TLEN);
    Layout L = plan_layout(max_len, n_segs, o, (size_t)1 << 30, 0, 1, dp_fast);
#endif
    std::vector<uint8_t> fast(L.fast_bytes + 64), slow(L.slow_bytes + 64);
    Work W;
    memset(&W, 0, sizeof(W));
    bind_work(W, L, fast.data(), slow.data());
    if (verbose) fprintf(stderr, "hostsim: k=%d w=%d mid_occ=%d fast=%zu slow=%zu max_tlen=%d\n", o.k, o.w, o.mid_occ, L.fast_bytes, L.slow_bytes, L.caps.max_tlen);
    const int n_items = paired ? n_reads / 2 : n_reads;
    int64_t used = 0;
    const char* poison = getenv("PMX_HS_POISON");   // fill the work arrays before every item: results must not depend on what a slab held
    for (int it = 0; it < n_items; ++it) {
        if (poison) {
            const int pat = (int)strtol(poison, nullptr, 0);
            memset(fast.data(), pat, fast.size());
            memset(slow.data(), pat, slow.size());
        }
        W.n_segs = n_segs;
        W.status = 0;
        bool too_long = false;
        for (int s = 0; s < n_segs; ++s) {
            const int idx = paired ? 2 * it + s : it;
            const int len = lens[idx];
            if (len > L.caps.max_qlen) { too_long = true; break; }
            W.qlen[s] = len;
            for (int i = 0; i < len; ++i) {
                const uint8_t c = nt4_of_char((unsigned char)reads[idx][i]);
                W.qseq[s][0][i] = c;
                W.qseq[s][1][len - 1 - i] = c < 4 ? 3 - c : 4;
            }
        }
        if (too_long) return -1;
#ifdef PMX_HOSTSIM_TPP
        int n_cached = 0;
        for (;;) {   // one thread-per-pair pass, then the DP service for every request it posted, then the replay
            const int ql[2] = {W.qlen[0], W.qlen[1]};
            bind_work(W, L, fast.data(), slow.data());
            W.qlen[0] = ql[0]; W.qlen[1] = ql[1];
            W.dp_req_base = dp_req.data();
            W.dp_res = dp_res.data();
            W.dp_slot = 0;
            W.dp_slot_cap = 1;
            W.dp_n_cached = n_cached;
            W.dp_post_end = n_cached;
            for (int j = 0; j < PMX_DP_REQ_PER_PASS; ++j) reinterpret_cast<DpReq*>(dp_req.data() + (size_t)j * sizeof(DpReq))->call = 0xffffffffu;
            map_frag(W, o, ri);
            if (W.status & PMX_ST_NEED_DP) W.status = W.status_pre | PMX_ST_NEED_DP;   // (align_kernel_tpp.hip)
            if ((W.status & (PMX_ST_OVERFLOW | PMX_ST_NEED_WAVE)) || !(W.status & PMX_ST_NEED_DP)) break;
            if (W.dp_post_end <= n_cached) return -4;
            for (int j = 0; j < PMX_DP_REQ_PER_PASS; ++j) {
                const DpReq* rq = reinterpret_cast<const DpReq*>(dp_req.data() + (size_t)j * sizeof(DpReq));
                if (rq->call == 0xffffffffu) continue;
                if ((int)rq->call != n_cached + j) return -4;
                if (getenv("PMX_HS_DUMP")) fprintf(stderr, "REQ %d %d %d %d\n", rq->qlen, rq->tlen, rq->w, rq->flag);
                Work W2;
                memset(&W2, 0, sizeof(W2));
                bind_work(W2, Lserve, fast.data(), slow.data());
                std::vector<uint8_t> seq(rq->seq, rq->seq + PMX_DP_SEQ_BYTES);
                Ez ez;
                ksw_extd2(W2, rq->qlen, seq.data(), rq->tlen, seq.data() + ((rq->qlen + 15) & ~15), o.mat, (int8_t)o.q, (int8_t)o.e, (int8_t)o.q2,
                          (int8_t)o.e2, rq->w, rq->zdrop, rq->end_bonus, rq->flag, ez);
                DpRes& R = dp_res[rq->call];
                const bool bad = (W2.status & PMX_ST_OVERFLOW) || ez.n_cigar > PMX_DP_MAX_CIGAR;
                R.ez = ez;
                R.key = bad ? 0xffffffffu : rq->key;
                if (!bad) for (int i = 0; i < ez.n_cigar; ++i) R.cigar[i] = W2.cig_tmp[i];
                ++n_requests;
            }
            n_cached = W.dp_post_end;
        }
        if (W.status & (PMX_ST_OVERFLOW | PMX_ST_NEED_WAVE)) {   // this pair goes to the wave-per-pair tiers
            ++n_wave;
            for (int s = 0; s < n_segs; ++s) {
                AlnRecord& rec = recs[paired ? 2 * it + s : it];
                memset(&rec, 0, sizeof(rec));
                rec.flags = 0x8000;
            }
            continue;
        }
#else
        map_frag(W, o, ri);
#endif
        const bool mapped = frag_is_mapped(W, paired);
        if (getenv("PMX_HS_DUMP")) {
            int off = 0;
            for (int c = 0; c < W.n_u; ++c) {
                fprintf(stderr, "chain %d: score=%d cnt=%d:", c, (int)(W.u[c] >> 32), (int)(uint32_t)W.u[c]);
                for (int j = 0; j < (int)(uint32_t)W.u[c]; ++j)
                    fprintf(stderr, " [x=%d|%llu y: seg=%d q=%d span=%d]", (int)(W.a[off + j].x >> 63), (unsigned long long)(W.a[off + j].x & 0xffffffffULL),
                            (int)((W.a[off + j].y & PMX_SEED_SEG_MASK) >> PMX_SEED_SEG_SHIFT), (int)(int32_t)W.a[off + j].y, (int)(W.a[off + j].y >> 32 & 0xff));
                fprintf(stderr, "\n");
                off += (int)(uint32_t)W.u[c];
            }
        }
        if (getenv("PMX_HS_DUMP"))
            for (int s = 0; s < n_segs; ++s) {
                fprintf(stderr, "item %d seg %d: n_regs=%d n_mv=%d status=%u\n", it, s, W.n_regs[s], W.n_mv, (unsigned)W.status);
                for (int j = 0; j < W.n_regs[s]; ++j) {
                    const Reg& r = W.regs[s][j];
                    fprintf(stderr, "   reg %d: cnt=%d score=%d q=[%d,%d) r=[%d,%d) rev=%d mlen=%d blen=%d dp_max=%d has_p=%d parent=%d subsc=%d mapq=%d\n", j, r.cnt, r.score, r.qs,
                            r.qe, r.rs, r.re, r.rev, r.mlen, r.blen, r.dp_max, (int)r.has_p, r.parent, r.subsc, r.mapq);
                }
            }
        for (int s = 0; s < n_segs; ++s) {
            AlnRecord& rec = recs[paired ? 2 * it + s : it];
            memset(&rec, 0, sizeof(rec));
            rec.flags = (uint16_t)(W.status & 3);
            if (edits_in_score) rec.score = read_errors(W, s);   // (debug aid: the --refine edit count instead of dp_max)
            if (!mapped) continue;
            rec.mapped = 1;
            const Reg& r = W.regs[s][0];
            if (!r.has_p) continue;
            rec.flags |= PMX_REC_HAS_ALN;
            rec.rs = r.rs; rec.re = r.re; rec.qs = r.qs; rec.qe = r.qe;
            rec.mapq = r.mapq; rec.rev = r.rev; rec.proper_frag = r.proper_frag;
            rec.n_cigar = (uint16_t)r.n_cigar;
            if (!edits_in_score) rec.score = r.dp_max;
            rec.cigar_off = (uint32_t)used;
            if (used + r.n_cigar > cig_cap) return -2;
            const uint32_t* cg = reg_cigar(W, r);
            for (uint32_t i = 0; i < r.n_cigar; ++i) cigars[used + i] = cg[i];
            used += r.n_cigar;
        }
    }
    *cig_used = used;
#ifdef PMX_HOSTSIM_TPP
    if (verbose) fprintf(stderr, "hostsim tpp: %lld DP requests, %lld pairs to the wave tier of %d\n", (long long)n_requests, (long long)n_wave, n_items);
#endif
    return 0;
}

#ifndef PMX_HOSTSIM_TPP
// The compact tier (align/aln_compact.hpp) on the host: every pair goes through compact_map_pair with a plain array as
// its "LDS"; done[i] = 1 when the tier finished pair i (its records are then final), 0 when it bailed (the general
// tiers would run it).  rc2 != 0: mate 2 arrives in FASTQ orientation and is reverse-complemented on the fly.
extern "C" int hs_align_compact(const char* ref, int64_t ref_len, int n_reads, const char** reads, const int* lens, int rc2, AlnRecord* recs,
                                uint32_t* cigars, int8_t* done) {
    int64_t total = 0;
    int max_len = 0;
    for (int i = 0; i < n_reads; ++i) { total += lens[i]; max_len = std::max(max_len, lens[i]); }
    const int avg_len = n_reads > 0 ? (int)(total / n_reads) : 150;
    Opt o = make_opt(avg_len);
    HostRefIndex hri;
    build_ref_index(ref, ref_len, o, std::max(4096, max_len * (o.a + 1) * 2 + 64), hri);
    const RefIndex ri = hri.view();
    std::vector<uint32_t> lds(CMemT<uint32_t>::kWords + 8);
    std::vector<uint8_t> pen(PMX_C_PEN_BYTES);
    for (int dd = 0; dd < PMX_C_PEN_DIFF; ++dd) {
        int ps, pd;
        c_pen_values(o.chn_pen_gap, dd, &ps, &pd);
        if (dd < PMX_C_PEN_SAME) pen[(size_t)dd] = (uint8_t)(ps < 255 ? ps : 255);
        pen[(size_t)PMX_C_PEN_SAME + dd] = (uint8_t)(pd < 255 ? pd : 255);
    }
    CPenTab tab{pen.data(), pen.data() + PMX_C_PEN_SAME};
    if (getenv("PMX_HS_COMPACT_NO_TAB")) tab.same = tab.diff = nullptr;
    for (int it = 0; it < n_reads / 2; ++it) {
        std::vector<uint64_t> w[2];
        std::vector<uint32_t> am[2];
        CRead rd[2];
        const uint32_t* amb[2];
        for (int s = 0; s < 2; ++s) {
            const int idx = 2 * it + s, len = lens[idx];
            w[s].assign((size_t)(len + 31) / 32 + 1, 0);
            am[s].assign((size_t)(len + 31) / 32 + 1, 0);
            for (int i = 0; i < len; ++i) {   // k_pack_reads (place_kernels.hip): A C G T -> 0 1 2 3, anything else ambiguous
                const unsigned char ch = (unsigned char)reads[idx][i] & 0xDF;
                const bool acgt = ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T';
                const uint64_t code = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : (ch == 'T' || ch == 'U') ? 3 : 0;
                w[s][(size_t)(i >> 5)] |= code << (2 * (i & 31));
                if (!acgt) am[s][(size_t)(i >> 5)] |= 1u << (i & 31);
            }
            rd[s].w = w[s].data();
            rd[s].len = len;
            rd[s].flip = rc2 && s == 1;
            amb[s] = am[s].data();
        }
        std::fill(lds.begin(), lds.end(), 0xdeadbeefu);
        CResult res;
        int rc;
        const bool pos16 = ref_len <= 32767 && !getenv("PMX_HS_COMPACT_POS32");
        const bool multi = getenv("PMX_HS_COMPACT_MULTI") != nullptr;   // several regions per mate (k_align_compact*_multi)
        SReg mregs[PMX_CM_SREGS];
        uint32_t mints[PMX_CM_INTS];
        const SWork mw{mregs, mints};
        if (getenv("PMX_HS_COMPACT_SPLIT")) {
            // the two-kernel form: seeds through the hand-over words (k_compact_seeds), then the chain part from a copy
            // of them in the pair's block (k_align_compact) -- the block itself never sees the sketch
            std::vector<uint32_t> q(2 * PMX_C_SEEDQ, 0xdeadbeefu), ho((size_t)PMX_C_CAP * 2, 0xdeadbeefu);
            auto split = [&](auto pt) {
                typedef decltype(pt) PT;
                uint32_t stage[8];
                CSeedOutT<PT> so{q.data(), ho.data(), stage};
                int n_s = 0, n_s0 = 0;
                res.mapped = 0;
                if (compact_seed_pair(so, o, ri, rd, amb, &n_s, &n_s0) != PMX_C_DONE) return (int)PMX_C_BAIL;
                // (the first form of the chain kernel keeps PMX_C_CAP1 anchors -- more seeds than that: the second form's pair)
                if (!multi) {
                    if (n_s > PMX_C_CAP1) return (int)PMX_C_BAIL;
                    CMemT<PT, PMX_C_CAP1> m1{lds.data()};
                    for (int i0 = 0; i0 < n_s; i0 += 4) {
                        uint32_t x[4], y[4];
                        so.get4(i0, x, y);
                        for (int b = 0; b < 4 && i0 + b < n_s; ++b) m1.setSeed(i0 + b, x[b], y[b]);
                    }
                    return compact_chain_pair(m1, o, ri, rd, n_s, n_s0, res, tab);
                }
                CMemT<PT> m{lds.data()};
                for (int i0 = 0; i0 < n_s; i0 += 4) {   // (as the kernels copy them: four seeds per request)
                    uint32_t x[4], y[4];
                    so.get4(i0, x, y);
                    for (int b = 0; b < 4 && i0 + b < n_s; ++b) m.setSeed(i0 + b, x[b], y[b]);
                }
                return multi ? compact_chain_pair<true>(m, o, ri, rd, n_s, n_s0, res, tab, nullptr, false, false, &mw) : compact_chain_pair(m, o, ri, rd, n_s, n_s0, res, tab);
            };
            rc = pos16 ? split((uint16_t)0) : split((uint32_t)0);
        } else if (pos16) { CMemT<uint16_t> m{lds.data()}; rc = multi ? compact_map_pair<true>(m, o, ri, rd, amb, res, tab, nullptr, false, false, &mw) : compact_map_pair(m, o, ri, rd, amb, res, tab); }
        else { CMemT<uint32_t> m{lds.data()}; rc = multi ? compact_map_pair<true>(m, o, ri, rd, amb, res, tab, nullptr, false, false, &mw) : compact_map_pair(m, o, ri, rd, amb, res, tab); }
        done[it] = rc == PMX_C_DONE ? 1 : 0;
        if (pmx_c_trace_out) {
            unsigned char* t = pmx_c_trace_out + (size_t)it * 129;
            t[0] = (unsigned char)pmx_c_trace_n;
            memcpy(t + 1, pmx_c_trace_trips, 64);
            memcpy(t + 65, pmx_c_trace_rescan, 64);
        }
        pmx_c_trace_n = 0;
        for (int s = 0; s < 2; ++s) {
            AlnRecord& rec = recs[2 * it + s];
            memset(&rec, 0, sizeof(rec));
            cigars[2 * it + s] = 0;
            if (rc != PMX_C_DONE || !res.mapped) continue;
            const CMate& t = res.m[s];
            rec.mapped = 1;
            rec.flags = PMX_REC_HAS_ALN;
            rec.rs = t.rs; rec.re = t.re; rec.qs = t.qs; rec.qe = t.qe;
            rec.mapq = t.mapq; rec.rev = t.rev; rec.proper_frag = t.proper_frag;
            rec.n_cigar = 1;
            rec.score = t.dp_max;
            rec.cigar_off = (uint32_t)(2 * it + s);
            cigars[2 * it + s] = t.cigar;
        }
    }
    return 0;
}

// Property check of the DP shortcuts: random extension / gap-fill problems (small alphabets provoke repeats, which
// is where a gapped path can rival the gap-free one); whenever ksw_shortcut answers, the full DP must give the
// same consumed fields (max, max_t, max_q, mqe_t when reach_end, reach_end, score, zdropped, CIGAR).
static uint64_t fz_state;
static inline uint32_t fz_rand() {
    fz_state ^= fz_state << 13; fz_state ^= fz_state >> 7; fz_state ^= fz_state << 17;
    return (uint32_t)(fz_state >> 11);
}
extern "C" int hs_shortcut_fuzz(uint64_t seed, int64_t n_cases, int64_t* counts /* declined, agreed, mismatched */, int verbose) {
    fz_state = seed * 0x9E3779B97F4A7C15ULL + 1;
    Opt o = make_opt(150);
    gen_simple_mat(o.mat, (int8_t)o.a, (int8_t)o.b, (int8_t)o.sc_ambi);
    Layout L = plan_layout(256, 2, o, (size_t)1 << 30);
    std::vector<uint8_t> fast(L.fast_bytes + 64), slow(L.slow_bytes + 64);
    Work W;
    memset(&W, 0, sizeof(W));
    bind_work(W, L, fast.data(), slow.data());
    std::vector<uint32_t> cig1(64);
    counts[0] = counts[1] = counts[2] = 0;
    uint8_t q[256], t[256];
    for (int64_t it = 0; it < n_cases; ++it) {
        const int mode = (int)(fz_rand() % 3);            // 0 right extension, 1 left extension, 2 gap fill (first pass)
        const int alpha = 2 + (int)(fz_rand() % 3);       // 2..4 letters
        const int qlen = 3 + (int)(fz_rand() % 70);
        const int period = (fz_rand() & 3) == 0 ? 1 + (int)(fz_rand() % 6) : 0;   // sometimes a tandem repeat
        int tlen = mode == 2 ? qlen : qlen + (int)(fz_rand() % 40);
        for (int i = 0; i < tlen; ++i) t[i] = (uint8_t)(period && i >= period ? t[i - period] : fz_rand() % alpha);
        for (int i = 0; i < qlen; ++i) q[i] = t[i];
        const int n_mut = (int)(fz_rand() % 4);
        for (int m = 0; m < n_mut; ++m) {
            const int p = (int)(fz_rand() % qlen);
            q[p] = (uint8_t)((q[p] + 1 + fz_rand() % 3) % 4);
        }
        if ((fz_rand() & 63) == 0) q[fz_rand() % qlen] = 4;   // an ambiguous base now and then
        if ((fz_rand() & 63) == 0) t[fz_rand() % tlen] = 4;
        const int flag = mode == 0 ? PMX_EZ_EXTZ_ONLY : mode == 1 ? (PMX_EZ_EXTZ_ONLY | PMX_EZ_RIGHT | PMX_EZ_REV_CIGAR) : PMX_EZ_APPROX_MAX;
        const int end_bonus = mode == 2 ? -1 : o.end_bonus, w = (int)(o.bw * 1.5 + 1.);
        Ez e1, e2;
        W.status = 0;
        const bool took = ksw_shortcut(W, qlen, q, tlen, t, o.mat, (int8_t)o.q, (int8_t)o.e, (int8_t)o.q2, (int8_t)o.e2, w, o.zdrop, end_bonus, flag, e1);
        if (!took) { ++counts[0]; continue; }
        for (int i = 0; i < e1.n_cigar; ++i) cig1[(size_t)i] = W.cig_tmp[i];
        ksw_extd2(W, qlen, q, tlen, t, o.mat, (int8_t)o.q, (int8_t)o.e, (int8_t)o.q2, (int8_t)o.e2, w, o.zdrop, end_bonus, flag, e2);
        bool same = e1.zdropped == e2.zdropped && e1.n_cigar == e2.n_cigar;
        if (mode == 2) same = same && e1.score == e2.score;
        else same = same && e1.max == e2.max && e1.max_t == e2.max_t && e1.max_q == e2.max_q && e1.reach_end == e2.reach_end &&
                    (!e1.reach_end || e1.mqe_t == e2.mqe_t);
        for (int i = 0; same && i < e1.n_cigar; ++i) same = cig1[(size_t)i] == W.cig_tmp[i];
        if (same) ++counts[1];
        else {
            if (verbose && counts[2] < 5) {
                fprintf(stderr, "shortcut mismatch mode=%d qlen=%d tlen=%d: max %u/%u max_t %d/%d max_q %d/%d reach %d/%d mqe_t %d/%d score %d/%d ncig %d/%d\n  q=", mode, qlen,
                        tlen, e1.max, e2.max, e1.max_t, e2.max_t, e1.max_q, e2.max_q, e1.reach_end, e2.reach_end, e1.mqe_t, e2.mqe_t, e1.score, e2.score,
                        e1.n_cigar, e2.n_cigar);
                for (int i = 0; i < qlen; ++i) fputc("ACGTN"[q[i]], stderr);
                fprintf(stderr, "\n  t=");
                for (int i = 0; i < tlen; ++i) fputc("ACGTN"[t[i]], stderr);
                fputc('\n', stderr);
            }
            ++counts[2];
        }
    }
    return 0;
}
#endif

// The reference sketch: slice == 0 -> the sequential sketch_segment_t (what build_ref_index runs); slice > 0 -> the
// concatenation of sketch_slice over consecutive slices of that many bases (what the device index build runs).
extern "C" int64_t hs_ref_sketch(const char* ref, int64_t len, int w, int k, int slice, uint64_t* out_x, uint64_t* out_y, int64_t cap) {
    std::vector<uint8_t> seq((size_t)len + 8, 0);
    for (int64_t i = 0; i < len; ++i) seq[(size_t)i] = nt4_of_char((unsigned char)ref[i]);
    int64_t n = 0;
    if (slice <= 0) {
        Work W;
        memset(&W, 0, sizeof(W));
        std::vector<A128> mv((size_t)len + 16), buf(256);
        W.mv = mv.data();
        W.sk_buf = buf.data();
        W.caps.max_mini = (int)mv.size();
        RingMem ring{W.sk_buf};
        sketch_segment_t(W, ring, (Ptr<const uint8_t>)seq.data(), (int)len, w, k, 0);
        for (int i = 0; i < W.n_mv && n < cap; ++i, ++n) { out_x[n] = mv[(size_t)i].x; out_y[n] = mv[(size_t)i].y; }
        return W.n_mv;
    }
    auto base_at = [&](int i) { return (int)seq[(size_t)i]; };
    auto emit = [&](uint64_t x, uint64_t y) { if (n < cap) { out_x[n] = x; out_y[n] = y; } ++n; };
    for (int64_t b = 0; b < len; b += slice)
        sketch_slice<12>((int)b, (int)std::min<int64_t>(len, b + slice), (int)len, w, k, base_at, emit);
    return n;
}
