// CPU unit-test build of the ALIGN pipeline sources (panmap_amd/csrc/align/*.hpp) with PMX_W = 1.
// TEST INFRASTRUCTURE: lets the host logic of the kernels be checked against the compiled reference
// aligner in a container without a GPU.  Never linked into libpanmap_amd.so; the product path is the
// HIP kernel in panmap_amd/csrc/align_kernel.hip.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "align/aln_host.hpp"

using namespace pmx::aln;

extern "C" int hs_align(const char* ref, int64_t ref_len, int n_reads, const char** reads, const int* lens, int paired, AlnRecord* recs,
                        uint32_t* cigars, int64_t cig_cap, int64_t* cig_used, int verbose) {
    int64_t total = 0;
    int max_len = 0;
    for (int i = 0; i < n_reads; ++i) { total += lens[i]; max_len = std::max(max_len, lens[i]); }
    const int avg_len = n_reads > 0 ? (int)(total / n_reads) : 150;
    Opt o = make_opt(avg_len);
    HostRefIndex hri;
    build_ref_index(ref, ref_len, o, std::max(4096, max_len * (o.a + 1) * 2 + 64), hri);
    const RefIndex ri = hri.view();
    const int n_segs = paired ? 2 : 1;
    Layout L = plan_layout(max_len, n_segs, o, (size_t)1 << 30);
    std::vector<uint8_t> fast(L.fast_bytes + 64), slow(L.slow_bytes + 64);
    Work W;
    memset(&W, 0, sizeof(W));
    bind_work(W, L, fast.data(), slow.data());
    if (verbose) fprintf(stderr, "hostsim: k=%d w=%d mid_occ=%d fast=%zu slow=%zu max_tlen=%d\n", o.k, o.w, o.mid_occ, L.fast_bytes, L.slow_bytes, L.caps.max_tlen);
    const int n_items = paired ? n_reads / 2 : n_reads;
    int64_t used = 0;
    for (int it = 0; it < n_items; ++it) {
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
        map_frag(W, o, ri);
        const bool mapped = frag_is_mapped(W, paired);
        for (int s = 0; s < n_segs; ++s) {
            AlnRecord& rec = recs[paired ? 2 * it + s : it];
            memset(&rec, 0, sizeof(rec));
            rec.flags = (uint16_t)(W.status & 3);
            if (!mapped) continue;
            rec.mapped = 1;
            const Reg& r = W.regs[s][0];
            if (!r.has_p) continue;
            rec.flags |= PMX_REC_HAS_ALN;
            rec.rs = r.rs; rec.re = r.re; rec.qs = r.qs; rec.qe = r.qe;
            rec.mapq = r.mapq; rec.rev = r.rev; rec.proper_frag = r.proper_frag;
            rec.n_cigar = (uint16_t)r.n_cigar;
            rec.score = r.dp_max;
            rec.cigar_off = (uint32_t)used;
            if (used + r.n_cigar > cig_cap) return -2;
            const uint32_t* cg = reg_cigar(W, r);
            for (uint32_t i = 0; i < r.n_cigar; ++i) cigars[used + i] = cg[i];
            used += r.n_cigar;
        }
    }
    *cig_used = used;
    return 0;
}
