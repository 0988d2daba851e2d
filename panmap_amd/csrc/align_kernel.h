// Launch interface of the ALIGN kernel (align_kernel.hip), shared with api_align.hip.
#pragma once
#include "align/aln_host.hpp"

#define PMX_ALIGN_WORK_BYTES 1280   // LDS bytes reserved for the Work descriptor

namespace pmx {
namespace aln {

struct AlignArgs {
    TppArena tpp;           // MUST stay first (IPtr::phys reads it from the kernarg segment); zero for the wave kernels
    // packed reads
    const uint64_t* words;
    const uint32_t* amb;
    const int64_t* woff;
    const int64_t* off;
    const uint8_t* recs;    // read records (64 bytes per read: words, ambiguity words, length; readset.hpp) or NULL: the compact tier reads them
    int64_t n_items;        // pairs (paired) or reads, or worklist length
    const uint32_t* worklist;   // tier 2: item ids to (re)process; NULL = 0..n_items-1
    uint32_t* retry_list;       // tier 1: items whose capacities overflowed (NULL in tier 2); tier 0: pairs for the wave tiers
    unsigned long long* retry_count;
    // tier 0 DP service (NULL elsewhere): see DpReq / DpRes in align/aln_types.hpp
    uint8_t* dp_req_base;       // slot_cap * sizeof(DpReq)
    DpRes* dp_res_base;         // slot_cap * PMX_DP_MAX_CALLS
    uint32_t* dp_ncached;       // served results per slot
    uint32_t* dp_slot_pairs;    // slot -> pair id
    uint32_t* dp_next_list;     // round >= 1: slots that posted another request
    unsigned long long* dp_count;   // round 0: slot allocator; round >= 1: length of dp_next_list
    uint32_t dp_slot_cap;
    int dp_round;
    int tpp_ring_w;             // k_align_reads_tpp: minimizer window length when the ring lives in LDS, else 0
    int sk_no_lane_ring;        // wave-per-pair kernels: 1 = sketch with the window ring in LDS instead of across the lanes
    int no_rows_dp;             // wave-per-read kernels: 1 = never the row-by-row DP (PMX_ALIGN_NO_ROWS_DP)
    // Minimizer hand-over: a pair that posts a DP request in the thread-per-pair kernel leaves its minimizer list here
    // (slot s: entry [s * mv_stride] holds the count in .x, the list follows), so the wave-per-pair tier does not have to
    // sketch the reads again when it takes the pair over (the sketch is ~40 % of a pair's time there).  Slots >= mv_slots
    // have no entry.
    // Thread-per-pair kernel, round 0: item of position `it` (pairs sorted by the first 16 bases of their first read,
    // so that the 64 pairs of a wave are neighbours on the genome: similar minimizers, anchor counts and loop trip
    // counts -> less lane divergence).  nullptr = identity.
    const uint32_t* pair_perm;
    A128* mv_handover;
    uint32_t mv_stride, mv_slots, mv_epoch;   // epoch: stamps the entries of this call (a slot may hold one of an earlier call)
    unsigned long long* work_queue;   // wave-per-item kernels: NULL = strided loop, else the counter the waves draw items from (zeroed by the host)
    int dp_class;               // k_align_dp_serve: 0 = every request, 1 = only small ones, 2 = only the others
    int dp_small_qlen, dp_small_tlen;   // register-DP class: qlen <=, tlen <=, traceback bytes <= dp_small_tb
    uint32_t dp_small_tb;
    const uint32_t* dp_left;    // k_align_dp_serve: NULL, or two counters of the requests the grouped service left (both zero: nothing to do)


    // Compact tier, two-kernel form (align_kernel_compact.hip): k_compact_seeds leaves the seeds of launch position `it` in
    // the hand-over words (align/aln_compact.hpp CSeedOutT) and their number in cseed_n[it]; NULL = the fused kernel
    uint32_t* cseeds;
    uint16_t* cseed_n;         // seeds | seeds of mate 1 << 8, or PMX_C_NSEED_BAIL
    // Second form of the compact tier (k_align_compact*_multi: several regions per mate, align/aln_compact_multi.hpp):
    // k_align_compact* appends the LAUNCH POSITIONS of the pairs that leave it after the seeds (their hand-over words stay
    // where they are) to multi_list, the second form runs over the list and appends what it cannot finish to retry_list.
    // NULL = every bail goes to retry_list at once.
    uint32_t* multi_list;
    unsigned long long* multi_count;   // [0] length of multi_list, [1] pairs the second form finished
    uint32_t* multi_ws;                // region records of the second form: PMX_CM_WS_WORDS * 64 words per wave of its grid
    // The pairs k_compact_seeds* gave up on (an ambiguous base, a sketch tie, ...) are known before the chain kernel starts:
    // k_compact_list_seed_bails lists them (early_list / early_count, retry_list order = launch order) and the general tiers
    // run them on a second stream BESIDE the chain kernel; seed_bails_listed != 0 tells k_align_compact* to pass them over.
    uint32_t* early_list;
    unsigned long long* early_count;
    int seed_bails_listed;

    unsigned long long* prof;   // 16 phase-cycle accumulators (diagnostic; NULL = off)
    int32_t* edits;             // NULL, or per read: count_read_errors (src/mm_align.c:122-133) of its first region, the read length without one
    unsigned long long* stats;  // [0] DP calls run, [1] DP cells (q * min(t, 2w+1)), [2] pairs the wave tiers ran a DP for
    int paired;
    int revcomp_mate2;
    // reference + options
    Opt opt;
    RefIndex ri;
    Layout layout;
    uint8_t* slow_base;
    size_t slow_stride;
    // outputs
    AlnRecord* records;
    uint32_t* cigars;
    uint64_t cigar_cap;
    unsigned long long* cigar_used;
};

__global__ void k_align_reads(AlignArgs A);
__global__ void k_align_reads_w4(AlignArgs A);
__global__ void k_align_reads_t1(AlignArgs A);
__global__ void k_align_reads_t1_w4(AlignArgs A);
__global__ void k_align_reads_tpp(AlignArgs A);
__global__ void k_align_dp_serve(AlignArgs A);
__global__ void k_align_compact16(AlignArgs A);   // retry_list / retry_count = its bail list; reference <= 32,767 bases; seeds from cseeds
__global__ void k_align_compact32(AlignArgs A);
__global__ void k_align_compact16_fused(AlignArgs A);   // sketch and probes inside (PMX_ALIGN_COMPACT_FUSED)
__global__ void k_align_compact32_fused(AlignArgs A);
__global__ void k_align_compact16_multi(AlignArgs A);   // over multi_list (grid-stride, the length is read on the device)
__global__ void k_align_compact32_multi(AlignArgs A);
__global__ void k_compact_list_seed_bails(AlignArgs A);   // cseed_n == PMX_C_NSEED_BAIL -> early_list / early_count
__global__ void k_compact_seeds16(AlignArgs A);   // sketch + index probes of every pair -> cseeds / cseed_n
__global__ void k_compact_seeds32(AlignArgs A);

// bytes of the traceback matrix ksw_extd2 needs for a request (same n_col as ksw2_extd2_sse.c:95-98)
PMX_HD size_t dp_request_tb_bytes(int qlen, int tlen, int w) {
    if (qlen <= 0 || tlen <= 0) return 0;
    if (w < 0) w = tlen > qlen ? tlen : qlen;
    int n_col = qlen < tlen ? qlen : tlen;
    n_col = (((n_col < w + 1 ? n_col : w + 1) + 15) / 16 + 1) * 16;
    return (size_t)(qlen + tlen - 1) * (size_t)n_col;
}

// a served DP goes to its place (the call index) in the slot's result list
PMX_HD void dp_store_result(const AlignArgs& A, int64_t slot, const DpReq* rq, const Ez& ez, const uint32_t* cigar, uint32_t status) {
    const uint32_t n = rq->call;
    if (n >= PMX_DP_MAX_CALLS) return;
    DpRes& R = A.dp_res_base[(size_t)slot * PMX_DP_MAX_CALLS + n];
    const bool bad = (status & PMX_ST_OVERFLOW) || ez.n_cigar > PMX_DP_MAX_CIGAR;
    R.ez = ez;
    R.key = bad ? 0xffffffffu : rq->key;
    if (!bad) for (int i = 0; i < ez.n_cigar; ++i) R.cigar[i] = cigar[i];
}

}  // namespace aln
}  // namespace pmx
