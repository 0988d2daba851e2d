// Launch interface of the grouped DP service (align_kernel_dpg.hip), shared with api_align.hip.
#pragma once
#include <stdint.h>

#include "align/aln_types.hpp"

#define PMX_DPG_G 8            // lanes that share one request
#define PMX_DPG_MAXLEN 128     // query and target length of a request this service takes
#define PMX_DPG_CLASSES 4      // target columns per lane: 4, 8, 12, 16 (targets up to 32, 64, 96, 128 bases)
#define PMX_DPG_KINDS 3        // 0: gap fill (approximate maximum), 1: extension to the right, 2: extension to the left
#define PMX_DPG_BUCKETS (PMX_DPG_CLASSES * PMX_DPG_KINDS)
#define PMX_DPG_NO_BUCKET 15
#define PMX_DPG_WORK 13        // counts[]: the task counter of k_align_dp_group (entries 12, 13 belong to no bucket)
#define PMX_DPG_LDS_PER_REQ (PMX_DPG_MAXLEN + 4 * 2 * PMX_DPG_MAXLEN + 4 * PMX_DPG_MAXLEN + 4 * PMX_DPG_MAXLEN + 4 * 24)
#define PMX_DPG_LDS_BYTES (PMX_DPG_LDS_PER_REQ * (64 / PMX_DPG_G))
#define PMX_DPG_TB_PER_REQ (PMX_DPG_MAXLEN * PMX_DPG_MAXLEN)
#define PMX_DPG_TB_BYTES ((size_t)PMX_DPG_TB_PER_REQ * (64 / PMX_DPG_G))   // traceback slab of one wave

namespace pmx {
namespace aln {

struct DpgArgs {
    uint8_t* dp_req_base;       // DpReq entries, PMX_DP_REQ_PER_PASS per slot
    DpRes* dp_res_base;
    const uint32_t* worklist;   // slots of this round (NULL: 0 .. n_slots-1)
    int64_t n_slots;
    uint32_t* keys;             // k_dpg_collect: (bucket << 8 | 128 - qlen) per entry, PMX_DPG_NO_BUCKET for the ones left to the wave service
    uint32_t* ids;              //                 entry numbers (slot * PMX_DP_REQ_PER_PASS + entry)
    const uint32_t* sorted_ids; // the entries ordered by key
    uint32_t* counts;           // [16] requests per bucket; [14] the requests left to the wave service, [15] what k_align_dp_group refused
    uint8_t* tb;                // traceback slabs, PMX_DPG_TB_BYTES per block
    int q, e, q2, e2;           // gap costs with q + e <= q2 + e2 (swapped by the host if need be)
    int sc_mch, sc_mis, sc_N;
    int long_thres, long_diff;  // ksw2_extd2_sse.c:103-105
    unsigned long long* stats;  // [0] DP calls, [1] cells
    unsigned long long* prof;   // diagnostic (NULL = off): wave cycles of [0] set-up, [1] fill, [2] replay, [3] traceback + hand-over, [4] tasks, [5] fill steps
    uint32_t n_entries;         // entries behind dp_req_base (an id beyond it is counted in counts[15] and skipped)
    int shadow;                 // diagnostic: results go to dp_res_base (a copy), the requests stay posted for the wave service
};

__global__ void k_dpg_collect(DpgArgs D);
__global__ void k_align_dp_group(DpgArgs D);
__global__ void k_dpg_refuse_left(DpgArgs D);

}  // namespace aln
}  // namespace pmx
