// C ABI, device part 2: ALIGN stage (see include/panmap_amd.h).
#include <hip/hip_runtime.h>

#include <mutex>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <tuple>

#include "align/aln_compact_defs.hpp"
#include "align_kernel.h"
#include "align_kernel_dpg.h"
#include "device/dev_util.hpp"
#include "readset.hpp"
#include "ref_index_device.h"

using namespace pmx;
using namespace pmx::aln;

static_assert(sizeof(Work) <= PMX_ALIGN_WORK_BYTES, "Work descriptor exceeds its LDS reservation");
static_assert(sizeof(AlnRecord) == sizeof(pmx_aln_record), "record layout mismatch");

struct pmx_aligner {
    Opt opt;
    HostRefIndex host;
    DevBuf<uint8_t> d_seq;
    DevBuf<uint64_t> d_pos;
    DevBuf<HtEnt> d_ht;
    DevBuf<float> d_logf_ratio, d_logf_int;
    DevBuf<uint64_t> d_pk, d_pk_amb;
    DevBuf<uint32_t> d_ht_pv;
    RefIndexDevice dev_index;   // the index when it was built on the device (set_reference)
    bool logf_uploaded = false;
    RefIndex ri;
    int mean_len = 150;
    // last result
    DevBuf<AlnRecord> records;
    DevBuf<uint32_t> cigars;
    DevBuf<unsigned long long> cigar_used;
    DevBuf<uint8_t> slow, slow2, slab0, slab_raw;
    DevBuf<A128> mv_handover;
    DevBuf<uint32_t> pp_idx, pp_idx2;   // pair order from the read order alone (PMX_ALIGN_PAIR_KEY1)
    DevBuf<char> pp_tmp;
    uint32_t mv_epoch = 0;
    DevBuf<uint32_t> retry_list2, bail_list;
    DevBuf<uint32_t> cseeds;            // compact tier, two-kernel form: seed hand-over (AlignArgs::cseeds / cseed_n)
    DevBuf<uint16_t> cseed_n;
    DevBuf<uint32_t> multi_list;        // compact tier, second form (several regions per mate): launch positions + counters
    DevBuf<unsigned long long> multi_count;
    DevBuf<uint32_t> multi_ws;
    DevBuf<uint32_t> early_list;        // pairs the compact tier's seeds kernel gave up on (run beside the chain kernels)
    DevBuf<uint8_t> dp_req;
    DevBuf<DpRes> dp_res;
    DevBuf<uint32_t> dp_ncached, dp_slot_pairs, dp_list_a, dp_list_b;
    DevBuf<uint32_t> dpg_keys, dpg_keys2, dpg_ids, dpg_ids2, dpg_counts;   // grouped DP service (align_kernel_dpg.hip)
    DevBuf<char> dpg_tmp;
    DevBuf<uint8_t> dpg_tb;
    DevBuf<DpRes> dpg_shadow;
    DevBuf<unsigned long long> dpg_prof;
    int64_t last_dp_requests = 0;
    int last_dp_rounds = 0;
    DevBuf<uint32_t> retry_list;
    DevBuf<unsigned long long> retry_count;
    int64_t last_retry = 0, last_tpp_retry = 0;
    DevBuf<unsigned long long> prof;
    DevBuf<unsigned long long> stats;   // AlignArgs::stats
    DevBuf<int32_t> edits;              // AlignArgs::edits while pmx_align_score_reads runs
    bool want_edits = false;
    pmx_align_stats last_stats;
    int64_t last_dp_slots = 0, last_compact = 0;
    int64_t n_records = 0;
    uint64_t cigar_cap = 0;
    size_t dev_total_mem = 0;            // hipMemGetInfo total, asked once
    double cigar_words_per_kbase = 0.0;   // CIGAR words per 1,000 read bases the last calls needed (sizes the next arena)
    unsigned long long last_cigar_used = 0;   // read back at the end of pmx_align_readset
    double last_occupancy = 0;
    hipEvent_t ev_results = nullptr, ev_fetched = nullptr;   // pmx_align_fetch_async: results ready / download finished
    bool fetch_pending = false;
};

// Grouped DP service (align_kernel_dpg.hip): scoring parameters for the kernel; false = the parameters leave the range in which
// plain 32-bit arithmetic stands for the reference's int8 lanes (no preset does): the wave service then takes everything
static bool dpg_setup(const Opt& o, DpgArgs& DG) {
    memset(&DG, 0, sizeof(DG));
    bool ok = true;
    int q = o.q, e = o.e, q2 = o.q2, e2 = o.e2;
    if (q2 + e2 < q + e) { std::swap(q, q2); std::swap(e, e2); }
    int min_sc = o.mat[1], max_abs = 0;
    for (int t = 0; t < 25; ++t) { if (t >= 1) min_sc = std::min<int>(min_sc, o.mat[t]); max_abs = std::max(max_abs, std::abs((int)o.mat[t])); }
    if (-min_sc > 2 * (q + e)) ok = false;   // (ksw2_extd2_sse.c:100: the reference returns without aligning)
    if (2 * (q2 + e2) + 2 * max_abs > 100 || q < 0 || e < 0 || q2 < 0 || e2 < 0) ok = false;
    DG.q = q; DG.e = e; DG.q2 = q2; DG.e2 = e2;
    DG.sc_mch = o.mat[0]; DG.sc_mis = o.mat[1]; DG.sc_N = o.mat[24] == 0 ? -e2 : o.mat[24];
    int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
    if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
    DG.long_thres = long_thres;
    DG.long_diff = long_thres * (e - e2) - (q2 - q) - e2;
    return ok;
}

// collect the requests of `n_slots` slots, order them by (columns per lane, kind, query length), serve them eight per wave
static void dpg_launch(pmx_ctx* ctx, pmx_aligner* al, DpgArgs& DG, int64_t n_slots, const uint32_t* worklist, int waves_per_cu, bool serve) {
    const int64_t n_ent = n_slots * PMX_DP_REQ_PER_PASS;
    al->dpg_keys.ensure((size_t)n_ent); al->dpg_keys2.ensure((size_t)n_ent); al->dpg_ids.ensure((size_t)n_ent); al->dpg_ids2.ensure((size_t)n_ent);
    al->dpg_counts.ensure(16);
    const int64_t grid = std::min<int64_t>((int64_t)ctx->n_cu * waves_per_cu, (n_ent + 7) / 8 + PMX_DPG_BUCKETS);
    al->dpg_tb.ensure((size_t)grid * PMX_DPG_TB_BYTES);   // a traceback window per launched wave (not per wave the chip could hold: --refine keeps an aligner per worker)
    DG.worklist = worklist; DG.n_slots = n_slots;
    DG.keys = al->dpg_keys.p; DG.ids = al->dpg_ids.p; DG.sorted_ids = al->dpg_ids2.p; DG.counts = al->dpg_counts.p;
    DG.tb = al->dpg_tb.p;
    if (!DG.dp_req_base || !DG.dp_res_base) throw std::runtime_error("grouped DP service: a buffer is missing");
    PMX_HIP(hipMemsetAsync(al->dpg_counts.p, 0, 16 * sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(k_dpg_collect, dim3((unsigned)std::min<int64_t>((n_ent + 255) / 256, (int64_t)ctx->n_cu * 8)), dim3(256), 0, ctx->stream, DG);
    size_t bytes = 0;
    PMX_HIP(rocprim::radix_sort_pairs(nullptr, bytes, al->dpg_keys.p, al->dpg_keys2.p, al->dpg_ids.p, al->dpg_ids2.p, (size_t)n_ent, 0, 12, ctx->stream));
    al->dpg_tmp.ensure(bytes);
    PMX_HIP(rocprim::radix_sort_pairs(al->dpg_tmp.p, bytes, al->dpg_keys.p, al->dpg_keys2.p, al->dpg_ids.p, al->dpg_ids2.p, (size_t)n_ent, 0, 12, ctx->stream));
    if (serve) hipLaunchKernelGGL(k_align_dp_group, dim3((unsigned)grid), dim3(64), PMX_DPG_LDS_BYTES, ctx->stream, DG);
    PMX_HIP(hipGetLastError());
}

// [0] += edit counts, [1] += records flagged invalid (pmx_align_score_reads)
// (off != NULL: a flagged record counts as an unmapped read -- its length -- the way the drop-in boundary reports it)
__global__ void k_sum_edits(const AlnRecord* __restrict__ recs, const int32_t* __restrict__ edits, int64_t n, unsigned long long* out,
                            const int64_t* __restrict__ off) {
    unsigned long long sum = 0, bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const bool flagged = (recs[i].flags & 3u) != 0;
        sum += flagged && off ? (unsigned long long)(off[i + 1] - off[i]) : (unsigned long long)(uint32_t)edits[i];
        bad += flagged ? 1ULL : 0ULL;
    }
    for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o); bad += __shfl_xor(bad, o); }
    if ((threadIdx.x & 63) == 0) {
        if (sum) atomicAdd(&out[0], sum);
        if (bad) atomicAdd(&out[1], bad);
    }
}

// pair order from the read order: first mates (even read indices) of the read set's locality order -> pair indices
struct IsEvenRead {
    __host__ __device__ bool operator()(const uint32_t& r) const { return (r & 1u) == 0u; }
};
// pair key = locality key of mate 1 (fragment start) in the high half, of mate 2 (fragment end) in the low half
__global__ void k_pair_keys(const uint32_t* __restrict__ read_key, int64_t n_pairs, uint64_t* key, uint32_t* idx) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pairs; i += (int64_t)gridDim.x * blockDim.x) {
        key[i] = (uint64_t)read_key[2 * i] << 32 | (uint64_t)read_key[2 * i + 1];
        idx[i] = (uint32_t)i;
    }
}
__global__ void k_halve(const uint32_t* __restrict__ in, int64_t n, uint32_t* out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = in[i] >> 1;
}

namespace {
int fail(int code, const std::string& msg) {
    set_error(msg);
    return code;
}
#define PMX_TRY try {
#define PMX_CATCH                                                      \
    }                                                                  \
    catch (const HipError& e) { return fail(PMX_ERR_DEVICE, e.msg); }  \
    catch (const std::exception& e) { return fail(PMX_ERR_DEVICE, e.what()); }

template <class T>
void upload(DevBuf<T>& d, const std::vector<T>& h, hipStream_t st) {
    d.ensure(h.size());
    if (!h.empty()) PMX_HIP(hipMemcpyAsync(d.p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice, st));
}
}  // namespace

extern "C" {

int pmx_aligner_set_reference(pmx_ctx* ctx, pmx_aligner* al, const char* reference, int64_t ref_len, int mean_read_len) {
    if (!ctx || !al || !reference || ref_len <= 0) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    al->mean_len = mean_read_len;
    al->opt = make_opt(mean_read_len);
    const int max_score = std::max(8192, (mean_read_len * 4 + 1024) * (al->opt.a + 1));
    RefIndex& r = al->ri;
    // Device build (ref_index_kernels.hip): stream-ordered behind whatever still reads the old index, one short host round
    // trip.  PMX_ALIGN_HOST_INDEX=1, an even k, a reference of 2 Mb or more, or a repeat-rich reference whose mid_occ the
    // counters cannot decide: the host build.
    bool on_device = false;
    if (!pmx::opt_str(pmx::O_ALIGN_HOST_INDEX) && ref_index_device_supported(al->opt, ref_len))
        on_device = build_ref_index_device(ctx->stream, reference, ref_len, al->opt, al->dev_index);
    if (on_device) {
        const bool new_tables = finish_ref_opt(al->opt, max_score, al->host) || !al->logf_uploaded;
        if (new_tables) {
            upload(al->d_logf_ratio, al->host.logf_ratio, ctx->stream);
            upload(al->d_logf_int, al->host.logf_int, ctx->stream);
            PMX_HIP(hipStreamSynchronize(ctx->stream));   // (pageable source)
            al->logf_uploaded = true;
        }
        const RefIndexDevice& d = al->dev_index;
        r.seq = d.seq.p;
        r.ht_mask = d.ht_mask;
        r.ht = d.ht.p;
        r.pos = d.pos.p;
        r.pk = d.pk.p;
        r.pk_amb = d.pk_amb.p;
        r.ht_pv = d.ht_pv.p;
    } else {
        PMX_HIP(hipStreamSynchronize(ctx->stream));   // nothing may still read the old index
        al->opt = make_opt(mean_read_len);
        build_ref_index(reference, ref_len, al->opt, max_score, al->host);
        upload(al->d_seq, al->host.seq, ctx->stream);
        upload(al->d_ht, al->host.ht, ctx->stream);
        upload(al->d_pos, al->host.pos, ctx->stream);
        upload(al->d_logf_ratio, al->host.logf_ratio, ctx->stream);
        upload(al->d_logf_int, al->host.logf_int, ctx->stream);
        upload(al->d_pk, al->host.pk, ctx->stream);
        upload(al->d_pk_amb, al->host.pk_amb, ctx->stream);
        upload(al->d_ht_pv, al->host.ht_pv, ctx->stream);
        al->logf_uploaded = true;
        r.seq = al->d_seq.p;
        r.ht_mask = (uint32_t)al->host.ht.size() - 1;
        r.ht = al->d_ht.p;
        r.pos = al->d_pos.p;
        r.pk = al->d_pk.p;
        r.pk_amb = al->d_pk_amb.p;
        r.ht_pv = al->d_ht_pv.p;
        PMX_HIP(hipStreamSynchronize(ctx->stream));
    }
    r.len = (int32_t)ref_len;
    r.logf_ratio = al->d_logf_ratio.p;
    r.logf_int = al->d_logf_int.p;
    r.n_logf = (int32_t)al->host.logf_int.size();
    return PMX_OK;
    PMX_CATCH
}

int pmx_aligner_index_digest(pmx_ctx* ctx, pmx_aligner* al, uint64_t out[5]) {
    if (!ctx || !al || !out) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    const RefIndex& r = al->ri;
    const size_t cap = (size_t)r.ht_mask + 1;
    std::vector<HtEnt> ht(cap);
    std::vector<uint32_t> pv(cap);
    PMX_HIP(hipMemcpyAsync(ht.data(), r.ht, cap * sizeof(HtEnt), hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipMemcpyAsync(pv.data(), r.ht_pv, cap * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    uint64_t n_pos = 0, n_keys = 0;
    for (const HtEnt& e : ht)
        if (e.key != UINT64_MAX) { ++n_keys; n_pos = std::max<uint64_t>(n_pos, (uint64_t)e.off + e.cnt); }
    std::vector<uint64_t> pos((size_t)n_pos);
    if (n_pos) PMX_HIP(hipMemcpy(pos.data(), r.pos, n_pos * sizeof(uint64_t), hipMemcpyDeviceToHost));
    uint64_t digest = 0, covered = 0;
    for (size_t s = 0; s < cap; ++s) {
        const HtEnt& e = ht[s];
        if (e.key == UINT64_MAX) continue;
        if (((uint32_t)mix64(e.key) & r.ht_mask) != s) {   // every slot between the home slot and this one must be taken
            for (uint32_t q = (uint32_t)mix64(e.key) & r.ht_mask; q != s; q = (q + 1) & r.ht_mask)
                if (ht[q].key == UINT64_MAX) return fail(PMX_ERR_DEVICE, "reference index: a key is not reachable by linear probing");
        }
        uint64_t h = mix64(e.key) ^ mix64((uint64_t)e.cnt + 0x9e3779b97f4a7c15ULL);
        for (uint32_t q = 0; q < e.cnt; ++q) h = mix64(h ^ (pos[(size_t)e.off + q] + q));
        if (e.cnt == 1 && (pos[e.off] >> 32) == 0 && pv[s] != (uint32_t)pos[e.off]) return fail(PMX_ERR_DEVICE, "reference index: ht_pv does not match pos");
        if (e.cnt != 1 && pv[s] != 0xffffffffu) return fail(PMX_ERR_DEVICE, "reference index: ht_pv set for a repeated minimizer");
        digest += h;
        covered += e.cnt;
    }
    out[0] = covered;
    out[1] = n_keys;
    out[2] = (uint64_t)(int64_t)al->opt.mid_occ;
    out[3] = digest;
    out[4] = r.ht == al->dev_index.ht.p ? 1 : 0;
    return PMX_OK;
    PMX_CATCH
}

int pmx_aligner_create(pmx_ctx* ctx, const char* reference, int64_t ref_len, int mean_read_len, pmx_aligner** out) {
    if (!ctx || !reference || ref_len <= 0 || !out) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<pmx_aligner> al(new pmx_aligner());
    al->cigar_used.alloc(1);
    const int rc = pmx_aligner_set_reference(ctx, al.get(), reference, ref_len, mean_read_len);
    if (rc != PMX_OK) return rc;
    *out = al.release();
    return PMX_OK;
    PMX_CATCH
}

void pmx_aligner_free(pmx_ctx* ctx, pmx_aligner* al) {
    if (ctx) (void)hipSetDevice(ctx->device);
    if (al && al->ev_results) (void)hipEventDestroy(al->ev_results);
    if (al && al->ev_fetched) (void)hipEventDestroy(al->ev_fetched);
    delete al;
}

// Pair order of a paired read set: pairs sorted by (locality key of mate 1, of mate 2) -- the 64 pairs of a wave then start
// AND end within a few bases of each other.  side == nullptr: on the context's stream (or, when it was enqueued earlier on a
// side stream, the context's stream waits for it); side != nullptr: enqueued there, behind everything the context's stream
// holds now (the read order).  -> the permutation (device), or nullptr when the read set has no locality order.
}  // extern "C"
namespace pmx {
const uint32_t* readset_pair_order(pmx_ctx* ctx, const pmx_readset* rs, hipStream_t side) {
    const int64_t n_items = rs->n / 2;
    if (rs->has_pair_order) {
        if (rs->pair_ev_pending && !side) { PMX_HIP(hipStreamWaitEvent(ctx->stream, rs->pair_ev, 0)); rs->pair_ev_pending = false; }
        return rs->pp_idx2.p;
    }
    if (!readset_locality_order(ctx, rs) || n_items < 1) return nullptr;
    rs->pp_key.ensure((size_t)n_items); rs->pp_key2.ensure((size_t)n_items); rs->pp_idx.ensure((size_t)n_items); rs->pp_idx2.ensure((size_t)n_items + 1);
    size_t bytes = 0;
    PMX_HIP(rocprim::radix_sort_pairs(nullptr, bytes, rs->pp_key.p, rs->pp_key2.p, rs->pp_idx.p, rs->pp_idx2.p, (size_t)n_items, 0, 64, ctx->stream));
    rs->pp_tmp.ensure(bytes);
    hipStream_t st = ctx->stream;
    if (side) {
        if (!rs->pair_ev) PMX_HIP(hipEventCreateWithFlags(&rs->pair_ev, hipEventDisableTiming));
        PMX_HIP(hipEventRecord(rs->pair_ev, ctx->stream));      // the read keys are in place behind this point
        PMX_HIP(hipStreamWaitEvent(side, rs->pair_ev, 0));
        st = side;
    }
    hipLaunchKernelGGL(k_pair_keys, dim3((unsigned)std::min<int64_t>((n_items + 255) / 256, (int64_t)ctx->n_cu * 8)), dim3(256), 0, st, rs->loc_key.p, n_items,
                       rs->pp_key.p, rs->pp_idx.p);
    PMX_HIP(rocprim::radix_sort_pairs(rs->pp_tmp.p, bytes, rs->pp_key.p, rs->pp_key2.p, rs->pp_idx.p, rs->pp_idx2.p, (size_t)n_items, 0, 64, st));
    if (side) { PMX_HIP(hipEventRecord(rs->pair_ev, side)); rs->pair_ev_pending = true; }
    rs->has_pair_order = true;
    return rs->pp_idx2.p;
}
}  // namespace pmx
extern "C" {

// Enqueue the align stage's pair order of a packed, paired read set NOW, on a side stream of the context: it depends on the
// reads alone, and made here it runs beside the place stage (scoring is latency-bound) instead of between the placement
// and the first align kernel (10M reads: ~1 ms).  Optional: an aligner that finds none makes it itself.
int pmx_readset_order_pairs(pmx_ctx* ctx, pmx_readset* rs) {
    if (!ctx || !rs) return PMX_ERR_ARG;
    if (!rs->packed) return fail(PMX_ERR_ARG, "read set is not packed");
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    if (rs->n < 2 || rs->n / 2 >= (int64_t)UINT32_MAX) return PMX_OK;
    if (!ctx->pair_stream) ctx->pair_stream = create_dedicated_stream(ctx->n_cu);
    (void)readset_pair_order(ctx, rs, ctx->pair_stream);
    return PMX_OK;
    PMX_CATCH
}

static int align_readset_once(pmx_ctx* ctx, pmx_aligner* al, const pmx_readset* rs, int paired, int revcomp_mate2, uint64_t cigar_cap) {
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    const int64_t n_items = paired ? rs->n / 2 : rs->n;   // an odd trailing read is ignored (src/mm_align.c:372)
    if (al->fetch_pending) {   // a download of the previous results on another stream (pmx_align_fetch_async) reads the buffers this call overwrites
        PMX_HIP(hipStreamWaitEvent(ctx->stream, al->ev_fetched, 0));
        al->fetch_pending = false;
    }
    al->n_records = rs->n;
    al->records.ensure((size_t)std::max<int64_t>(rs->n, 1));
    PMX_HIP(hipMemsetAsync(al->records.p, 0, sizeof(AlnRecord) * (size_t)std::max<int64_t>(rs->n, 1), ctx->stream));
    al->cigar_cap = cigar_cap;
    al->cigars.ensure(al->cigar_cap);
    PMX_HIP(hipMemsetAsync(al->cigar_used.p, 0, sizeof(unsigned long long), ctx->stream));
    // (the counters are read back after every call, an empty read set's too: a rank whose shard holds no read)
    al->stats.ensure(4);
    PMX_HIP(hipMemsetAsync(al->stats.p, 0, 4 * sizeof(unsigned long long), ctx->stream));
    if (n_items <= 0) {
        al->last_dp_slots = 0; al->last_compact = 0; al->last_tpp_retry = 0; al->last_retry = 0; al->last_dp_rounds = 0; al->last_dp_requests = 0;
        memset(&al->last_stats, 0, sizeof(al->last_stats));
        return PMX_OK;
    }

    // Tier 1: compact all-LDS layout (typical short-read pairs); tier 2: general capacities for the pairs
    // that overflowed tier 1 (and for everything when the compact layout does not fit LDS).
    int waves_per_simd = 4;
    size_t lds_budget = 24 * 1024;
    if (const char* e = pmx::opt_str(pmx::O_ALIGN_LDS_KB)) lds_budget = (size_t)atoi(e) * 1024;
    if (const char* e = pmx::opt_str(pmx::O_ALIGN_WAVES)) waves_per_simd = atoi(e);
    const bool use_tier1 = !pmx::opt_str(pmx::O_ALIGN_NO_TIER1);
    auto kern = waves_per_simd >= 4 ? k_align_reads_w4 : k_align_reads;
    const int n_segs = paired ? 2 : 1;

    AlignArgs A;
    A.tpp.base = nullptr; A.tpp.wave_stride = 0; A.tpp.pad = 0;
    A.work_queue = nullptr;
    A.cseeds = nullptr; A.cseed_n = nullptr;
    A.multi_list = nullptr; A.multi_count = nullptr; A.multi_ws = nullptr;
    A.early_list = nullptr; A.early_count = nullptr; A.seed_bails_listed = 0;
    A.words = rs->words.p; A.amb = rs->amb.p; A.woff = rs->woff.p; A.off = rs->off.p;
    A.recs = rs->has_recs && rs->packed ? rs->recs.p : nullptr;
    A.paired = paired ? 1 : 0;
    A.revcomp_mate2 = revcomp_mate2 ? 1 : 0;
    A.opt = al->opt;
    A.ri = al->ri;
    A.records = al->records.p;
    A.cigars = al->cigars.p;
    A.cigar_cap = al->cigar_cap;
    A.cigar_used = al->cigar_used.p;
    al->stats.ensure(4);
    PMX_HIP(hipMemsetAsync(al->stats.p, 0, 4 * sizeof(unsigned long long), ctx->stream));
    A.stats = al->stats.p;
    A.edits = nullptr;
    if (al->want_edits) { al->edits.ensure((size_t)std::max<int64_t>(rs->n, 1)); A.edits = al->edits.p; }
    al->last_dp_slots = 0;
    al->last_compact = 0;
    al->last_tpp_retry = 0; al->last_retry = 0; al->last_dp_rounds = 0; al->last_dp_requests = 0;
    memset(&al->last_stats, 0, sizeof(al->last_stats));
    al->last_stats.n_items = n_items;
    A.prof = nullptr;
    if (pmx::opt_str(pmx::O_ALIGN_PROF)) {
        al->prof.ensure(32);
        PMX_HIP(hipMemsetAsync(al->prof.p, 0, 32 * sizeof(unsigned long long), ctx->stream));
        A.prof = al->prof.p;
    }

    auto kern_t1 = waves_per_simd >= 4 ? k_align_reads_t1_w4 : k_align_reads_t1;
    auto launch = [&](decltype(kern) kfn, const Layout& L, int64_t n_work, const uint32_t* worklist, uint32_t* retry_list, DevBuf<uint8_t>& slab,
                      int64_t max_grid = 0) {
        const size_t lds_bytes = PMX_ALIGN_WORK_BYTES + L.fast_bytes + 16;
        if (pmx::opt_str(pmx::O_ALIGN_VERBOSE)) fprintf(stderr, "[pmx align] wave-tier launch: %lld items, %zu LDS bytes per wave, %zu HBM slab bytes per wave\n", (long long)n_work, lds_bytes, (size_t)L.slow_bytes);
        if (lds_bytes > 160 * 1024) throw std::runtime_error("reads too long for the LDS work arena");
        if (lds_bytes > 64 * 1024) PMX_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        int waves_per_cu = (int)std::min<size_t>((size_t)(waves_per_simd >= 4 ? 16 : 8), (size_t)(160 * 1024) / lds_bytes);
        if (waves_per_cu < 1) waves_per_cu = 1;
        int64_t grid = (int64_t)ctx->n_cu * waves_per_cu;
        // a few thousand pairs: one workgroup each, so that the hardware hands a free slot the next pair (with a resident
        // grid and a strided loop a wave that drew two slow pairs decides the launch)
        if (n_work <= 16384 && !pmx::opt_str(pmx::O_ALIGN_RESIDENT_GRID)) grid = n_work;
        if (grid > n_work) grid = n_work;
        if (max_grid > 0 && grid > max_grid) grid = max_grid;
        A.layout = L;
        A.slow_stride = (L.slow_bytes + 255) & ~(size_t)255;
        {
            // every workgroup owns a slab (long reads: ~20 MB each, 8 MB of it traceback): the grid is what a third of the
            // device memory -- at most 96 GB -- pays for (10 kb reads: 4,096 waves = 82 GB, the resident set of the chip);
            // the kernel strides over the items with whatever grid it gets
            if (al->dev_total_mem == 0) {
                size_t free_b = 0, total_b = 0;
                PMX_HIP(hipMemGetInfo(&free_b, &total_b));
                al->dev_total_mem = total_b;
            }
            const size_t total_b = al->dev_total_mem;
            size_t budget = std::max<size_t>(std::min<size_t>((size_t)96 << 30, total_b / 3), slab.n * sizeof(uint8_t));
            if (const char* e = pmx::opt_str(pmx::O_ALIGN_SLAB_MB)) budget = (size_t)std::max<long long>(atoll(e), 1) << 20;   // tests: force a small grid
            const int64_t fit = (int64_t)(budget / std::max<size_t>(A.slow_stride, 1));
            if (grid > fit) grid = std::max<int64_t>(fit, 1);
        }
        slab.ensure(A.slow_stride * (size_t)grid);
        A.slow_base = slab.p;
        A.n_items = n_work;
        A.worklist = worklist;
        A.retry_list = retry_list;
        A.retry_count = al->retry_count.p;
        hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(64), lds_bytes, ctx->stream, A);
        PMX_HIP(hipGetLastError());
    };

    al->retry_count.ensure(4);   // [0] pairs for the next (wave) tier, [1] DP requests of the current tier-0 round, [2] compact-tier bails
    PMX_HIP(hipMemsetAsync(al->retry_count.p, 0, 4 * sizeof(unsigned long long), ctx->stream));
    // test hook: cap the CIGAR operations per region in EVERY tier, so that gapped alignments overflow and the
    // boundary's handling of invalid records can be exercised (tests/test_align_gpu.py)
    int test_max_cigar = 0;
    if (const char* e = pmx::opt_str(pmx::O_ALIGN_TEST_MAX_CIGAR)) test_max_cigar = atoi(e);
    auto hooked = [&](Layout L) { if (test_max_cigar > 0 && L.caps.max_cigar > test_max_cigar) L.caps.max_cigar = test_max_cigar; return L; };
    const Layout general = hooked(plan_layout((int)rs->max_len, n_segs, al->opt, lds_budget));
    const Layout compact = hooked(plan_layout_compact((int)rs->max_len, n_segs, al->opt));
    const bool tier1_fits = use_tier1 && al->opt.is_sr_like && PMX_ALIGN_WORK_BYTES + compact.fast_bytes + 16 <= 40 * 1024;
    const bool use_tier0 = tier1_fits && !pmx::opt_str(pmx::O_ALIGN_NO_TPP);
    const bool use_dp_service = !pmx::opt_str(pmx::O_ALIGN_NO_DP_SERVICE);
    // reads both counters; [1] is reset for the next round, [0] only when asked
    auto read_counts = [&](int64_t& n_next_tier, int64_t& n_dp, bool reset_next_tier) {
        unsigned long long h[2] = {0, 0};
        PMX_HIP(hipMemcpyAsync(h, al->retry_count.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        PMX_HIP(hipStreamSynchronize(ctx->stream));
        PMX_HIP(hipMemsetAsync(al->retry_count.p + (reset_next_tier ? 0 : 1), 0, sizeof(unsigned long long) * (reset_next_tier ? 2 : 1), ctx->stream));
        n_next_tier = (int64_t)h[0];
        n_dp = (int64_t)h[1];
    };
    A.dp_req_base = nullptr; A.dp_res_base = nullptr; A.dp_ncached = nullptr; A.dp_slot_pairs = nullptr;
    A.dp_next_list = nullptr; A.dp_count = nullptr; A.dp_slot_cap = 0; A.dp_round = 0;
    A.dp_left = nullptr; A.dp_class = 0; A.dp_small_qlen = 0; A.dp_small_tlen = 0; A.dp_small_tb = 0; A.tpp_ring_w = 0;
    A.sk_no_lane_ring = pmx::opt_str(pmx::O_ALIGN_NO_LANE_RING) ? 1 : 0;
    A.no_rows_dp = pmx::opt_str(pmx::O_ALIGN_NO_ROWS_DP) ? 1 : 0;
    A.mv_handover = nullptr; A.mv_stride = 0; A.mv_slots = 0; A.mv_epoch = ++al->mv_epoch;
    A.pair_perm = nullptr;
    timer_begin(ctx, "align");
    if (tier1_fits) {
        al->retry_list.ensure((size_t)n_items);
        al->retry_list2.ensure((size_t)n_items);
        al->last_tpp_retry = 0;
        al->last_dp_requests = 0;
        al->last_dp_rounds = 0;
        al->last_dp_slots = 0;
        al->last_retry = 0;
        // the wave-per-pair tiers over a list of pairs (nullptr: every item)
        auto run_wave_tiers = [&](int64_t n_t1, const uint32_t* t1_list) {
            int64_t n_retry = 0, unused = 0;
            if (n_t1 > 0) {
                launch(kern_t1, compact, n_t1, t1_list, al->retry_list.p, al->slow);
                read_counts(n_retry, unused, true);
            }
            al->last_retry += n_retry;
            if (n_retry > 0) {
                // general capacities; what overflows even those (a mate whose every minimizer hits a long repeat: hundreds of
                // anchors per minimizer) runs once more with 16x the anchors (the chain cells index anchors with 16 bits), a few waves with their arrays in HBM
                launch(kern, general, n_retry, al->retry_list.p, al->retry_list2.p, al->slow2);
                int64_t n_huge = 0;
                read_counts(n_huge, unused, true);
                if (n_huge > 0) {
                    const Layout huge = hooked(plan_layout((int)rs->max_len, n_segs, al->opt, lds_budget, 0, 16));
                    launch(kern, huge, n_huge, al->retry_list2.p, nullptr, al->slow2, 64);
                }
            }
        };
        if (!use_tier0) run_wave_tiers(n_items, nullptr);
        if (use_tier0) {   // tier 0: thread per pair + DP service rounds
            int tpp_waves = 16;   // 4 per SIMD: what k_align_reads_tpp's register allocation targets (PMX_TPP_OCC)
            if (const char* e = pmx::opt_str(pmx::O_ALIGN_TPP_WAVES)) tpp_waves = atoi(e);
            const int64_t max_grid = std::min<int64_t>((int64_t)ctx->n_cu * tpp_waves, (n_items + 63) / 64);
            // thread-per-pair layout: interleaved arena per wave + a small contiguous struct region per thread
            size_t tpp_tb = 0;   // in-lane DPs measured slower than request + replay (divergence): off
            if (const char* e = pmx::opt_str(pmx::O_ALIGN_TPP_TB)) tpp_tb = (size_t)atoll(e);
            const Layout tpp_layout = hooked(plan_layout_tpp((int)rs->max_len, n_segs, al->opt, tpp_tb));
            const size_t tpp_wave_stride = tpp_arena_bytes(tpp_layout) * 64;
            const size_t tpp_raw_stride = (tpp_layout.raw_bytes + 255) & ~(size_t)255;
            al->slab0.ensure(tpp_wave_stride * (size_t)max_grid);
            al->slab_raw.ensure(tpp_raw_stride * (size_t)max_grid);   // per wave
            if (tpp_wave_stride > UINT32_MAX) throw std::runtime_error("thread-per-pair arena stride exceeds 32 bits");
            A.tpp.base = al->slab0.p;
            A.tpp.wave_stride = (uint32_t)tpp_wave_stride;
            const Layout dp_layout = plan_layout_dp((int)rs->max_len, n_segs, al->opt);
            const size_t dp_lds = PMX_ALIGN_WORK_BYTES + dp_layout.fast_bytes + 16;
            const size_t dp_stride = (dp_layout.slow_bytes + 255) & ~(size_t)255;
            // A replay round costs a fixed ~4-5 ms (one pair's pass through the thread-per-pair kernel) plus the DPs, the
            // wave tier ~0.2 us per easy pair and ~0.9 us per hard one.  First round (pairs asking for their first DP: mostly
            // easy ones): the wave tier below 16,384 pairs.  Later rounds hold the pairs that needed a DP before, i.e.
            // hard ones: another service round pays down to a quarter of that (real 150 bp reads: 53.8 -> 50.3 ms).
            // (round 3, 10M reads: 13.9k first-round requests through the service + one replay: 34.4 ms for the stage, through
            // the wave tier 35.9)
            int64_t small_rounds = 8192;
            if (const char* e = pmx::opt_str(pmx::O_ALIGN_TPP_MIN)) small_rounds = atoll(e);
            const int64_t dp_max_grid = (int64_t)ctx->n_cu * (int64_t)std::min<size_t>(16, (size_t)(160 * 1024) / dp_lds);
            const int small_qlen = 192, small_tlen = 192;   // ksw_extd2_reg<3>: up to three target columns per lane
            const Layout dps_layout = plan_layout_dp((int)rs->max_len, n_segs, al->opt, small_qlen, small_tlen);
            const size_t dps_lds = PMX_ALIGN_WORK_BYTES + dps_layout.fast_bytes + 16;
            const int64_t dps_max_grid = (int64_t)ctx->n_cu * (int64_t)std::min<size_t>(16, (size_t)(160 * 1024) / dps_lds);
            const size_t dps_stride = (dps_layout.slow_bytes + 255) & ~(size_t)255;
            const bool dp_two_class = !pmx::opt_str(pmx::O_ALIGN_DP_ONE_CLASS);
            DpgArgs DG;
            bool dpg_ok = dpg_setup(al->opt, DG) && !pmx::opt_str(pmx::O_ALIGN_NO_DP_GROUP);
            int dpg_waves = 8;
            if (const char* e = pmx::opt_str(pmx::O_ALIGN_DPG_WAVES)) dpg_waves = std::max(1, atoi(e));
            if (use_dp_service) {
                al->dp_req.ensure((size_t)n_items * PMX_DP_REQ_PER_PASS * sizeof(DpReq));
                al->dp_res.ensure((size_t)n_items * PMX_DP_MAX_CALLS);
                al->dp_ncached.ensure((size_t)n_items);
                al->dp_slot_pairs.ensure((size_t)n_items);
                al->dp_list_a.ensure((size_t)n_items);
                al->dp_list_b.ensure((size_t)n_items);
                al->slow.ensure(dp_stride * (size_t)dp_max_grid);
                al->slow2.ensure(dps_stride * (size_t)dps_max_grid);
                if (!pmx::opt_str(pmx::O_ALIGN_NO_MV_HANDOVER))
                    al->mv_handover.ensure((size_t)std::min<int64_t>(std::min<int64_t>(n_items, UINT32_MAX - 1), 131072) * ((size_t)tpp_layout.caps.max_mini + 1u));
            }
            // minimizer window ring in LDS when 16 waves per CU still fit (12 B x w x 64 lanes per wave)
            size_t tpp_lds_bytes = (size_t)al->opt.w * 64 * 12;
            if (tpp_lds_bytes * (size_t)tpp_waves > (size_t)150 * 1024 || pmx::opt_str(pmx::O_ALIGN_NO_LDS_RING)) tpp_lds_bytes = 0;
            // what the thread-per-pair passes and the DP service read from the launch arguments (a run of the tail clears them
            // at its end; the tail may run twice per call: see run_tail)
            auto tail_setup = [&]() {
                if (use_dp_service) {
                    PMX_HIP(hipMemsetAsync(al->dp_ncached.p, 0, sizeof(uint32_t) * (size_t)n_items, ctx->stream));
                    A.dp_req_base = al->dp_req.p; A.dp_res_base = al->dp_res.p; A.dp_ncached = al->dp_ncached.p;
                    A.dp_slot_pairs = al->dp_slot_pairs.p;
                    A.dp_slot_cap = (uint32_t)std::min<int64_t>(n_items, UINT32_MAX - 1);
                    if (!pmx::opt_str(pmx::O_ALIGN_NO_MV_HANDOVER)) {
                        A.mv_stride = (uint32_t)tpp_layout.caps.max_mini + 1u;
                        A.mv_slots = (uint32_t)std::min<int64_t>(A.dp_slot_cap, 131072);
                        A.mv_handover = al->mv_handover.p;
                        A.mv_epoch = ++al->mv_epoch;   // (a slot may hold an entry of an earlier run)
                    }
                }
                A.tpp_ring_w = tpp_lds_bytes ? al->opt.w : 0;
                A.dp_count = al->retry_count.p + 1;
                A.retry_list = al->retry_list2.p;
                A.retry_count = al->retry_count.p;
                A.layout = tpp_layout;
                A.worklist = nullptr;
                A.dp_round = 0;
            };
            auto launch_tpp = [&](int round, int64_t n_work, const uint32_t* worklist, uint32_t* next_list) {
                int64_t grid = std::min<int64_t>(max_grid, (n_work + 63) / 64);
                A.slow_stride = tpp_raw_stride;
                A.slow_base = al->slab_raw.p;
                A.n_items = n_work;
                A.worklist = worklist;
                A.dp_round = round;
                A.dp_next_list = next_list;
                hipLaunchKernelGGL(k_align_reads_tpp, dim3((unsigned)grid), dim3(64), tpp_lds_bytes, ctx->stream, A);
                PMX_HIP(hipGetLastError());
            };
            const uint32_t* order = nullptr;   // launch order of the first pass: pairs sorted by a locality key
            if (!pmx::opt_str(pmx::O_ALIGN_NO_PAIR_SORT)) {
                // the read set's locality order (shared with the seeding stage); pairs: the even reads of it, in that order
                const uint32_t* read_order = readset_locality_order(ctx, rs);
                if (read_order && !paired) order = read_order;
                else if (read_order && !pmx::opt_str(pmx::O_ALIGN_PAIR_KEY1)) {
                    // pairs by (key of mate 1, key of mate 2): the 64 pairs of a wave then start AND end within a few bases
                    // of each other -- same anchors, same overlap of the mates, same trip counts in every per-lane loop
                    // (round 3, 10M reads: k_align_compact16 28.2 -> 25.0 ms against the order by mate 1 alone, which
                    // PMX_ALIGN_PAIR_KEY1 still selects; the extra 64-bit sort of the pairs is ~1 ms of that)
                    // (made ahead of time by pmx_readset_order_pairs when the host asked for it: then only an event to wait for)
                    order = readset_pair_order(ctx, rs, nullptr);
                } else if (read_order) {
                    al->pp_idx.ensure((size_t)rs->n); al->pp_idx2.ensure((size_t)n_items + 1);
                    size_t bytes = 0;
                    PMX_HIP(rocprim::select(nullptr, bytes, read_order, al->pp_idx.p, al->pp_idx2.p + n_items, (size_t)rs->n, IsEvenRead(), ctx->stream));
                    al->pp_tmp.ensure(bytes);
                    PMX_HIP(rocprim::select(al->pp_tmp.p, bytes, read_order, al->pp_idx.p, al->pp_idx2.p + n_items, (size_t)rs->n, IsEvenRead(), ctx->stream));
                    hipLaunchKernelGGL(k_halve, dim3((unsigned)std::min<int64_t>((n_items + 255) / 256, (int64_t)ctx->n_cu * 8)), dim3(256), 0, ctx->stream,
                                       al->pp_idx.p, n_items, al->pp_idx2.p);
                    order = al->pp_idx2.p;
                }
            }
            // Compact tier (align_kernel_compact.hip): every pair first, work state in LDS; what it cannot finish comes
            // back as the bail list, which is the launch order of the general thread-per-pair kernel below.
            int64_t n_t0 = n_items;
            const bool use_compact = paired && al->opt.is_sr_like && al->opt.w == PMX_C_W && (al->opt.k & 1) && rs->max_len <= PMX_C_MAXLEN && n_items < (int64_t)UINT32_MAX && !pmx::opt_str(pmx::O_ALIGN_NO_COMPACT);
            if (!use_compact) timer_begin(ctx, "align_dom");   // the dominant kernel on its own (bench.py roofline)
            bool early_running = false;
            // THE TAIL: the general tiers over a list of pairs in launch order (`order`, n_t0 of them; nullptr = every item):
            // thread-per-pair pass, DP service rounds with replays, then the wave-per-pair tiers for what is left.  With the
            // compact tier it runs twice per call: once BESIDE the compact chain kernel, on the context's second stream, for
            // the pairs the seeds kernel gave up on, and once after it for the pairs the chain kernels hand back.
            auto run_tail = [&](const uint32_t* order, int64_t n_t0) {
            tail_setup();
            int64_t n_t1 = n_t0;
            const uint32_t* t1_list = order;
            // Few bails: a thread-per-pair launch that small cannot fill the chip and lasts as long as a full one (a wave takes
            // ~2 ms whatever the grid) before the wave-per-pair tier gets the pairs that need a DP; below 4096 bails the wave
            // tier takes all of them at once (measured with 2.8k bails of 500k pairs: 5.8 ms for the stage instead of 7.0).
            int64_t bail_tpp_min = 4096;
            if (const char* e = pmx::opt_str(pmx::O_ALIGN_BAIL_TPP_MIN)) bail_tpp_min = atoll(e);
            const bool skip_t0 = use_compact && n_t0 < bail_tpp_min;
            A.pair_perm = order;
            if (n_t0 > 0 && !skip_t0) launch_tpp(0, n_t0, nullptr, nullptr);
            A.pair_perm = nullptr;
            if (!use_compact) timer_end(ctx, "align_dom", 1);
            int64_t n_dp = 0;
            if (!skip_t0) read_counts(n_t1, n_dp, false);
            n_dp = std::min<int64_t>(n_dp, (int64_t)A.dp_slot_cap);
            al->last_dp_slots += n_dp;
            const uint32_t* cur = nullptr;   // round 1 serves slots 0..n_dp-1
            uint32_t* lists[2] = {al->dp_list_a.p, al->dp_list_b.p};
            int round = 1;
            int64_t n_small = 0;
            while (n_dp > 0) {   // ends by itself: a pair posts at most PMX_DP_MAX_CALLS requests, then goes to the wave tier
                if (n_dp < (round == 1 ? small_rounds : small_rounds / 4)) {   // remainder: wave-per-pair kernel over the slots (dp_slot_pairs maps them to pairs)
                    n_small = n_dp;
                    break;
                }
                al->last_dp_requests += n_dp;
                if (pmx::opt_str(pmx::O_DP_HIST)) {   // diagnostic: the shapes of the posted requests
                    std::vector<DpReq> h((size_t)n_dp * PMX_DP_REQ_PER_PASS);
                    PMX_HIP(hipMemcpyAsync(h.data(), A.dp_req_base, h.size() * sizeof(DpReq), hipMemcpyDeviceToHost, ctx->stream));
                    PMX_HIP(hipStreamSynchronize(ctx->stream));
                    std::map<std::tuple<int, int, int, int>, std::pair<long, long>> hist;   // (flag, q bucket, t bucket, band-free) -> (n, cells)
                    long n_req = 0;
                    for (const DpReq& r : h) {
                        if (r.call == 0xffffffffu) continue;
                        ++n_req;
                        const int w = r.w < 0 ? std::max(r.qlen, r.tlen) : r.w;
                        const int free_band = w >= std::max(r.qlen, r.tlen) - 1;
                        auto& e = hist[std::make_tuple(r.flag, (r.qlen + 15) / 16 * 16, (r.tlen + 31) / 32 * 32, free_band)];
                        ++e.first;
                        e.second += (long)r.qlen * r.tlen;
                    }
                    fprintf(stderr, "[pmx dp requests, round %d] %ld\n  flag  qlen<= tlen<= bandfree        n      cells\n", round, n_req);
                    for (auto& kv : hist)
                        fprintf(stderr, "  0x%02x %6d %6d %8d %8ld %10ld\n", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first), kv.second.first, kv.second.second);
                }
                if (dpg_ok) {
                    // the grouped service first (eight lanes per request: align_kernel_dpg.hip): it takes every request whose band
                    // never cuts its matrix and whose sides are <= 128 bases -- on 150 bp reads all of them -- and marks them served;
                    // the wave-per-request launches below see what is left
                    const int64_t n_ent = n_dp * PMX_DP_REQ_PER_PASS;
                    DG.dp_req_base = al->dp_req.p; DG.dp_res_base = al->dp_res.p; DG.stats = A.stats;
                    DG.n_entries = (uint32_t)std::min<size_t>(al->dp_req.n / sizeof(DpReq), UINT32_MAX);
                    const bool dpg_shadow = pmx::opt_str(pmx::O_DPG_SHADOW) != nullptr;   // diagnostic: both services run, results compared
                    if (dpg_shadow) {
                        al->dpg_shadow.ensure((size_t)n_items * PMX_DP_MAX_CALLS);
                        PMX_HIP(hipMemsetAsync(al->dpg_shadow.p, 0xee, sizeof(DpRes) * (size_t)n_items * PMX_DP_MAX_CALLS, ctx->stream));
                        DG.dp_res_base = al->dpg_shadow.p; DG.stats = nullptr; DG.shadow = 1;
                    }
                    if (pmx::opt_str(pmx::O_DPG_PROF)) { al->dpg_prof.ensure(8); PMX_HIP(hipMemsetAsync(al->dpg_prof.p, 0, 64, ctx->stream)); DG.prof = al->dpg_prof.p; }
                    dpg_launch(ctx, al, DG, n_dp, cur, dpg_waves, !pmx::opt_str(pmx::O_DPG_NO_SERVE));
                    if (pmx::opt_str(pmx::O_DPG_CHECK_LIST)) {   // diagnostic: the sorted request list against the bucket counts
                        std::vector<uint32_t> k2((size_t)n_ent), i2((size_t)n_ent), cn(16);
                        PMX_HIP(hipStreamSynchronize(ctx->stream));
                        PMX_HIP(hipMemcpy(k2.data(), al->dpg_keys2.p, (size_t)n_ent * 4, hipMemcpyDeviceToHost));
                        PMX_HIP(hipMemcpy(i2.data(), al->dpg_ids2.p, (size_t)n_ent * 4, hipMemcpyDeviceToHost));
                        PMX_HIP(hipMemcpy(cn.data(), al->dpg_counts.p, 64, hipMemcpyDeviceToHost));
                        long unsorted = 0, bad_id = 0, cnt[16] = {0};
                        for (int64_t i = 0; i < n_ent; ++i) {
                            if (i && k2[(size_t)i] < k2[(size_t)i - 1]) ++unsorted;
                            if (i2[(size_t)i] >= DG.n_entries) ++bad_id;
                            ++cnt[(k2[(size_t)i] >> 8) & 15];
                        }
                        fprintf(stderr, "[dpg list] %lld entries, %ld out of order, %ld ids out of range; buckets (device/host):", (long long)n_ent, unsorted, bad_id);
                        for (int b = 0; b < 16; ++b) fprintf(stderr, " %u/%ld", cn[(size_t)b], cnt[b]);
                        fprintf(stderr, "\n");
                    }
                    if (DG.prof) {
                        unsigned long long h[8];
                        PMX_HIP(hipMemcpyAsync(h, al->dpg_prof.p, 64, hipMemcpyDeviceToHost, ctx->stream));
                        PMX_HIP(hipStreamSynchronize(ctx->stream));
                        const double t = (double)std::max<unsigned long long>(h[4], 1);
                        fprintf(stderr, "[dpg prof] %llu tasks; cycles per task (lane 0 of the wave): set-up %.0f fill %.0f replay %.0f traceback %.0f; fill steps %.1f\n", h[4], h[0] / t, h[1] / t, h[2] / t, h[3] / t, h[5] / t);
                    }
                }
                A.dp_left = nullptr;
                bool wave_service = true;
                if (dpg_ok && !pmx::opt_str(pmx::O_DPG_SHADOW) && !pmx::opt_str(pmx::O_DPG_NO_SERVE)) {
                    // what did the grouped service leave?  Nothing: no launch of the wave service.  A handful (a side beyond 128
                    // bases on 150 bp reads: ~8 requests per 400k pairs): those pairs go to the wave-per-pair tier instead
                    uint32_t left[2] = {0, 0};
                    PMX_HIP(hipMemcpyAsync(left, al->dpg_counts.p + PMX_DPG_NO_BUCKET - 1, sizeof(left), hipMemcpyDeviceToHost, ctx->stream));
                    PMX_HIP(hipStreamSynchronize(ctx->stream));
                    int64_t few = 256;
                    if (const char* e = pmx::opt_str(pmx::O_DPG_LEFT_TO_WAVE_TIER)) few = atoll(e);
                    if (left[0] + left[1] == 0) wave_service = false;
                    else if ((int64_t)left[0] + left[1] <= few) {
                        hipLaunchKernelGGL(k_dpg_refuse_left, dim3((unsigned)std::min<int64_t>((n_dp * PMX_DP_REQ_PER_PASS + 255) / 256, (int64_t)ctx->n_cu * 8)), dim3(256), 0, ctx->stream, DG);
                        wave_service = false;
                    }
                }
                if (wave_service) {
                A.layout = dp_layout;
                A.slow_stride = dp_stride;
                A.slow_base = al->slow.p;
                A.n_items = n_dp;
                A.worklist = cur;
                A.dp_small_qlen = small_qlen;
                A.dp_small_tlen = small_tlen;
                A.dp_small_tb = (uint32_t)dps_layout.tb_cap;
                A.dp_class = dp_two_class ? 2 : 0;
                hipLaunchKernelGGL(k_align_dp_serve, dim3((unsigned)std::min<int64_t>(dp_max_grid, n_dp * PMX_DP_REQ_PER_PASS)), dim3(64), dp_lds, ctx->stream, A);
                if (dp_two_class) {
                    A.layout = dps_layout;
                    A.slow_stride = dps_stride;
                    A.slow_base = al->slow2.p;
                    A.dp_class = 1;
                    hipLaunchKernelGGL(k_align_dp_serve, dim3((unsigned)std::min<int64_t>(dps_max_grid, n_dp * PMX_DP_REQ_PER_PASS)), dim3(64), dps_lds, ctx->stream, A);
                }
                }
                PMX_HIP(hipGetLastError());
                if (dpg_ok && pmx::opt_str(pmx::O_DPG_SHADOW)) {
                    const size_t n_ent = (size_t)n_dp * PMX_DP_REQ_PER_PASS, n_res = (size_t)n_items * PMX_DP_MAX_CALLS;
                    std::vector<uint32_t> keys(n_ent), ids(n_ent);
                    std::vector<DpRes> a(n_res), b(n_res);
                    std::vector<DpReq> rq((size_t)n_items * PMX_DP_REQ_PER_PASS);
                    PMX_HIP(hipStreamSynchronize(ctx->stream));
                    PMX_HIP(hipMemcpy(keys.data(), al->dpg_keys.p, n_ent * 4, hipMemcpyDeviceToHost));
                    PMX_HIP(hipMemcpy(ids.data(), al->dpg_ids.p, n_ent * 4, hipMemcpyDeviceToHost));
                    PMX_HIP(hipMemcpy(a.data(), al->dp_res.p, n_res * sizeof(DpRes), hipMemcpyDeviceToHost));
                    PMX_HIP(hipMemcpy(b.data(), al->dpg_shadow.p, n_res * sizeof(DpRes), hipMemcpyDeviceToHost));
                    // (the requests were marked served by the wave service: their headers are intact apart from `call`, which the
                    //  collect pass read before; the call index is recovered from the result that carries the request's key)
                    long n_cmp = 0, n_bad = 0, shown = 0;
                    for (size_t i = 0; i < n_ent; ++i) {
                        if ((keys[i] >> 8) >= PMX_DPG_NO_BUCKET) continue;
                        const size_t slot = ids[i] / PMX_DP_REQ_PER_PASS;
                        for (int c = 0; c < PMX_DP_MAX_CALLS; ++c) {
                            const DpRes& y = b[slot * PMX_DP_MAX_CALLS + c];
                            if (y.key == 0xeeeeeeeeu) continue;   // not written by the grouped service
                            const DpRes& x = a[slot * PMX_DP_MAX_CALLS + c];
                            ++n_cmp;
                            bool same = x.key == y.key;
                            if (same && x.key != 0xffffffffu) {
                                same = memcmp(&x.ez, &y.ez, sizeof(Ez)) == 0;
                                for (int k = 0; same && k < x.ez.n_cigar && k < PMX_DP_MAX_CIGAR; ++k) same = x.cigar[k] == y.cigar[k];
                            }
                            if (!same) {
                                ++n_bad;
                                if (shown++ < 12) {
                                    fprintf(stderr, "[dpg shadow] slot %zu call %d key %08x/%08x\n  wave : max %u zd %d maxq %d maxt %d mqe %d mqe_t %d mte %d mte_q %d score %d ncig %d reach %d\n  group: max %u zd %d maxq %d maxt %d mqe %d mqe_t %d mte %d mte_q %d score %d ncig %d reach %d\n",
                                            slot, c, x.key, y.key, x.ez.max, x.ez.zdropped, x.ez.max_q, x.ez.max_t, x.ez.mqe, x.ez.mqe_t, x.ez.mte, x.ez.mte_q, x.ez.score, x.ez.n_cigar, x.ez.reach_end,
                                            y.ez.max, y.ez.zdropped, y.ez.max_q, y.ez.max_t, y.ez.mqe, y.ez.mqe_t, y.ez.mte, y.ez.mte_q, y.ez.score, y.ez.n_cigar, y.ez.reach_end);
                                    fprintf(stderr, "  wave cigar:");
                                    for (int k = 0; k < x.ez.n_cigar && k < PMX_DP_MAX_CIGAR; ++k) fprintf(stderr, " %u%c", x.cigar[k] >> 4, "MID"[x.cigar[k] & 3]);
                                    fprintf(stderr, "\n  group cigar:");
                                    for (int k = 0; k < y.ez.n_cigar && k < PMX_DP_MAX_CIGAR; ++k) fprintf(stderr, " %u%c", y.cigar[k] >> 4, "MID"[y.cigar[k] & 3]);
                                    fprintf(stderr, "\n");
                                }
                            }
                        }
                    }
                    fprintf(stderr, "[dpg shadow] round %d: %ld results compared, %ld differ\n", round, n_cmp, n_bad);
                }
                uint32_t* next = lists[round & 1];
                A.layout = tpp_layout;
                launch_tpp(round, n_dp, cur, next);
                read_counts(n_t1, n_dp, false);
                cur = next;
                ++round;
            }
            al->last_dp_rounds = std::max(al->last_dp_rounds, round - 1);
            A.dp_req_base = nullptr; A.dp_res_base = nullptr; A.dp_ncached = nullptr;
            A.dp_next_list = nullptr; A.dp_count = nullptr; A.dp_slot_cap = 0; A.dp_round = 0;
            if (n_small > 0) {   // their capacity overflows (rare) join the tier-1 retry list through counter [0]
                if (!cur) {      // round-1 remainder: slots are 0..n-1
                    std::vector<uint32_t> iota((size_t)n_small);
                    for (int64_t i = 0; i < n_small; ++i) iota[(size_t)i] = (uint32_t)i;
                    PMX_HIP(hipMemcpyAsync(lists[0], iota.data(), sizeof(uint32_t) * (size_t)n_small, hipMemcpyHostToDevice, ctx->stream));
                    PMX_HIP(hipStreamSynchronize(ctx->stream));
                    cur = lists[0];
                }
                launch(kern_t1, compact, n_small, cur, al->retry_list2.p, al->slow);   // A.dp_slot_pairs still set
                int64_t unused2 = 0;
                read_counts(n_t1, unused2, false);
            }
            A.dp_slot_pairs = nullptr;
            PMX_HIP(hipMemsetAsync(al->retry_count.p, 0, 2 * sizeof(unsigned long long), ctx->stream));
            t1_list = al->retry_list2.p;
            if (skip_t0) { t1_list = order; n_t1 = n_t0; }
            al->last_tpp_retry += n_t1;
            A.mv_handover = nullptr; A.mv_stride = 0; A.mv_slots = 0;
            run_wave_tiers(n_t1, t1_list);
            };   // run_tail
            if (use_compact) {
                al->bail_list.ensure((size_t)n_items);
                const bool pos16 = al->ri.len <= 32767 && !pmx::opt_str(pmx::O_ALIGN_COMPACT_POS32);
                const bool c_fused = pmx::opt_str(pmx::O_ALIGN_COMPACT_FUSED) != nullptr;
                auto c_kern = c_fused ? (pos16 ? k_align_compact16_fused : k_align_compact32_fused) : (pos16 ? k_align_compact16 : k_align_compact32);
                // (the first form of the two-kernel chain kernel keeps 48 anchors per pair: eight waves per CU; the fused kernels
                //  and the second form all 56)
                const size_t c_lds_full = (size_t)(pos16 ? PMX_C_LANE_WORDS16 : PMX_C_LANE_WORDS32) * 64 * sizeof(uint32_t) + PMX_C_PEN_BYTES;
                const size_t c_lds = c_fused ? c_lds_full : (size_t)(pos16 ? PMX_C_LANE_WORDS16_1 : PMX_C_LANE_WORDS32_1) * 64 * sizeof(uint32_t) + PMX_C_PEN_BYTES;
                // One workgroup (wave) per 64 pairs, handed out by the dispatcher as CUs free up: the pairs of a wave cost what
                // their worst lane costs, and with a resident grid striding over the positions (PMX_ALIGN_COMPACT_WAVES = waves
                // per CU brings it back) the slowest stride set the kernel's end -- 10M reads: 17.05 -> 15.5 ms, and the seeds
                // kernel below 4.77 -> 4.10 ms.  (The hardware keeps 160 KB / c_lds = seven waves per CU resident either way.)
                int64_t c_grid = (n_items + 63) / 64;
                if (const char* e = pmx::opt_str(pmx::O_ALIGN_COMPACT_WAVES)) c_grid = std::min<int64_t>((int64_t)ctx->n_cu * std::max(atoi(e), 1), c_grid);
                A.n_items = n_items;
                A.pair_perm = order;
                A.retry_list = al->bail_list.p;
                A.retry_count = al->retry_count.p + 2;
                // Two-kernel form (default): sketch + index probes in k_compact_seeds, whose only LDS is the minimizer queue
                // -- 7 KB per wave against the 21 KB of the pairs' work state, so that part runs at the occupancy its
                // registers allow instead of seven waves per CU; the seeds cross in HBM (224 bytes per pair with 16-bit
                // position words).  PMX_ALIGN_COMPACT_FUSED keeps everything in k_align_compact.
                if (!c_fused) {
                    const size_t blocks = (size_t)((n_items + 63) / 64);
                    al->cseeds.ensure(blocks * (size_t)PMX_C_CAP * (pos16 ? 1 : 2) * 64);
                    al->cseed_n.ensure(blocks * 64);
                    A.cseeds = al->cseeds.p;
                    A.cseed_n = al->cseed_n.p;
                    // (four waves per SIMD by the kernel's 113 VGPRs; a resident grid of 8 / 12 / 16 waves per CU -- PMX_ALIGN_CSEED_WAVES --
                    // takes 7.5 / 6.0 / 4.8 ms per 5M pairs, one workgroup per 64 pairs 4.1; the register budget of five waves per
                    // SIMD spills and gains 1 %, of six loses)
                    auto s_kern = pos16 ? k_compact_seeds16 : k_compact_seeds32;
                    int64_t s_grid = (n_items + 63) / 64;
                    if (const char* e = pmx::opt_str(pmx::O_ALIGN_CSEED_WAVES)) s_grid = std::min<int64_t>((int64_t)ctx->n_cu * std::max(atoi(e), 1), s_grid);
                    timer_begin(ctx, "align_cseeds");
                    hipLaunchKernelGGL(s_kern, dim3((unsigned)s_grid), dim3(64), ((size_t)PMX_C_SEEDQ * 2 + 8) * 64 * sizeof(uint32_t),   // queues + eight staging words per lane
                                       ctx->stream, A);
                    PMX_HIP(hipGetLastError());
                    timer_end(ctx, "align_cseeds", 1);
                }
                // PMX_ALIGN_EARLY_TAIL (off by default): the pairs the seeds kernel gave up on (a read with an `N`, a sketch tie,
                // a repeated minimizer, ...) are known now; list them and let the general tiers run them on the context's second
                // stream while this stream runs the chain kernels.  Measured at 10M reads (profiles/r04/README.md): the first
                // thread-per-pair pass does run beside k_align_compact16, the DP service behind it cannot (the chain kernel's
                // seven waves per CU leave 9.5 KB of LDS) and waits for its end, and the pairs the chain kernels hand back then
                // take a tail of their own instead of sharing one: 37.1 ms per step against 35.  Kept as a tested switch.
                int64_t n_early = 0;
                const bool c_early = !c_fused && pmx::opt_str(pmx::O_ALIGN_EARLY_TAIL) && !A.prof;
                if (c_early) {
                    al->early_list.ensure((size_t)n_items);
                    al->multi_count.ensure(4);
                    PMX_HIP(hipMemsetAsync(al->multi_count.p, 0, 4 * sizeof(unsigned long long), ctx->stream));
                    A.early_list = al->early_list.p;
                    A.early_count = al->multi_count.p + 2;
                    hipLaunchKernelGGL(k_compact_list_seed_bails, dim3((unsigned)std::min<int64_t>((n_items + 255) / 256, (int64_t)ctx->n_cu * 8)), dim3(256), 0, ctx->stream, A);
                    PMX_HIP(hipGetLastError());
                    unsigned long long h_early = 0;
                    PMX_HIP(hipMemcpyAsync(&h_early, A.early_count, sizeof(h_early), hipMemcpyDeviceToHost, ctx->stream));
                    if (!ctx->tail_go) {
                        PMX_HIP(hipEventCreateWithFlags(&ctx->tail_go, hipEventDisableTiming));
                        PMX_HIP(hipEventCreateWithFlags(&ctx->tail_done, hipEventDisableTiming));
                    }
                    PMX_HIP(hipEventRecord(ctx->tail_go, ctx->stream));   // the list is made: what the second stream waits for
                    PMX_HIP(hipStreamSynchronize(ctx->stream));
                    n_early = (int64_t)h_early;
                    A.seed_bails_listed = 1;
                }
                // Second form (k_align_compact*_multi): the pairs that leave the first one after their seeds -- a third chain, two
                // regions on one mate (mates that overlap on the reference: 55 % of the real example pairs), ... -- are run again
                // from their hand-over words with up to four chains and several regions per mate; what is still left goes to the
                // thread-per-pair tier.  PMX_ALIGN_NO_MULTI: every bail goes there at once.
                const bool c_multi = !c_fused && !pmx::opt_str(pmx::O_ALIGN_NO_MULTI);
                if (c_multi) {
                    al->multi_list.ensure((size_t)n_items);
                    al->multi_count.ensure(4);
                    PMX_HIP(hipMemsetAsync(al->multi_count.p, 0, 2 * sizeof(unsigned long long), ctx->stream));
                    A.multi_list = al->multi_list.p;
                    A.multi_count = al->multi_count.p;
                }
                timer_begin(ctx, "align_dom");   // the dominant kernel on its own (bench.py roofline)
                hipLaunchKernelGGL(c_kern, dim3((unsigned)c_grid), dim3(64), c_lds, ctx->stream, A);
                PMX_HIP(hipGetLastError());
                timer_end(ctx, "align_dom", 1);
                if (c_multi) {
                    // (the list's length stays on the device: a resident grid -- seven waves per CU by the LDS -- strides over it)
                    const int64_t m_grid = std::min<int64_t>((n_items + 63) / 64, (int64_t)ctx->n_cu * 7);
                    al->multi_ws.ensure((size_t)m_grid * PMX_CM_WS_WORDS * 64);
                    A.multi_ws = al->multi_ws.p;
                    timer_begin(ctx, "align_cmulti");
                    hipLaunchKernelGGL(pos16 ? k_align_compact16_multi : k_align_compact32_multi, dim3((unsigned)m_grid), dim3(64), c_lds_full, ctx->stream, A);
                    PMX_HIP(hipGetLastError());
                    timer_end(ctx, "align_cmulti", 1);
                }
                A.cseeds = nullptr; A.cseed_n = nullptr;
                A.multi_list = nullptr; A.multi_count = nullptr; A.multi_ws = nullptr;
                A.early_list = nullptr; A.early_count = nullptr; A.seed_bails_listed = 0;
                if (n_early > 0) {
                    // the early tail: same host code, enqueued on the second stream (every launch, copy and wait of the tail goes
                    // through ctx->stream); the chain kernels are already queued on the first
                    if (!ctx->pair_stream) ctx->pair_stream = create_dedicated_stream(ctx->n_cu);
                    hipStream_t main_stream = ctx->stream;
                    PMX_HIP(hipStreamWaitEvent(ctx->pair_stream, ctx->tail_go, 0));
                    ctx->stream = ctx->pair_stream;
                    try { run_tail(al->early_list.p, n_early); } catch (...) { ctx->stream = main_stream; throw; }
                    PMX_HIP(hipEventRecord(ctx->tail_done, ctx->stream));
                    ctx->stream = main_stream;
                    early_running = true;
                }
                if (A.prof) {   // the compact tier's own phase profile, then the accumulators start over for the general tiers
                    unsigned long long h[8];
                    PMX_HIP(hipMemcpyAsync(h, al->prof.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
                    PMX_HIP(hipStreamSynchronize(ctx->stream));
                    PMX_HIP(hipMemsetAsync(al->prof.p, 0, 32 * sizeof(unsigned long long), ctx->stream));
                    static const char* cn[8] = {"sketch", "probes", "merge", "chain fill", "backtrack", "regions", "align+mapq", "pairing"};
                    const double waves = (double)((n_items + 63) / 64);
                    fprintf(stderr, "[pmx compact tier: cycles per wave (lane 0)]");
                    for (int k = 0; k < 8; ++k) fprintf(stderr, " %s=%.0f", cn[k], (double)h[k] / waves);
                    fprintf(stderr, "\n");
                }
                unsigned long long h_bail = 0;
                PMX_HIP(hipMemcpyAsync(&h_bail, al->retry_count.p + 2, sizeof(h_bail), hipMemcpyDeviceToHost, ctx->stream));
                PMX_HIP(hipStreamSynchronize(ctx->stream));
                n_t0 = (int64_t)h_bail;
                order = al->bail_list.p;
                al->last_compact = n_items - n_t0 - n_early;
                A.retry_list = al->retry_list2.p;
                A.retry_count = al->retry_count.p;
            }
            if (!use_compact) run_tail(order, n_items);
            else {
                // the pairs the chain kernels hand back; the early tail (if one runs) must be through with the shared work buffers
                if (early_running) {
                    PMX_HIP(hipStreamWaitEvent(ctx->stream, ctx->tail_done, 0));
                    PMX_HIP(hipEventSynchronize(ctx->tail_done));
                }
                run_tail(al->bail_list.p, n_t0);
            }
        }
    } else {
        // Long reads (map-ont / map-hifi branch): wave per read with the general capacities.  The band of those presets
        // allows traceback matrices up to max_sw_mat bytes (100 MB) although nearly every DP between two anchors is a few
        // hundred bases wide: the first launch gives every wave 8 MB of traceback in HBM, the reads that need more come
        // back on the retry list and run in a second launch of few waves with the full capacity.
        size_t tb_small = (size_t)8 << 20;
        if (const char* e = pmx::opt_str(pmx::O_ALIGN_TB_KB)) tb_small = std::max<size_t>((size_t)atoll(e), 1) << 10;
        // The arrays of a 10 kb read (anchors, chain cells, the DP arrays sized for the longest allowed target) live in the
        // wave's HBM slab whatever the LDS budget, and the kernel is bound by the latency of those accesses: what counts is
        // resident waves (16 per CU: 32.4 k reads/s, 8 per CU: 21.1 k) and that the DPs -- nearly all a few hundred bases
        // wide -- run on a small LDS copy of their arrays (plan_layout dp_fast_tlen)
        int dp_fast = PMX_DP_FAST_TLEN;
        if (pmx::opt_str(pmx::O_ALIGN_NO_DP_FAST)) dp_fast = 0;
        size_t lr_budget = 8900;
        if (const char* e = pmx::opt_str(pmx::O_ALIGN_LDS_KB)) lr_budget = (size_t)atoi(e) * 1024;
        const Layout g1 = hooked(plan_layout((int)rs->max_len, n_segs, al->opt, lr_budget, tb_small, 1, dp_fast));
        al->retry_list.ensure((size_t)n_items);
        timer_begin(ctx, "align_dom");
        if (!pmx::opt_str(pmx::O_ALIGN_NO_WORK_QUEUE)) {
            PMX_HIP(hipMemsetAsync(al->retry_count.p + 3, 0, sizeof(unsigned long long), ctx->stream));
            A.work_queue = al->retry_count.p + 3;
        }
        launch(kern, g1, n_items, nullptr, general.tb_cap > g1.tb_cap ? al->retry_list.p : nullptr, al->slow2);
        A.work_queue = nullptr;
        timer_end(ctx, "align_dom", 1);
        int64_t n_retry = 0, unused = 0;
        read_counts(n_retry, unused, true);
        al->last_retry = n_retry;
        if (n_retry > 0) {
            const size_t stride = (general.slow_bytes + 255) & ~(size_t)255;
            const int64_t big_grid = std::max<int64_t>(1, std::min<int64_t>(64, (int64_t)(((size_t)24 << 30) / std::max<size_t>(stride, 1))));
            launch(kern, general, n_retry, al->retry_list.p, nullptr, al->slow, big_grid);
        }
    }
    timer_end(ctx, "align", 1);
    if (A.prof) {
        unsigned long long h[32];
        PMX_HIP(hipMemcpyAsync(h, al->prof.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        PMX_HIP(hipStreamSynchronize(ctx->stream));
        static const char* names[16] = {"decode", "sketch", "seed+heap", "chain", "gen_regs+post", "seg_gen", "squeeze", "align1(all regs)", "filter/sort/parent", "mapq", "pair", "output", "", "", "", ""};
        fprintf(stderr, "[pmx align phase cycles per item]");
        for (int k = 0; k < 12; ++k) fprintf(stderr, " %s=%.0f", names[k], (double)h[k] / (double)n_items);
        // sub-phases (thread-per-pair kernel): "seed+heap" then holds only the heap merge, "chain" only the compaction
        fprintf(stderr, " [of which index lookups=%.0f stage+heapify=%.0f chain fill=%.0f backtrack=%.0f]", (double)h[16] / (double)n_items,
                (double)h[17] / (double)n_items, (double)h[18] / (double)n_items, (double)h[19] / (double)n_items);
        // align1 (thread-per-pair kernel): "align1(all regs)" then holds only what follows the right extension
        fprintf(stderr, " [align1: prologue+filters=%.0f left ext=%.0f gap fills=%.0f right ext=%.0f]", (double)h[20] / (double)n_items,
                (double)h[21] / (double)n_items, (double)h[22] / (double)n_items, (double)h[23] / (double)n_items);
        fprintf(stderr, " [dp serve: cycles traceback=%.0f ksw=%.0f store=%.0f, anti-diagonals filled=%.1f per pair that posted requests]", (double)h[12] / std::max<double>(1, (double)al->last_dp_requests),
                (double)h[13] / std::max<double>(1, (double)al->last_dp_requests), (double)h[14] / std::max<double>(1, (double)al->last_dp_requests),
                (double)h[15] / std::max<double>(1, (double)al->last_dp_requests));
        fprintf(stderr, " dp_requests=%lld dp_rounds=%d tpp_retry=%lld retry=%lld\n", (long long)al->last_dp_requests, al->last_dp_rounds,
                (long long)al->last_tpp_retry, (long long)al->last_retry);
        if (!tier1_fits)   // wave-per-read kernels: slots 23..31 count the DPs by the kernel that ran them
            fprintf(stderr, "[pmx long-read DPs] row by row: %llu calls, %.1f Mcells, %.0f cycles each; anti-diagonals in LDS: %llu calls, %.1f Mcells, %.0f cycles each; anti-diagonals, general arrays: %llu calls, %.1f Mcells, %.0f cycles each\n",
                    h[23], h[24] / 1e6, (double)h[25] / std::max<double>(1, (double)h[23]), h[26], h[27] / 1e6, (double)h[28] / std::max<double>(1, (double)h[26]),
                    h[29], h[30] / 1e6, (double)h[31] / std::max<double>(1, (double)h[29]));
    }
    PMX_HIP(hipGetLastError());
    return PMX_OK;
    PMX_CATCH
}

// The CIGAR arena is sized optimistically (16 words per read); the kernels count what they WOULD have written
// exactly (cigar_used runs past the capacity), so a call that overflowed is redone once with the counted size:
// no record ever leaves this function with PMX_REC_OVERFLOW set because of the arena.
int pmx_align_readset(pmx_ctx* ctx, pmx_aligner* al, const pmx_readset* rs, int paired, int revcomp_mate2) {
    if (!ctx || !al || !rs) return PMX_ERR_ARG;
    if (!rs->packed) return fail(PMX_ERR_ARG, "read set is not packed (call pmx_readset_pack first)");
    uint64_t cap = (uint64_t)std::max<int64_t>(rs->n * 16, 4096);
    // long reads: an operation every ~20 bases at 5 % errors; and whatever the previous call on this aligner needed per base
    // (a redo costs the whole stage again: 4.5 s per 100k reads of 10 kb)
    if (!al->opt.is_sr_like) cap = std::max<uint64_t>(cap, (uint64_t)rs->total / 8 + (uint64_t)rs->n * 16);
    if (al->cigar_words_per_kbase > 0.0) cap = std::max<uint64_t>(cap, (uint64_t)(al->cigar_words_per_kbase * 1.25 * (double)rs->total / 1000.0) + 4096);
    if (const char* e = pmx::opt_str(pmx::O_ALIGN_CIGAR_CAP)) cap = (uint64_t)std::max<long long>(atoll(e), 16);   // tests: force the redo
    for (int attempt = 0;; ++attempt) {
        const int rc = align_readset_once(ctx, al, rs, paired, revcomp_mate2, cap);
        if (rc != PMX_OK) return rc;
        unsigned long long used = 0, st[4] = {0, 0, 0, 0};
        PMX_TRY
        PMX_HIP(hipMemcpyAsync(&used, al->cigar_used.p, sizeof(used), hipMemcpyDeviceToHost, ctx->stream));
        PMX_HIP(hipMemcpyAsync(st, al->stats.p, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
        PMX_HIP(hipStreamSynchronize(ctx->stream));
        PMX_CATCH
        al->last_cigar_used = used;
        al->last_stats.dp_calls = (int64_t)st[0];
        al->last_stats.dp_cells = (int64_t)st[1];
        al->last_stats.dp_pairs = al->last_dp_slots + (int64_t)st[2];
        al->last_stats.dp_rounds = al->last_dp_rounds;
        al->last_stats.wave_tier_items = al->last_tpp_retry;
        al->last_stats.general_tier_items = al->last_retry;
        al->last_stats.compact_tier_items = al->last_compact;
        if (rs->total > 0) al->cigar_words_per_kbase = std::max(al->cigar_words_per_kbase * 0.5, (double)used * 1000.0 / (double)rs->total);
        if (used <= cap) return PMX_OK;
        if (attempt >= 2) return fail(PMX_ERR_CAPACITY, "CIGAR arena overflow persists after resizing");
        cap = used + 64;
    }
}

// score_reads_vs_reference (src/mm_align.c:144-199): minus the summed count_read_errors of every read against the
// aligner's reference -- the alignment-based score of one --refine candidate (src/placement.cpp:489-514)
static int score_reads_impl(pmx_ctx* ctx, pmx_aligner* al, const pmx_readset* rs, int paired, int revcomp_mate2, int64_t* score, int64_t* n_flagged) {
    al->want_edits = true;
    const int rc = pmx_align_readset(ctx, al, rs, paired, revcomp_mate2);
    al->want_edits = false;
    if (rc != PMX_OK) return rc;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    al->stats.ensure(4);
    PMX_HIP(hipMemsetAsync(al->stats.p, 0, 2 * sizeof(unsigned long long), ctx->stream));
    const int64_t n = al->n_records;
    if (n > 0)
        hipLaunchKernelGGL(k_sum_edits, dim3((unsigned)std::min<int64_t>((n + 255) / 256, (int64_t)ctx->n_cu * 8)), dim3(256), 0, ctx->stream, al->records.p,
                           al->edits.p, n, al->stats.p, n_flagged ? rs->off.p : (const int64_t*)nullptr);
    unsigned long long h[2] = {0, 0};
    PMX_HIP(hipMemcpyAsync(h, al->stats.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    if (n_flagged) *n_flagged = (int64_t)h[1];
    else if (h[1]) return fail(PMX_ERR_UNSUPPORTED, "reads with flagged (overflow / unsupported) records: their edit counts are not the reference's");
    *score = -(int64_t)h[0];
    return PMX_OK;
    PMX_CATCH
}

int pmx_align_score_reads(pmx_ctx* ctx, pmx_aligner* al, const pmx_readset* rs, int paired, int revcomp_mate2, int64_t* score) {
    if (!ctx || !al || !rs || !score) return PMX_ERR_ARG;
    if (paired && (rs->n & 1)) return fail(PMX_ERR_UNSUPPORTED, "paired scoring of an odd number of reads (the reference maps the last one alone: pmx_score_reads_vs_reference does)");
    return score_reads_impl(ctx, al, rs, paired, revcomp_mate2, score, nullptr);
}

// Drop-in for score_reads_vs_reference (src/mm_align.h:13-17, src/mm_align.c:144-199), the reference's own signature: host
}  // extern "C"

// Contexts of the two drop-in boundaries below.  The reference calls them from several TBB workers at a time, call after
// call; a context per call would create and destroy its hardware queues (CU-masked streams) over and over, beside the
// running queues of the other workers -- the pattern behind the hang of round 4 (a destroyed stream's queue recycled under
// the next one's kernels) and one hung run of tests/test_align_gpu.py::test_direct_boundary_is_reentrant.  The contexts are
// kept: a call borrows one (or makes one: there are never more than the workers that were inside at the same time) and
// hands it back; they live as long as the process.
namespace {
std::mutex g_ctx_pool_mu;
std::vector<pmx_ctx*> g_ctx_pool;
pmx_ctx* borrow_ctx(int dev) {
    {
        std::lock_guard<std::mutex> lk(g_ctx_pool_mu);
        for (size_t i = 0; i < g_ctx_pool.size(); ++i)
            if (g_ctx_pool[i]->device == dev) {
                pmx_ctx* c = g_ctx_pool[i];
                g_ctx_pool.erase(g_ctx_pool.begin() + (long)i);
                if (hipSetDevice(dev) != hipSuccess) { (void)hipGetLastError(); g_ctx_pool.push_back(c); return nullptr; }
                return c;
            }
    }
    pmx_ctx* c = nullptr;
    return pmx_ctx_create(dev, &c) == PMX_OK ? c : nullptr;
}
void return_ctx(pmx_ctx* c) {
    if (!c) return;
    if (hipStreamSynchronize(c->stream) != hipSuccess) {   // a context whose stream failed is not handed to the next caller
        (void)hipGetLastError();
        pmx_ctx_destroy(c);
        return;
    }
    std::lock_guard<std::mutex> lk(g_ctx_pool_mu);
    g_ctx_pool.push_back(c);
}
}  // namespace

extern "C" {

// strings in, minus the total edit distance out; 0 on failure, as the reference returns 0 when its index cannot be built.
// An odd read of a paired set is mapped alone (:178-185).  A pair whose record a kernel flagged invalid counts as unmapped
// reads (their lengths), the way pmx_align_reads_direct reports such pairs; pmx_last_error() says how many there were.
int64_t pmx_score_reads_vs_reference(const char* reference, int n_reads, const char** reads, const int* r_lens, int kmer_size, bool paired_end) {
    (void)kmer_size;
    if (!reference || !reads || !r_lens || n_reads <= 0) return 0;
    pmx_ctx* ctx = nullptr;
    int dev = 0;
    if (const char* e = pmx::opt_str(pmx::O_DEVICE)) dev = atoi(e);
    if (!(ctx = borrow_ctx(dev))) return 0;
    pmx_aligner* al = nullptr;
    int64_t total = 0, withheld = 0;
    bool ok = false;
    do {
        std::vector<int64_t> off((size_t)n_reads + 1, 0);
        for (int i = 0; i < n_reads; ++i) off[(size_t)i + 1] = off[(size_t)i] + r_lens[i];
        std::string concat;
        concat.reserve((size_t)off[(size_t)n_reads]);
        for (int i = 0; i < n_reads; ++i) concat.append(reads[i], (size_t)r_lens[i]);
        for (char& ch : concat)
            if ((unsigned char)ch < 4) ch = "ACGT"[(unsigned char)ch];   // (pre-encoded bases, as in pmx_align_reads_direct)
        const int avg_len = (int)(off[(size_t)n_reads] / n_reads);        // setup_minimap2 looks at every read (src/mm_align.c:124-130)
        if (pmx_aligner_create(ctx, reference, (int64_t)strlen(reference), avg_len, &al) != PMX_OK) break;
        const bool paired = paired_end && n_reads >= 2;
        const int n_main = paired ? n_reads & ~1 : n_reads;
        auto run = [&](int first, int count, int as_pairs) {
            pmx_readset* rs = nullptr;
            if (pmx_readset_upload(ctx, concat.data(), off.data() + first, count, &rs) != PMX_OK) return false;
            int64_t sc = 0, flagged = 0;
            const bool good = pmx_readset_pack(ctx, rs) == PMX_OK && score_reads_impl(ctx, al, rs, as_pairs, 0, &sc, &flagged) == PMX_OK;
            pmx_readset_free(ctx, rs);
            total += sc;
            withheld += flagged;
            return good;
        };
        if (!run(0, n_main, paired ? 1 : 0)) break;
        if (n_main < n_reads && !run(n_main, 1, 0)) break;   // the odd read, alone
        ok = true;
    } while (0);
    if (al) pmx_aligner_free(ctx, al);
    return_ctx(ctx);
    if (ok && withheld) set_error("pmx_score_reads_vs_reference: " + std::to_string(withheld) + " read(s) of flagged records counted as unmapped");
    return ok ? total : 0;
}

int64_t pmx_align_num_records(const pmx_aligner* al) { return al ? al->n_records : 0; }

int64_t pmx_align_cigar_words(pmx_ctx* ctx, pmx_aligner* al) {
    if (!ctx || !al) return PMX_ERR_ARG;
    return (int64_t)std::min<unsigned long long>(al->last_cigar_used, al->cigar_cap);
}

int pmx_align_fetch(pmx_ctx* ctx, pmx_aligner* al, pmx_aln_record* records, int64_t n_records, uint32_t* cigar_arena, int64_t arena_cap) {
    if (!ctx || !al || !records || n_records < al->n_records) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    const int64_t used = pmx_align_cigar_words(ctx, al);
    if (used < 0) return (int)used;
    if (used > arena_cap || (used > 0 && !cigar_arena)) return fail(PMX_ERR_CAPACITY, "CIGAR arena buffer too small");
    if (al->n_records > 0)
        PMX_HIP(hipMemcpyAsync(records, al->records.p, sizeof(AlnRecord) * (size_t)al->n_records, hipMemcpyDeviceToHost, ctx->stream));
    if (used > 0) PMX_HIP(hipMemcpyAsync(cigar_arena, al->cigars.p, sizeof(uint32_t) * (size_t)used, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
    PMX_CATCH
}

int pmx_align_scoring(const pmx_aligner* al, int32_t out[9]) {
    if (!al || !out) return PMX_ERR_ARG;
    const Opt& o = al->opt;
    const int32_t v[9] = {o.a, o.b, o.q, o.e, o.q2, o.e2, o.sc_ambi, o.zdrop, o.end_bonus};
    memcpy(out, v, sizeof(v));
    return PMX_OK;
}

// ksw_extd2 on a batch of sequence pairs through the grouped DP service (include/panmap_amd.h): one request slot per pair
int pmx_align_dp_batch(pmx_ctx* ctx, pmx_aligner* al, const uint8_t* seqs, const int64_t* q_off, const int64_t* t_off, int64_t n, const int32_t* w,
                       const int32_t* zdrop, const int32_t* end_bonus, const int32_t* flag, pmx_dp_result* out, int reps, double* kernel_ms) {
    if (!ctx || !al || n < 0 || (n > 0 && (!seqs || !q_off || !t_off || !w || !zdrop || !end_bonus || !flag || !out))) return PMX_ERR_ARG;
    if (n > (int64_t)(UINT32_MAX / PMX_DP_REQ_PER_PASS) - 1) return PMX_ERR_CAPACITY;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    if (kernel_ms) *kernel_ms = 0;
    if (n == 0) return PMX_OK;
    DpgArgs DG;
    const bool ok = dpg_setup(al->opt, DG);
    std::vector<DpReq> req((size_t)n * PMX_DP_REQ_PER_PASS);
    for (int64_t i = 0; i < n; ++i) {
        for (int j = 0; j < PMX_DP_REQ_PER_PASS; ++j) req[(size_t)i * PMX_DP_REQ_PER_PASS + j].call = 0xffffffffu;
        DpReq& r = req[(size_t)i * PMX_DP_REQ_PER_PASS];
        const int64_t ql = q_off[i + 1] - q_off[i], tl = t_off[i + 1] - t_off[i];
        if (ql < 0 || tl < 0) return fail(PMX_ERR_ARG, "pmx_align_dp_batch: descending offsets");
        r.qlen = (int32_t)std::min<int64_t>(ql, INT32_MAX); r.tlen = (int32_t)std::min<int64_t>(tl, INT32_MAX);
        r.w = w[i]; r.zdrop = zdrop[i]; r.end_bonus = end_bonus[i]; r.flag = flag[i];
        r.key = (uint32_t)i;
        if (ql >= 1 && tl >= 1 && ((ql + 15) & ~(int64_t)15) + tl <= PMX_DP_SEQ_BYTES) {   // (longer ones cannot be posted: left unserved)
            r.call = 0;
            memset(r.seq, 0, sizeof(r.seq));
            memcpy(r.seq, seqs + q_off[i], (size_t)ql);
            memcpy(r.seq + ((ql + 15) & ~(int64_t)15), seqs + t_off[i], (size_t)tl);
        }
    }
    DevBuf<uint8_t> d_req;
    DevBuf<DpRes> d_res;
    d_req.alloc(req.size() * sizeof(DpReq));
    d_res.alloc((size_t)n * PMX_DP_MAX_CALLS);
    PMX_HIP(hipMemcpyAsync(d_req.p, req.data(), req.size() * sizeof(DpReq), hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemsetAsync(d_res.p, 0xff, sizeof(DpRes) * (size_t)n * PMX_DP_MAX_CALLS, ctx->stream));
    DG.dp_req_base = d_req.p; DG.dp_res_base = d_res.p;
    DG.n_entries = (uint32_t)req.size();
    DG.shadow = 1;   // the requests stay posted: the launch can be repeated
    int dpg_waves = 8;
    if (const char* e = pmx::opt_str(pmx::O_ALIGN_DPG_WAVES)) dpg_waves = std::max(1, atoi(e));
    if (ok) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        PMX_HIP(hipEventCreate(&e0)); PMX_HIP(hipEventCreate(&e1));
        float best = 0;
        for (int rep = 0; rep < std::max(reps, 1); ++rep) {
            dpg_launch(ctx, al, DG, n, nullptr, dpg_waves, false);   // collect + order
            PMX_HIP(hipEventRecord(e0, ctx->stream));
            const int64_t grid = std::min<int64_t>((int64_t)ctx->n_cu * dpg_waves, (n * PMX_DP_REQ_PER_PASS + 7) / 8 + PMX_DPG_BUCKETS);
            hipLaunchKernelGGL(k_align_dp_group, dim3((unsigned)grid), dim3(64), PMX_DPG_LDS_BYTES, ctx->stream, DG);
            PMX_HIP(hipEventRecord(e1, ctx->stream));
            PMX_HIP(hipEventSynchronize(e1));
            float ms = 0;
            PMX_HIP(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 0 || ms < best) best = ms;
        }
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        if (kernel_ms) *kernel_ms = best;
    }
    std::vector<DpRes> res((size_t)n * PMX_DP_MAX_CALLS);
    PMX_HIP(hipMemcpyAsync(res.data(), d_res.p, res.size() * sizeof(DpRes), hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    for (int64_t i = 0; i < n; ++i) {
        const DpRes& R = res[(size_t)i * PMX_DP_MAX_CALLS];
        pmx_dp_result& o = out[i];
        memset(&o, 0, sizeof(o));
        if (R.key != (uint32_t)i) continue;   // not taken (0xffffffff: never written, or more CIGAR operations than a result holds)
        o.served = 1;
        o.max = R.ez.max; o.zdropped = R.ez.zdropped; o.max_q = R.ez.max_q; o.max_t = R.ez.max_t; o.mqe = R.ez.mqe; o.mqe_t = R.ez.mqe_t;
        o.mte = R.ez.mte; o.mte_q = R.ez.mte_q; o.score = R.ez.score; o.n_cigar = R.ez.n_cigar; o.reach_end = R.ez.reach_end;
        for (int k = 0; k < R.ez.n_cigar && k < PMX_DP_MAX_CIGAR; ++k) o.cigar[k] = R.cigar[k];
    }
    return PMX_OK;
    PMX_CATCH
}

// The download of pmx_align_fetch on a stream of the caller's choice, without waiting for it: the copies start when the
// results are complete (event on the context's stream) and the next pmx_align_readset on this aligner waits for them
// before it overwrites the buffers.  The caller synchronises `stream` before it reads the host buffers (pinned memory, or
// the copies are staged).  stream NULL = the context's stream.
int pmx_align_fetch_async(pmx_ctx* ctx, pmx_aligner* al, pmx_aln_record* records, int64_t n_records, uint32_t* cigar_arena, int64_t arena_cap, void* stream) {
    if (!ctx || !al || !records || n_records < al->n_records) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    const int64_t used = pmx_align_cigar_words(ctx, al);
    if (used < 0) return (int)used;
    if (used > arena_cap || (used > 0 && !cigar_arena)) return fail(PMX_ERR_CAPACITY, "CIGAR arena buffer too small");
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    if (!al->ev_results) {
        PMX_HIP(hipEventCreateWithFlags(&al->ev_results, hipEventDisableTiming));
        PMX_HIP(hipEventCreateWithFlags(&al->ev_fetched, hipEventDisableTiming));
    }
    if (st != ctx->stream) {
        PMX_HIP(hipEventRecord(al->ev_results, ctx->stream));
        PMX_HIP(hipStreamWaitEvent(st, al->ev_results, 0));
    }
    if (al->n_records > 0) PMX_HIP(hipMemcpyAsync(records, al->records.p, sizeof(AlnRecord) * (size_t)al->n_records, hipMemcpyDeviceToHost, st));
    if (used > 0) PMX_HIP(hipMemcpyAsync(cigar_arena, al->cigars.p, sizeof(uint32_t) * (size_t)used, hipMemcpyDeviceToHost, st));
    if (st != ctx->stream) {
        PMX_HIP(hipEventRecord(al->ev_fetched, st));
        al->fetch_pending = true;
    }
    return PMX_OK;
    PMX_CATCH
}

int pmx_align_copy_records_device(pmx_ctx* ctx, pmx_aligner* al, void* d_records, int64_t n_records) {
    if (!ctx || !al || !d_records || n_records < al->n_records) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    if (al->n_records > 0)
        PMX_HIP(hipMemcpyAsync(d_records, al->records.p, sizeof(AlnRecord) * (size_t)al->n_records, hipMemcpyDeviceToDevice, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
    PMX_CATCH
}

int pmx_align_copy_cigars_device(pmx_ctx* ctx, pmx_aligner* al, void* d_cigars, int64_t n_words) {
    if (!ctx || !al || (!d_cigars && n_words > 0)) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    const int64_t used = pmx_align_cigar_words(ctx, al);
    if (n_words < used) return fail(PMX_ERR_CAPACITY, "CIGAR arena buffer too small");
    if (used > 0) PMX_HIP(hipMemcpyAsync(d_cigars, al->cigars.p, sizeof(uint32_t) * (size_t)used, hipMemcpyDeviceToDevice, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
    PMX_CATCH
}

int pmx_align_get_stats(pmx_ctx* ctx, pmx_aligner* al, pmx_align_stats* out) {
    if (!ctx || !al || !out) return PMX_ERR_ARG;
    *out = al->last_stats;
    return PMX_OK;
}

const void* pmx_align_device_records(const pmx_aligner* al) { return al ? al->records.p : nullptr; }
const void* pmx_align_device_cigars(const pmx_aligner* al) { return al ? al->cigars.p : nullptr; }

// Drop-in for align_reads_direct (src/mm_align.h:44-53): host strings in, read_align_t out.
void pmx_align_reads_direct(const char* reference, const char* refName, int n_reads, const char** reads, const char** quality,
                            const char** read_names, const int* r_lens, align_pair_result_t* results, bool pairedEndReads, int n_threads) {
    (void)refName; (void)quality; (void)read_names; (void)n_threads;
    if (!reference || !reads || !r_lens || !results || n_reads <= 0) return;
    pmx_ctx* ctx = nullptr;
    pmx_readset* rs = nullptr;
    pmx_aligner* al = nullptr;
    int dev = 0;
    if (const char* e = pmx::opt_str(pmx::O_DEVICE)) dev = atoi(e);
    if (!(ctx = borrow_ctx(dev))) return;   // like the reference: results stay untouched on failure
    do {
        int64_t total = 0;
        std::vector<int64_t> off((size_t)n_reads + 1, 0);
        for (int i = 0; i < n_reads; ++i) { off[i] = total; total += r_lens[i]; }
        off[n_reads] = total;
        std::string concat;
        concat.reserve((size_t)total);
        for (int i = 0; i < n_reads; ++i) concat.append(reads[i], (size_t)r_lens[i]);
        // seq_nt4_table (sketch.c:9-26) passes the bytes 0..3 through as pre-encoded bases; the packed read set knows letters
        // only (those bytes would be ambiguous), so they become letters here
        for (char& ch : concat)
            if ((unsigned char)ch < 4) ch = "ACGT"[(unsigned char)ch];
        const int avg_len = (int)(total / n_reads);   // src/mm_align.c:124-130
        if (pmx_readset_upload(ctx, concat.data(), off.data(), n_reads, &rs) != PMX_OK) break;
        if (pmx_readset_pack(ctx, rs) != PMX_OK) break;
        if (pmx_aligner_create(ctx, reference, (int64_t)strlen(reference), avg_len, &al) != PMX_OK) break;
        if (pmx_align_readset(ctx, al, rs, pairedEndReads ? 1 : 0, 0) != PMX_OK) break;
        std::vector<pmx_aln_record> recs((size_t)n_reads);
        const int64_t words = pmx_align_cigar_words(ctx, al);
        if (words < 0) break;
        std::vector<uint32_t> arena((size_t)std::max<int64_t>(words, 1));
        if (pmx_align_fetch(ctx, al, recs.data(), n_reads, arena.data(), (int64_t)arena.size()) != PMX_OK) break;
        auto fill = [&](const pmx_aln_record& r, read_align_t* o) {
            memset(o, 0, sizeof(*o));
            if (r.mapped && (r.flags & PMX_REC_HAS_ALN)) {
                o->pos = r.rs + 1; o->rs = r.rs; o->re = r.re; o->qs = r.qs; o->qe = r.qe;
                o->mapq = r.mapq; o->rev = r.rev; o->proper_frag = r.proper_frag;
                o->n_cigar = r.n_cigar;
                o->cigar = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)std::max<int>(r.n_cigar, 1));
                memcpy(o->cigar, arena.data() + r.cigar_off, sizeof(uint32_t) * r.n_cigar);
            } else o->pos = INT_MAX;
        };
        const int n_items = pairedEndReads ? n_reads / 2 : n_reads;
        // A record flagged PMX_REC_OVERFLOW / PMX_REC_UNSUPPORTED is INVALID (a work capacity or an unrestated
        // branch of the reference was hit): it never leaves the boundary as an alignment.  The pair is reported
        // unmapped and pmx_last_error() says how many were withheld.
        int64_t n_withheld = 0;
        for (int k = 0; k < n_items; ++k) {
            align_pair_result_t* res = &results[k];
            memset(res, 0, sizeof(*res));
            const bool invalid = pairedEndReads ? ((recs[2 * k].flags | recs[2 * k + 1].flags) & 3) != 0 : (recs[k].flags & 3) != 0;
            if (invalid) {
                ++n_withheld;
                res->mapped = 0; res->r1.pos = INT_MAX; res->r2.pos = INT_MAX;
            } else if (pairedEndReads) {
                const pmx_aln_record &a = recs[2 * k], &b = recs[2 * k + 1];
                if (a.mapped) { res->mapped = 1; fill(a, &res->r1); fill(b, &res->r2); }
                else { res->mapped = 0; res->r1.pos = INT_MAX; res->r2.pos = INT_MAX; }
            } else {
                const pmx_aln_record& a = recs[k];
                if (a.mapped) { res->mapped = 1; fill(a, &res->r1); }
                else res->r1.pos = INT_MAX;
            }
        }
        if (n_withheld) set_error("pmx_align_reads_direct: " + std::to_string(n_withheld) + " invalid record(s) withheld (reported unmapped)");
    } while (0);
    if (al) pmx_aligner_free(ctx, al);
    if (rs) pmx_readset_free(ctx, rs);
    return_ctx(ctx);
}

}  // extern "C"
