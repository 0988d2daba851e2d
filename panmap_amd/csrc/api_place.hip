// C ABI, device part 1: context, read sets, PLACE stage orchestration (see include/panmap_amd.h).
#include <hip/hip_runtime.h>
#include <memory>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "device/dev_util.hpp"
#include "device/pmx_math.h"
#include "host/index_build.hpp"
#include "place_kernels.h"
#include "readset.hpp"

using namespace pmx;

// --------------------------------------------------------------------------------- objects
struct pmx_place {
    // index (device, replicated per GPU)
    int64_t n_nodes = 0, n_changes = 0;
    SyncmerParams params;
    DevBuf<uint32_t> parent;
    DevBuf<uint64_t> offsets, ch_hash;
    DevBuf<int16_t> ch_par, ch_child;
    DevBuf<uint32_t> level_nodes;          // nodes sorted by (depth, DFS index) == single-thread BFS order
    std::vector<int64_t> level_off;        // host: per-level ranges into level_nodes
    std::vector<uint32_t> h_parent, h_order;
    std::vector<uint8_t> h_has_child;
    uint64_t root_beg = 0, root_end = 0;
    // seed table
    DevBuf<uint64_t> keys, keys_spare;
    DevBuf<unsigned long long> vals, vals_spare;
    uint64_t cap = 0;                      // logical table size (power of two); the buffers may be larger
    DevBuf<unsigned long long> counters;   // PMX_CTR_N
    int64_t n_reads_added = 0;
    bool table_dirty = false;              // something was inserted since the last reset
    bool needs_clear = false;              // reset happened: the slots are cleared (or the table re-made at a better size) by the next reservation
    double bases_added = 0;                // read bases (n x max_len) seeded since the last reset
    unsigned long long h_ctr[PMX_CTR_N];   // the counters as last read back; valid while nothing has been inserted since
    bool h_ctr_valid = false;
    double keys_per_base = 0;              // distinct seeds per read base of the last finished histogram (0: none yet): sizes the next table
    // distinct-read tiles of the seeding chunks in flight (k_collapse_reads -> k_seed_histogram_ks), one set per concurrent chunk
    DevBuf<uint64_t> t_words[4];
    DevBuf<uint32_t> t_amb[4], t_len[4], t_mult[4];
    DevBuf<unsigned long long> t_count;    // [4]
    bool collapse_attr_set = false;
    // finalised histogram
    DevBuf<uint64_t> hist_hash, hist_hash_tmp;
    DevBuf<int64_t> hist_count, hist_count_tmp;
    int64_t n_hist = 0;
    bool hist_sorted = false;
    DevBuf<uint8_t> dead;
    DevBuf<uint32_t> flag, pos;
    DevBuf<uint64_t> kept_hash;
    DevBuf<double> kept_log;
    int64_t n_kept = 0;
    DevBuf<uint64_t> tkeys;
    DevBuf<double> tvals;
    uint64_t tcap = 0;
    DevBuf<double> scalars;                // [0]=sum L^2 [1]=sum L [2]=wc_den
    DevBuf<double> partial;                // block sums of the canonical-order reduction
    DevBuf<unsigned long long> stats;
    DevBuf<char> tmp;                      // rocprim temp storage
    // node outputs
    hipGraphExec_t level_graph_exec = nullptr;   // captured k_score_level chain
    const void* level_graph_sig[3] = {nullptr, nullptr, nullptr};
    DevBuf<double> metrics5, scores5, scores_bfs, terms;
    DevBuf<uint32_t> chain_off, chain_nodes;   // heavy-path chains (k_score_chains), heads in BFS order
    DevBuf<uint64_t> chain_beg, chain_end;
    int64_t n_chains = 0;
    DevBuf<uint32_t> tree_done;            // k_score_tree: epoch of the call that last finished each node; [n_nodes] = status word
    uint32_t tree_epoch = 0;
    DevBuf<uint64_t> dd_h1, dd_h2, dd_h1s, dd_key;   // --dedup scratch
    DevBuf<uint32_t> dd_idx, dd_idx2;
    DevBuf<uint8_t> dd_keep;
    const pmx_readset* dd_for = nullptr;   // the read set dd_keep / dd_h1 / dd_h2 were computed for (one-shot)
    DevBuf<unsigned long long> dd_count;
    DevBuf<uint8_t> term_meta;
    DevBuf<int64_t> counts2;
    std::vector<double> h_scores;
    std::vector<std::vector<uint32_t>> tied;
};

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

// (the device memory stays at its high-water mark: a batch whose ranges want a small, then a large table would otherwise
// free and allocate gigabytes per batch -- hipFree synchronises the device, and a fresh allocation of that size takes
// hundreds of milliseconds; `cap` is the logical size, a power of two)
void table_alloc(pmx_ctx* ctx, pmx_place* pl, uint64_t cap) {
    pl->keys.ensure(cap);
    pl->vals.ensure(cap);
    pl->cap = cap;
    hipLaunchKernelGGL(k_fill_u64, dim3(grid_for((int64_t)cap, 256, ctx->n_cu * 8)), dim3(256), 0, ctx->stream, pl->keys.p, PMX_EMPTY_KEY, cap);
    PMX_HIP(hipMemsetAsync(pl->vals.p, 0, cap * sizeof(unsigned long long), ctx->stream));
}

uint64_t next_pow2(uint64_t x) {
    uint64_t p = 1;
    while (p < x) p <<= 1;
    return p;
}

// make room for `bound_new` more distinct keys at load factor <= 0.7 (never overflows afterwards)
void table_reserve(pmx_ctx* ctx, pmx_place* pl, uint64_t bound_new) {
    if (pl->needs_clear || pl->cap == 0) {   // empty table (the counters are already zero): size it for this batch, no counter round trip
        uint64_t need0 = next_pow2((uint64_t)((double)bound_new / 0.7) + 1024);
        if (need0 < (1u << 16)) need0 = 1u << 16;
        pl->needs_clear = false;
        if (pl->cap < need0 || pl->cap > 4 * need0) table_alloc(ctx, pl, need0);   // (also shrinks: clearing and compacting scan every slot)
        else {
            hipLaunchKernelGGL(k_fill_u64, dim3(grid_for((int64_t)pl->cap, 256, ctx->n_cu * 8)), dim3(256), 0, ctx->stream, pl->keys.p, PMX_EMPTY_KEY, pl->cap);
            PMX_HIP(hipMemsetAsync(pl->vals.p, 0, pl->cap * sizeof(unsigned long long), ctx->stream));
        }
        return;
    }
    unsigned long long h_ctr[PMX_CTR_N];
    PMX_HIP(hipMemcpyAsync(h_ctr, pl->counters.p, sizeof(h_ctr), hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    uint64_t entries = 0;
    for (int i = 0; i < PMX_CTR_NSHARD; ++i) entries += h_ctr[PMX_CTR_SHARD0 + i];
    uint64_t need = next_pow2((uint64_t)((double)(entries + bound_new) / 0.7) + 1024);
    if (need < (1u << 16)) need = 1u << 16;
    if (pl->cap >= need) return;
    if (pl->cap == 0 || entries == 0) {
        table_alloc(ctx, pl, need);
        return;
    }
    // the old table moves to the spare pair of buffers (kept for the next rehash), the new one takes the other pair
    DevBuf<uint64_t>& okeys = pl->keys_spare;
    DevBuf<unsigned long long>& ovals = pl->vals_spare;
    okeys.swap(pl->keys);
    ovals.swap(pl->vals);
    const uint64_t ocap = pl->cap;
    table_alloc(ctx, pl, need);
    PMX_HIP(hipMemsetAsync(pl->counters.p + PMX_CTR_SHARD0, 0, sizeof(unsigned long long) * PMX_CTR_NSHARD, ctx->stream));
    hipLaunchKernelGGL(k_table_rehash, dim3(grid_for((int64_t)ocap, 256, ctx->n_cu * 8)), dim3(256), 0, ctx->stream, okeys.p, ovals.p, ocap,
                       pl->keys.p, pl->vals.p, pl->cap - 1, pl->counters.p);
    PMX_HIP(hipGetLastError());
}

// table -> hash-sorted (hash,count) arrays
void finalize_histogram(pmx_ctx* ctx, pmx_place* pl) {
    if (pl->hist_sorted) return;
    unsigned long long* h_ctr = pl->h_ctr;
    if (!pl->h_ctr_valid) {   // (the overflow check that ends an optimistic seeding pass has read them already)
        PMX_HIP(hipMemcpyAsync(h_ctr, pl->counters.p, sizeof(pl->h_ctr), hipMemcpyDeviceToHost, ctx->stream));
        PMX_HIP(hipStreamSynchronize(ctx->stream));
        pl->h_ctr_valid = true;
    }
    if (h_ctr[PMX_CTR_OVERFLOW]) throw std::runtime_error("seed table overflow (internal sizing error)");
    int64_t n = 0;
    for (int i = 0; i < PMX_CTR_NSHARD; ++i) n += (int64_t)h_ctr[PMX_CTR_SHARD0 + i];
    pl->n_hist = n;
    if (pl->bases_added > 0 && n > 0) pl->keys_per_base = (double)n / pl->bases_added;
    pl->hist_hash.ensure(n);
    pl->hist_count.ensure(n);
    pl->hist_hash_tmp.ensure(n);
    pl->hist_count_tmp.ensure(n);
    if (n > 0) {
        PMX_HIP(hipMemsetAsync(pl->counters.p + PMX_CTR_COMPACT, 0, sizeof(unsigned long long), ctx->stream));
        hipLaunchKernelGGL(k_table_compact, dim3(grid_for((int64_t)pl->cap, 256, ctx->n_cu * 8)), dim3(256), 0, ctx->stream, pl->keys.p,
                           pl->vals.p, pl->cap, pl->hist_hash_tmp.p, pl->hist_count_tmp.p, pl->counters.p + PMX_CTR_COMPACT);
        size_t bytes = 0;
        PMX_HIP(rocprim::radix_sort_pairs(nullptr, bytes, pl->hist_hash_tmp.p, pl->hist_hash.p, pl->hist_count_tmp.p, pl->hist_count.p,
                                          (size_t)n, 0, 64, ctx->stream));
        pl->tmp.ensure(bytes);
        PMX_HIP(rocprim::radix_sort_pairs(pl->tmp.p, bytes, pl->hist_hash_tmp.p, pl->hist_hash.p, pl->hist_count_tmp.p, pl->hist_count.p,
                                          (size_t)n, 0, 64, ctx->stream));
    }
    pl->hist_sorted = true;
}

uint64_t homopolymer_hash(int code, int k) {   // src/placement.cpp:41-76
    uint64_t b = kBaseHash[code], c = kBaseHash[3 - code], f = 0, r = 0;
    for (int i = 0; i < k; ++i) { f ^= h_rol(b, (unsigned)(k - i - 1)); r ^= h_rol(c, (unsigned)(k - i - 1)); }
    return f < r ? f : r;
}

// PlacementResult::update*Score (src/placement.cpp:355-371)
struct Best {
    double best = 0.0;
    uint32_t idx = UINT32_MAX;
    std::vector<uint32_t> tied;
    void update(uint32_t node, double score) {
        double tol = std::max(best * 0.0001, 1e-9);
        if (score > best + tol) {
            best = score;
            idx = node;
            tied.clear();
            tied.push_back(node);
        } else if (score >= best - tol && score > 0) {
            if (tied.empty() || tied.back() != idx) tied.push_back(idx);
            if (node != idx) tied.push_back(node);
        }
    }
};

}  // namespace

namespace pmx {
// keys + stable radix sort of the reads [r0, r1): their slice of loc_perm then holds them (absolute indices) in locality order
static void order_range(pmx_ctx* ctx, const pmx_readset* rs, int64_t r0, int64_t r1) {
    const int64_t n = rs->n, m = r1 - r0;
    rs->loc_key.ensure((size_t)n); rs->loc_key2.ensure((size_t)n); rs->loc_idx.ensure((size_t)n); rs->loc_perm.ensure((size_t)n);
    if (m <= 0) return;
    hipLaunchKernelGGL(k_read_prefix_keys, dim3(grid_for(m, 256, ctx->n_cu * 8)), dim3(256), 0, ctx->stream, rs->words.p, rs->woff.p, r0, r1, rs->loc_key.p,
                       rs->loc_idx.p);
    // (sorting on the key's top 20 bits alone -- three radix passes instead of four -- returned index arrays that were no
    //  permutation on this rocprim: measured, not understood; the full key it is)
    const unsigned lo_bit = 0u;
    size_t bytes = 0;
    PMX_HIP(rocprim::radix_sort_pairs(nullptr, bytes, rs->loc_key.p + r0, rs->loc_key2.p + r0, rs->loc_idx.p + r0, rs->loc_perm.p + r0, (size_t)m, lo_bit, 32, ctx->stream));
    rs->loc_tmp.ensure(bytes);
    PMX_HIP(rocprim::radix_sort_pairs(rs->loc_tmp.p, bytes, rs->loc_key.p + r0, rs->loc_key2.p + r0, rs->loc_idx.p + r0, rs->loc_perm.p + r0, (size_t)m, lo_bit, 32, ctx->stream));
}

const uint32_t* readset_locality_order(pmx_ctx* ctx, const pmx_readset* rs) {
    const int64_t n = rs->n;
    if (!rs->packed || n < 4096 || n >= (int64_t)UINT32_MAX) return nullptr;
    if (rs->has_order) return rs->loc_perm.p;
    order_range(ctx, rs, 0, n);
    rs->ordered_ranges.clear();
    rs->ordered_ranges.add(0, n);
    rs->has_order = true;
    return rs->loc_perm.p;
}

const uint32_t* readset_locality_order_range(pmx_ctx* ctx, const pmx_readset* rs, int64_t r0, int64_t r1) {
    const int64_t n = rs->n;
    if (n < 4096 || n >= (int64_t)UINT32_MAX || r1 - r0 < 4096) return nullptr;
    if (rs->has_order || rs->ordered_ranges.covers(r0, r1)) return rs->loc_perm.p;
    if (!rs->packed && !rs->packed_ranges.covers(r0, r1)) return nullptr;
    // the larger buffers may be re-made by ensure(): only before the first range of a set
    if (rs->ordered_ranges.iv.empty()) { rs->loc_key.ensure((size_t)n); rs->loc_key2.ensure((size_t)n); rs->loc_idx.ensure((size_t)n); rs->loc_perm.ensure((size_t)n); }
    order_range(ctx, rs, r0, r1);
    rs->ordered_ranges.add(r0, r1);
    if (rs->ordered_ranges.covers(0, n)) rs->has_order = true;
    return rs->loc_perm.p;
}
}  // namespace pmx

extern "C" {

// ---------------------------------------------------------------------------------- context
int pmx_ctx_create(int device_ordinal, pmx_ctx** out) {
    if (!out) return PMX_ERR_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(PMX_ERR_NO_DEVICE, "no HIP device available (the GPU path has no CPU fallback)");
    if (device_ordinal < 0 || device_ordinal >= n) return fail(PMX_ERR_ARG, "device ordinal out of range");
    PMX_TRY
    PMX_HIP(hipSetDevice(device_ordinal));
    pmx_ctx* c = new pmx_ctx();
    c->device = device_ordinal;
    hipDeviceProp_t prop;
    PMX_HIP(hipGetDeviceProperties(&prop, device_ordinal));
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c->stream = create_dedicated_stream(c->n_cu);   // a hardware queue of its own: contexts overlap their kernels
    *out = c;
    return PMX_OK;
    PMX_CATCH
}

void pmx_ctx_destroy(pmx_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (auto& kv : ctx->timers) {
        if (kv.second.e0) (void)hipEventDestroy(kv.second.e0);
        if (kv.second.e1) (void)hipEventDestroy(kv.second.e1);
    }
    for (int j = 0; j < 3; ++j) {
        if (ctx->seed_streams[j]) (void)hipStreamDestroy(ctx->seed_streams[j]);
        if (ctx->seed_done[j]) (void)hipEventDestroy(ctx->seed_done[j]);
    }
    if (ctx->seed_go) (void)hipEventDestroy(ctx->seed_go);
    if (ctx->tail_go) (void)hipEventDestroy(ctx->tail_go);
    if (ctx->tail_done) (void)hipEventDestroy(ctx->tail_done);
    if (ctx->pair_stream) (void)hipStreamDestroy(ctx->pair_stream);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int pmx_ctx_synchronize(pmx_ctx* ctx) {
    if (!ctx) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
    PMX_CATCH
}

void* pmx_ctx_stream(pmx_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

double pmx_last_kernel_ms(pmx_ctx* ctx, const char* name) {
    if (!ctx || !name) return -1.0;
    auto it = ctx->timers.find(name);
    if (it == ctx->timers.end() || !it->second.pending) return -1.0;
    float ms = 0.f;
    if (hipEventSynchronize(it->second.e1) != hipSuccess) return -1.0;
    if (hipEventElapsedTime(&ms, it->second.e0, it->second.e1) != hipSuccess) return -1.0;
    return it->second.launches > 0 ? (double)ms / it->second.launches : (double)ms;
}

// --------------------------------------------------------------------------------- read sets
static int readset_finish(pmx_ctx* ctx, pmx_readset* rs, const int64_t* h_off) {
    // every read starts on a 32-base word
    std::vector<int64_t> woff((size_t)rs->n + 1);
    int64_t w = 0, maxlen = 0, total = 0;
    for (int64_t i = 0; i < rs->n; ++i) {
        woff[i] = w;
        int64_t len = h_off[i + 1] - h_off[i];
        if (len < 0) throw std::runtime_error("read offsets are not monotone");
        w += (len + 31) / 32;
        maxlen = std::max(maxlen, len);
        total += len;
    }
    woff[rs->n] = w;
    rs->n_words = w;
    rs->max_len = maxlen;
    rs->total = total;
    rs->woff.alloc((size_t)rs->n + 1);
    PMX_HIP(hipMemcpyAsync(rs->woff.p, woff.data(), sizeof(int64_t) * ((size_t)rs->n + 1), hipMemcpyHostToDevice, ctx->stream));
    rs->words.alloc((size_t)w);
    rs->amb.alloc((size_t)w);
    rs->has_recs = rs->n > 0 && maxlen <= 160;
    if (rs->has_recs) rs->recs.alloc((size_t)rs->n * 64);
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
}

int pmx_readset_upload(pmx_ctx* ctx, const char* concat, const int64_t* offsets, int64_t n_reads, pmx_readset** out) {
    if (!ctx || !offsets || !out || n_reads < 0 || (!concat && n_reads > 0 && offsets[n_reads] > 0)) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<pmx_readset> rs(new pmx_readset());
    rs->n = n_reads;
    const int64_t total = offsets[n_reads] - offsets[0];
    rs->ascii.alloc((size_t)total + 32);
    rs->off.alloc((size_t)n_reads + 1);
    if (total > 0) PMX_HIP(hipMemcpyAsync(rs->ascii.p, concat + offsets[0], (size_t)total, hipMemcpyHostToDevice, ctx->stream));
    std::vector<int64_t> rel((size_t)n_reads + 1);
    for (int64_t i = 0; i <= n_reads; ++i) rel[i] = offsets[i] - offsets[0];
    PMX_HIP(hipMemcpyAsync(rs->off.p, rel.data(), sizeof(int64_t) * ((size_t)n_reads + 1), hipMemcpyHostToDevice, ctx->stream));
    readset_finish(ctx, rs.get(), rel.data());
    *out = rs.release();
    return PMX_OK;
    PMX_CATCH
}

int pmx_readset_wrap_device(pmx_ctx* ctx, const void* d_concat, const void* d_offsets, int64_t n_reads, int64_t total_bytes,
                            int64_t max_read_len, pmx_readset** out) {
    if (!ctx || !d_concat || !d_offsets || !out || n_reads < 0 || total_bytes < 0) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<pmx_readset> rs(new pmx_readset());
    rs->n = n_reads;
    rs->ascii.wrap((uint8_t*)d_concat, (size_t)total_bytes);
    rs->off.wrap((int64_t*)d_offsets, (size_t)n_reads + 1);
    std::vector<int64_t> h_off((size_t)n_reads + 1);
    PMX_HIP(hipMemcpy(h_off.data(), d_offsets, sizeof(int64_t) * ((size_t)n_reads + 1), hipMemcpyDeviceToHost));
    // the kernels trust these offsets: they must lie inside the wrapped buffer, in order.  (offsets need not start at 0:
    // a slice [r0, r1] of a larger offsets array addresses its reads in the same buffer.)
    if (h_off[0] < 0 || h_off[n_reads] > total_bytes) return fail(PMX_ERR_ARG, "device offsets run outside the wrapped buffer");
    for (int64_t i = 0; i < n_reads; ++i)
        if (h_off[i + 1] < h_off[i]) return fail(PMX_ERR_ARG, "device offsets are not monotone");
    if (max_read_len > 0)
        for (int64_t i = 0; i < n_reads; ++i)
            if (h_off[i + 1] - h_off[i] > max_read_len) return fail(PMX_ERR_ARG, "a read is longer than max_read_len");
    rs->off0 = h_off[0];
    readset_finish(ctx, rs.get(), h_off.data());
    *out = rs.release();
    return PMX_OK;
    PMX_CATCH
}

// A read set object re-pointed at another batch in device memory: the word offsets are computed ON THE DEVICE (count
// kernel + exclusive scan; offsets checked there too) and the packed-read buffers of the object are reused when they are
// large enough -- the streaming form of pmx_readset_wrap_device (no host pass over the offsets, no hipMalloc per batch).
// One 24-byte read-back per call.
int pmx_readset_rewrap_device(pmx_ctx* ctx, pmx_readset* rs, const void* d_concat, const void* d_offsets, int64_t n_reads, int64_t total_bytes,
                              int64_t max_read_len) {
    if (!ctx || !rs || !d_concat || !d_offsets || n_reads < 0 || total_bytes < 0) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    rs->packed = false;
    rs->has_order = false;
    rs->has_pair_order = false;
    rs->packed_ranges.clear();
    rs->ordered_ranges.clear();
    rs->has_qual = false;
    rs->n = n_reads;
    rs->ascii.wrap((uint8_t*)d_concat, (size_t)total_bytes);
    rs->off.wrap((int64_t*)d_offsets, (size_t)n_reads + 1);
    rs->nw_tmp.ensure((size_t)n_reads + 1);
    rs->woff.ensure((size_t)n_reads + 1);
    rs->stats.ensure(2);
    PMX_HIP(hipMemsetAsync(rs->stats.p, 0, 2 * sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(k_read_word_counts, dim3(grid_for(n_reads + 1, 256, ctx->n_cu * 8)), dim3(256), 0, ctx->stream, rs->off.p, n_reads, total_bytes,
                       rs->nw_tmp.p, rs->stats.p);
    size_t bytes = 0;
    PMX_HIP(rocprim::exclusive_scan(nullptr, bytes, rs->nw_tmp.p, rs->woff.p, (int64_t)0, (size_t)n_reads + 1, rocprim::plus<int64_t>(), ctx->stream));
    rs->scan_tmp.ensure(bytes);
    PMX_HIP(rocprim::exclusive_scan(rs->scan_tmp.p, bytes, rs->nw_tmp.p, rs->woff.p, (int64_t)0, (size_t)n_reads + 1, rocprim::plus<int64_t>(), ctx->stream));
    struct { int64_t n_words; unsigned long long st[2]; } h = {0, {0, 0}};
    PMX_HIP(hipMemcpyAsync(&h.n_words, rs->woff.p + n_reads, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipMemcpyAsync(h.st, rs->stats.p, sizeof(h.st), hipMemcpyDeviceToHost, ctx->stream));
    int64_t first = 0, last = 0;
    PMX_HIP(hipMemcpyAsync(&first, rs->off.p, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipMemcpyAsync(&last, rs->off.p + n_reads, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    if (h.st[1]) return fail(PMX_ERR_ARG, "device offsets are not monotone or run outside the wrapped buffer");
    if (max_read_len > 0 && (int64_t)h.st[0] > max_read_len) return fail(PMX_ERR_ARG, "a read is longer than max_read_len");
    rs->n_words = h.n_words;
    rs->max_len = (int64_t)h.st[0];
    rs->total = last - first;
    rs->off0 = first;
    rs->words.ensure((size_t)std::max<int64_t>(h.n_words, 1));
    rs->amb.ensure((size_t)std::max<int64_t>(h.n_words, 1));
    rs->has_recs = n_reads > 0 && rs->max_len <= 160;
    if (rs->has_recs) rs->recs.ensure((size_t)n_reads * 64);
    return PMX_OK;
    PMX_CATCH
}

int pmx_readset_set_qualities(pmx_ctx* ctx, pmx_readset* rs, const char* qual_concat) {
    if (!ctx || !rs || (rs->total > 0 && !qual_concat)) return PMX_ERR_ARG;
    if (rs->off0 != 0) return fail(PMX_ERR_ARG, "qualities need a read set whose offsets start at 0");
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    rs->qual.ensure((size_t)rs->total + 32);
    if (rs->total > 0) PMX_HIP(hipMemcpyAsync(rs->qual.p, qual_concat, (size_t)rs->total, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    rs->has_qual = true;
    return PMX_OK;
    PMX_CATCH
}

int pmx_readset_pack(pmx_ctx* ctx, pmx_readset* rs) {
    if (!ctx || !rs) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    timer_begin(ctx, "pack");
    // every read has max_len bases (total = n x max_len): the word -> read mapping is a division (k_pack_reads_fixed)
    const bool fixed = rs->n > 0 && rs->max_len > 0 && rs->total == rs->n * rs->max_len;
    uint8_t* recs = rs->has_recs ? rs->recs.p : nullptr;
    // (ragged sets: a read without a word gets no record written -- cleared first, so that it reads as length 0)
    if (recs && !fixed) PMX_HIP(hipMemsetAsync(recs, 0, (size_t)rs->n * 64, ctx->stream));
    if (rs->n_words > 0 && fixed)
        hipLaunchKernelGGL(k_pack_reads_fixed, dim3(grid_for(rs->n_words, 256, 1 << 30)), dim3(256), 0, ctx->stream, rs->ascii.p, rs->off0, (int)rs->max_len,
                           (int64_t)0, rs->n_words, rs->words.p, rs->amb.p, recs);
    else if (rs->n_words > 0)
        hipLaunchKernelGGL(k_pack_reads, dim3(grid_for(rs->n_words, 256, ctx->n_cu * 16)), dim3(256), 0, ctx->stream, rs->ascii.p, rs->off.p,
                           rs->woff.p, rs->n, rs->n_words, rs->words.p, rs->amb.p, (int64_t)0, (int64_t)-1, recs);
    PMX_HIP(hipGetLastError());
    timer_end(ctx, "pack", 1);
    rs->packed = true;
    rs->has_order = false;   // (the buffer behind a wrapped read set may hold new reads)
    rs->has_pair_order = false;
    rs->packed_ranges.clear();
    rs->packed_ranges.add(0, rs->n);
    rs->ordered_ranges.clear();
    return PMX_OK;
    PMX_CATCH
}

// streaming: the reads [r0, r1) alone (their bases have landed; the offsets of the whole set are in place since the
// (re)wrap).  The word range comes from the device-side word offsets: no host round trip.
int pmx_readset_pack_range(pmx_ctx* ctx, pmx_readset* rs, int64_t r0, int64_t r1) {
    if (!ctx || !rs || r0 < 0 || r1 < r0 || r1 > rs->n) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    if (r1 > r0 && rs->n_words > 0) {
        // grid sized by the range's share of the words (exact for reads of one length; any grid is correct: the kernel strides)
        const int64_t est = (int64_t)((double)rs->n_words * (double)(r1 - r0) / (double)std::max<int64_t>(rs->n, 1)) + 1;
        // (a read without a word gets no record written: the range's records are cleared first)
        if (rs->has_recs) PMX_HIP(hipMemsetAsync(rs->recs.p + (size_t)r0 * 64, 0, (size_t)(r1 - r0) * 64, ctx->stream));
        hipLaunchKernelGGL(k_pack_reads, dim3(grid_for(est, 256, ctx->n_cu * 16)), dim3(256), 0, ctx->stream, rs->ascii.p, rs->off.p,
                           rs->woff.p, rs->n, rs->n_words, rs->words.p, rs->amb.p, r0, r1, rs->has_recs ? rs->recs.p : nullptr);
        PMX_HIP(hipGetLastError());
    }
    if (rs->packed) {   // re-packing part of a packed set: the order of those reads may have changed
        rs->has_order = false;
        rs->has_pair_order = false;
        rs->ordered_ranges.clear();
    }
    rs->packed_ranges.add(r0, r1);
    if (rs->packed_ranges.covers(0, rs->n)) rs->packed = true;
    return PMX_OK;
    PMX_CATCH
}

void pmx_readset_free(pmx_ctx* ctx, pmx_readset* rs) {
    if (ctx) (void)hipSetDevice(ctx->device);
    delete rs;
}
int64_t pmx_readset_num_reads(const pmx_readset* rs) { return rs ? rs->n : 0; }

// ------------------------------------------------------------------------------------- place
int pmx_place_create(pmx_ctx* ctx, const pmx_index* idx, pmx_place** out) {
    if (!ctx || !idx || !out) return PMX_ERR_ARG;
    const LiteIndex* L = pmx_index_internal(idx);
    if (L->params.k > 32 || L->params.k < 1) return fail(PMX_ERR_UNSUPPORTED, "device seeding supports 1 <= k <= 32");
    if (L->params.l > 64) return fail(PMX_ERR_UNSUPPORTED, "device seeding supports l <= 64");
    if (L->hpc) return fail(PMX_ERR_UNSUPPORTED, "HPC indexes are not supported on the device yet");
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    pmx_place* pl = new pmx_place();
    const int64_t n = (int64_t)L->parent.size(), m = (int64_t)L->hash.size();
    pl->n_nodes = n;
    pl->n_changes = m;
    pl->params = L->params;
    pl->parent.alloc(n); pl->offsets.alloc(n + 1); pl->ch_hash.alloc(m); pl->ch_par.alloc(m); pl->ch_child.alloc(m);
    PMX_HIP(hipMemcpyAsync(pl->parent.p, L->parent.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(pl->offsets.p, L->offsets.data(), sizeof(uint64_t) * (n + 1), hipMemcpyHostToDevice, ctx->stream));
    if (m > 0) {
        PMX_HIP(hipMemcpyAsync(pl->ch_hash.p, L->hash.data(), sizeof(uint64_t) * m, hipMemcpyHostToDevice, ctx->stream));
        PMX_HIP(hipMemcpyAsync(pl->ch_par.p, L->parent_count.data(), sizeof(int16_t) * m, hipMemcpyHostToDevice, ctx->stream));
        PMX_HIP(hipMemcpyAsync(pl->ch_child.p, L->child_count.data(), sizeof(int16_t) * m, hipMemcpyHostToDevice, ctx->stream));
    }
    pl->root_beg = L->offsets[0];
    pl->root_end = L->offsets[1];
    // BFS levels: visit order of the single-threaded traversal = ascending (depth, DFS index)
    pl->h_parent = L->parent;
    std::vector<int32_t> depth(n);
    pl->h_has_child.assign(n, 0);
    int32_t maxd = 0;
    for (int64_t i = 0; i < n; ++i) {
        depth[i] = i == 0 ? 0 : depth[L->parent[i]] + 1;
        if (i > 0) pl->h_has_child[L->parent[i]] = 1;
        maxd = std::max(maxd, depth[i]);
    }
    pl->level_off.assign((size_t)maxd + 2, 0);
    for (int64_t i = 0; i < n; ++i) ++pl->level_off[depth[i] + 1];
    for (int32_t d = 0; d <= maxd; ++d) pl->level_off[d + 1] += pl->level_off[d];
    pl->h_order.resize(n);
    {
        std::vector<int64_t> fill(pl->level_off.begin(), pl->level_off.end());
        for (int64_t i = 0; i < n; ++i) pl->h_order[fill[depth[i]]++] = (uint32_t)i;
    }
    pl->level_nodes.alloc(n);
    PMX_HIP(hipMemcpyAsync(pl->level_nodes.p, pl->h_order.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, ctx->stream));
    // heavy-path chains for k_score_chains: weight of a node = its change count (the serial adds it costs) plus a
    // constant; a chain = head (the root or a non-heaviest child), then heaviest child after heaviest child
    std::vector<uint32_t> h_chain_off, h_chain_nodes;
    std::vector<uint64_t> h_chain_beg, h_chain_end;
    {
        std::vector<uint64_t> sw((size_t)n);
        std::vector<uint32_t> heavy((size_t)n, UINT32_MAX), n_child((size_t)n, 0);
        for (int64_t i = 0; i < n; ++i) sw[i] = (L->offsets[i + 1] - L->offsets[i]) + 64;
        for (int64_t i = n - 1; i > 0; --i) sw[L->parent[i]] += sw[i];   // parents precede their children
        for (int64_t i = 1; i < n; ++i) {
            const uint32_t p = L->parent[i];
            ++n_child[p];
            if (heavy[p] == UINT32_MAX || sw[i] > sw[heavy[p]]) heavy[p] = (uint32_t)i;
        }
        h_chain_nodes.reserve((size_t)n); h_chain_beg.reserve((size_t)n); h_chain_end.reserve((size_t)n);
        for (int64_t j = 0; j < n; ++j) {
            const uint32_t hd = pl->h_order[j];
            if (hd != 0 && heavy[L->parent[hd]] == hd) continue;   // inside its parent's chain
            h_chain_off.push_back((uint32_t)h_chain_nodes.size());
            for (uint32_t v = hd; v != UINT32_MAX; v = heavy[v]) {
                h_chain_nodes.push_back(v | (n_child[v] >= 2 ? 0x80000000u : 0u));
                h_chain_beg.push_back(L->offsets[v]);
                h_chain_end.push_back(L->offsets[v + 1]);
            }
        }
        h_chain_off.push_back((uint32_t)h_chain_nodes.size());
        if ((int64_t)h_chain_nodes.size() != n) throw std::runtime_error("chain decomposition does not cover the tree");
    }
    pl->n_chains = (int64_t)h_chain_off.size() - 1;
    pl->chain_off.alloc(h_chain_off.size()); pl->chain_nodes.alloc((size_t)n); pl->chain_beg.alloc((size_t)n); pl->chain_end.alloc((size_t)n);
    PMX_HIP(hipMemcpyAsync(pl->chain_off.p, h_chain_off.data(), sizeof(uint32_t) * h_chain_off.size(), hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(pl->chain_nodes.p, h_chain_nodes.data(), sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(pl->chain_beg.p, h_chain_beg.data(), sizeof(uint64_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(pl->chain_end.p, h_chain_end.data(), sizeof(uint64_t) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    pl->counters.alloc(PMX_CTR_N);
    PMX_HIP(hipMemsetAsync(pl->counters.p, 0, sizeof(unsigned long long) * PMX_CTR_N, ctx->stream));
    pl->scalars.alloc(8);
    pl->stats.alloc(8);
    pl->metrics5.alloc(5 * (size_t)n);
    pl->scores5.alloc(5 * (size_t)n);
    pl->scores_bfs.alloc(5 * (size_t)n);
    pl->counts2.alloc(2 * (size_t)n);
    pl->tied.assign(5, {});
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    *out = pl;
    return PMX_OK;
    PMX_CATCH
}

void pmx_place_free(pmx_ctx* ctx, pmx_place* pl) {
    if (ctx) (void)hipSetDevice(ctx->device);
    if (pl && pl->level_graph_exec) (void)hipGraphExecDestroy(pl->level_graph_exec);
    delete pl;
}

int pmx_place_reset(pmx_ctx* ctx, pmx_place* pl) {
    if (!ctx || !pl) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    PMX_HIP(hipMemsetAsync(pl->counters.p, 0, sizeof(unsigned long long) * PMX_CTR_N, ctx->stream));
    pl->h_ctr_valid = false;
    pl->needs_clear = pl->cap != 0;   // the slots are cleared by the next reservation, which may also pick a better size
    pl->bases_added = 0;
    pl->n_reads_added = 0;
    pl->hist_sorted = false;
    pl->n_hist = 0;
    pl->table_dirty = false;
    return PMX_OK;
    PMX_CATCH
}

// keep[read] = 1 for the first copy of every distinct read string of the set: two independent 64-bit hashes of the raw
// ASCII, a stable radix sort on the 128-bit key (by h2, then by h1), an exact byte comparison with the predecessor
static void dedup_local(pmx_ctx* ctx, pmx_place* pl, const pmx_readset* rs) {
    const int64_t n = rs->n;
    const int G = ctx->n_cu * 8;
    pl->dd_h1.ensure((size_t)n); pl->dd_h2.ensure((size_t)n); pl->dd_h1s.ensure((size_t)n); pl->dd_key.ensure((size_t)n);
    pl->dd_idx.ensure((size_t)n); pl->dd_idx2.ensure((size_t)n); pl->dd_keep.ensure((size_t)n);
    if (n == 0) { pl->dd_for = rs; return; }
    hipLaunchKernelGGL(k_read_hashes, dim3(grid_for(n, 256, G)), dim3(256), 0, ctx->stream, rs->ascii.p, rs->off.p, n, pl->dd_h1.p, pl->dd_h2.p, pl->dd_idx.p);
    size_t bytes = 0;
    PMX_HIP(rocprim::radix_sort_pairs(nullptr, bytes, pl->dd_h2.p, pl->dd_key.p, pl->dd_idx.p, pl->dd_idx2.p, (size_t)n, 0, 64, ctx->stream));
    pl->tmp.ensure(bytes);
    PMX_HIP(rocprim::radix_sort_pairs(pl->tmp.p, bytes, pl->dd_h2.p, pl->dd_key.p, pl->dd_idx.p, pl->dd_idx2.p, (size_t)n, 0, 64, ctx->stream));
    hipLaunchKernelGGL(k_gather_u64, dim3(grid_for(n, 256, G)), dim3(256), 0, ctx->stream, pl->dd_h1.p, pl->dd_idx2.p, n, pl->dd_key.p);
    PMX_HIP(rocprim::radix_sort_pairs(nullptr, bytes, pl->dd_key.p, pl->dd_h1s.p, pl->dd_idx2.p, pl->dd_idx.p, (size_t)n, 0, 64, ctx->stream));
    pl->tmp.ensure(bytes);
    PMX_HIP(rocprim::radix_sort_pairs(pl->tmp.p, bytes, pl->dd_key.p, pl->dd_h1s.p, pl->dd_idx2.p, pl->dd_idx.p, (size_t)n, 0, 64, ctx->stream));
    hipLaunchKernelGGL(k_mark_first_of_run, dim3(grid_for(n, 256, G)), dim3(256), 0, ctx->stream, rs->ascii.p, rs->off.p, pl->dd_h1s.p, pl->dd_h2.p, pl->dd_idx.p, n,
                       pl->dd_keep.p);
    PMX_HIP(hipGetLastError());
    pl->dd_for = rs;
}

static int add_reads_impl(pmx_ctx* ctx, pmx_place* pl, const pmx_readset* rs, int64_t rr0, int64_t rr1, const pmx_place_params* pp) {
    if (!ctx || !pl || !rs || !pp || rr0 < 0 || rr1 < rr0 || rr1 > rs->n) return PMX_ERR_ARG;
    const bool whole = rr0 == 0 && rr1 == rs->n;
    if (!rs->packed && !rs->packed_ranges.covers(rr0, rr1)) return fail(PMX_ERR_ARG, "read set is not packed (call pmx_readset_pack / pmx_readset_pack_range first)");
    if (!whole && pp->dedup_reads) return fail(PMX_ERR_UNSUPPORTED, "--dedup collapses duplicates over a whole read set: seed it with pmx_place_add_reads");
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    pl->h_ctr_valid = false;
    SeedParams sp;
    sp.k = pl->params.k; sp.s = pl->params.s; sp.t = pl->params.t; sp.l = pl->params.l; sp.open = pl->params.open ? 1 : 0;
    sp.trim_start = pp->trim_start; sp.trim_end = pp->trim_end;
    const int w = sp.k - sp.s + 1, l = sp.l < 1 ? 1 : sp.l;
    const size_t lds = (size_t)(2 * w + l) * PMX_SEED_BLOCK * sizeof(uint64_t) +
                       (size_t)(PMX_SEED_BLOCK / 64) * (PMX_SEED_QCAP * sizeof(uint64_t) + 8);   // + the waves' seed queues and their counters
    if (lds > 160 * 1024) return fail(PMX_ERR_UNSUPPORTED, "k-s+1 too large for the LDS ring");
    if (lds > 64 * 1024)
        PMX_HIP(hipFuncSetAttribute((const void*)k_seed_histogram, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (rr1 > rr0) {
        // Reads go in chunks of <= ~64M bases, three at a time.  Before each group the table is grown (rehash) if the distinct
        // keys seen so far plus one new key per base of the chunk would push the load factor past 0.7, so an
        // insert can never fail, yet the table is sized by what the reads actually contain (a few million
        // distinct seeds for a 1M-read sample) instead of by the one-key-per-base bound of the whole batch:
        // a 16x smaller table to clear, probe and compact.
        const bool quality_mode = pp->min_seed_quality > 0 && rs->has_qual;   // src/placement.cpp:1386: no dedup in this branch
        const uint8_t* keep = nullptr;
        if (pp->dedup_reads && !quality_mode) {   // --dedup: every distinct read sequence of this read set counts once
            if (pl->dd_for != rs) dedup_local(ctx, pl, rs);   // (pmx_place_dedup_* may have prepared -- and thinned -- the mask already)
            keep = pl->dd_keep.p;
            pl->dd_for = nullptr;                               // one use: the next call starts over
        }
        // (a group of three chunks = one table reservation: 1M x 150 bp is one group; 16 MB chunks measured 2.46 ms for the stage,
        //  one group 2.09).  A large range is cut into as few groups as 512 MB chunks allow: 10M x 150 bp in 24 launches of 64 MB
        //  took 10.8 ms, in 16 of 96 MB 9.6 ms, in 4 of 384 MB 9.05 ms -- a third of the range per chunk, between 64 and 512 MB
        int64_t chunk_mb = std::min<int64_t>(512, std::max<int64_t>(64, (((rr1 - rr0) * std::max<int64_t>(rs->max_len, 1) / 3) >> 20) + 1));
        if (const char* e = pmx::opt_str(pmx::O_SEED_CHUNK_MB)) chunk_mb = std::max<int64_t>(1, atoll(e));
        const int64_t chunk_reads_opt = std::max<int64_t>(1, (chunk_mb << 20) / std::max<int64_t>(rs->max_len, 1));
        // (with the safe bound -- one key per base of the group -- the groups stay three chunks of 64 MB: the table is grown by
        //  what a group can add, and a 1.5 GB group would reserve a 68 GB table for a sample that overflowed the optimistic one)
        const int64_t chunk_reads_safe = std::max<int64_t>(1, (std::min<int64_t>(chunk_mb, 64) << 20) / std::max<int64_t>(rs->max_len, 1));
        // Table sizing.  The safe bound on the distinct keys a chunk can add is one per base; real reads add one seed per
        // 5-6 bases and most of those repeat.  When the table is empty at the start of the call the chunks are first run
        // with an eighth of the safe bound (a smaller table to clear, probe and compact); an insert that finds no slot is
        // counted (probe sequences are capped), and in that case the table is cleared and the call is redone with the safe
        // bound.  Same histogram either way.
        // Once a histogram has been finished its density (distinct seeds per read base) sizes the next optimistic table:
        // twice that, at least 1/256 of the safe bound (1/64 until round 4: 2^25 slots at a load of 3.5 % for 10M reads; every table atomic missed the L2 and the compaction scanned 512 MB) -- batches of one run look alike, and a table 16x smaller is 16x
        // cheaper to clear and to compact.
        int64_t bound_div = 1;
        double bound_frac = 0;   // > 0: optimistic bound as a fraction of the safe one (takes the place of 1 / bound_div)
        if (!pl->table_dirty && !pmx::opt_str(pmx::O_SEED_SAFE_BOUND)) {
            bound_div = 8;
            if (const char* e = pmx::opt_str(pmx::O_SEED_BOUND_DIV)) bound_div = std::max<int64_t>(1, atoll(e));   // (tests force the redo with a large value)
            else if (pl->keys_per_base > 0 && !pmx::opt_str(pmx::O_SEED_NO_HINT)) bound_frac = std::min(1.0 / 8, std::max(2 * pl->keys_per_base, 1.0 / 256));
        }
        timer_begin(ctx, "seed");
        // seeding order (default-parameter kernel): reads that start with the same 16 bases next to each other, so that a
        // block's (seed, count) cache sees its seeds many times (k_seed_histogram_ks)
        const bool ks_path = sp.k == 19 && sp.s == 8 && sp.t == 0 && (l == 3 || l == 1) && !quality_mode && !pmx::opt_str(pmx::O_SEED_GENERIC);
        const uint32_t* perm = ks_path && !pmx::opt_str(pmx::O_SEED_NO_SORT) ? (whole ? readset_locality_order(ctx, rs) : readset_locality_order_range(ctx, rs, rr0, rr1)) : nullptr;
        // the specialised kernel keeps its rings in registers: LDS = the waves' seed queues + the block cache (keys 8 B +
        // counts 4 B + admission tags 2 B per entry)
        const size_t lds_ks = (size_t)(PMX_SEED_BLOCK / 64) * PMX_SEED_QCAP_KS * sizeof(uint64_t) + (size_t)PMX_SEED_CACHE * 14 + 35 * sizeof(uint64_t) +   // + the base-hash tables
                              (size_t)(PMX_SEED_BLOCK / 64) * (64 * sizeof(uint32_t) + PMX_SEED_QCAP_KS);                                                   // + multiplicities, pushing lanes
        // Read collapse ahead of the seeding kernel (k_collapse_reads: src/placement.cpp:1550-1593 seeds every distinct read once,
        // with its multiplicity): reads of up to 160 bases on the specialised kernel's path
        const bool collapse = ks_path && rs->max_len <= 160 && !pmx::opt_str(pmx::O_SEED_NO_COLLAPSE);
        if (collapse && !pl->collapse_attr_set) {
            PMX_HIP(hipFuncSetAttribute((const void*)k_collapse_reads, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PMX_DEDUP_LDS_BYTES));
            pl->collapse_attr_set = true;
        }
        if (collapse) pl->t_count.ensure(4);
        // every read of the set has max_len bases (total = n x max_len): word offsets follow from the read index
        const int fixed_len = (rs->n > 0 && rs->total == rs->n * rs->max_len) ? (int)rs->max_len : 0;
        for (int attempt = 0; attempt < 2; ++attempt) {
        // A launch of one chunk is latency-bound (every wave walks its 150 bases one after the other, a few waves per
        // SIMD): the chunks of a group run concurrently on side streams, sharing the table (all they do is atomics).
        // (round 4, after the read collapse: one chunk at a time is as fast -- 10M reads 4.8 ms either way, 1.25M 1.42 against
        //  1.53 -- so ranges under 4M reads stay on the context's own stream: every side stream is a hardware queue, and a host that
        //  keeps four batches of a small job in flight, bench.py up to 3M reads, ran into the queue limit with them: its
        //  one-batch-at-a-time leg fell from 102 to 57-67 M reads/s)
        int n_par = rr1 - rr0 >= 4000000 ? 3 : 1;
        if (const char* e = pmx::opt_str(pmx::O_SEED_PAR)) n_par = std::max(1, std::min(4, atoi(e)));
        if (n_par > 1 && !ctx->seed_go) {
            PMX_HIP(hipEventCreateWithFlags(&ctx->seed_go, hipEventDisableTiming));
            for (int j = 0; j < 3; ++j) {   // (own hardware queues: see create_dedicated_stream)
                ctx->seed_streams[j] = create_dedicated_stream(ctx->n_cu);
                PMX_HIP(hipEventCreateWithFlags(&ctx->seed_done[j], hipEventDisableTiming));
            }
        }
        const int64_t chunk_reads = bound_div > 1 ? chunk_reads_opt : chunk_reads_safe;
        for (int64_t g0 = rr0; g0 < rr1; g0 += chunk_reads * n_par) {
            const int64_t g1 = std::min<int64_t>(rr1, g0 + chunk_reads * n_par);
            const double safe_bound = (double)(g1 - g0) * (double)rs->max_len;
            table_reserve(ctx, pl, (uint64_t)(bound_div > 1 && bound_frac > 0 ? safe_bound * bound_frac : safe_bound / (double)bound_div) + 1);
            if (n_par > 1) PMX_HIP(hipEventRecord(ctx->seed_go, ctx->stream));
            int j = 0;
            for (int64_t r0 = g0; r0 < g1; r0 += chunk_reads, ++j) {
            const int64_t r1 = std::min<int64_t>(g1, r0 + chunk_reads);
            hipStream_t st = j == 0 ? ctx->stream : ctx->seed_streams[j - 1];
            if (j > 0) PMX_HIP(hipStreamWaitEvent(st, ctx->seed_go, 0));
            // batches of PMX_SEED_BLOCK reads per block of the specialised kernel, contiguous in the seeding order.  More than one
            // saves cache flushes (memory-side atomics) but measured slower: 1.55 ms for the stage with 1, 1.59 / 1.84 / 1.86 / 3.14
            // with 2 / 4 / 8 / 16 -- fewer, longer blocks fill the chip worse, and the atomics are not what bounds the kernel
            int seed_batches = 1;
            if (const char* e = pmx::opt_str(pmx::O_SEED_BATCHES)) seed_batches = std::max(1, atoi(e));
            // (the specialised kernel: one batch of reads per block whatever the chunk's size -- the dispatcher hands the blocks out;
            //  a grid capped at what is resident made every block walk several batches, which measured slower, see above)
            const dim3 grid(ks_path ? grid_for(r1 - r0, PMX_SEED_BLOCK * seed_batches, 1 << 30) : grid_for(r1 - r0, PMX_SEED_BLOCK, ctx->n_cu * 16)), block(PMX_SEED_BLOCK);
            // the default seeding parameters run the kernel specialised for them (same results, ~3x fewer instructions)
            if (ks_path && collapse) {
                const int64_t n_c = r1 - r0, n_tiles = (n_c + 63) / 64;
                pl->t_words[j].ensure((size_t)n_tiles * 5 * 64); pl->t_amb[j].ensure((size_t)n_tiles * 5 * 64);
                pl->t_len[j].ensure((size_t)n_tiles * 64); pl->t_mult[j].ensure((size_t)n_tiles * 64);
                PMX_HIP(hipMemsetAsync(pl->t_count.p + j, 0, sizeof(unsigned long long), st));
                hipLaunchKernelGGL(k_collapse_reads, dim3((unsigned)((n_c + PMX_DEDUP_BLOCK - 1) / PMX_DEDUP_BLOCK)), dim3(PMX_DEDUP_BLOCK), PMX_DEDUP_LDS_BYTES, st,
                                   rs->words.p, rs->amb.p, rs->woff.p, rs->off.p, r0, r1, keep, perm, sp.k, fixed_len, rs->has_recs ? rs->recs.p : (const uint8_t*)nullptr, pl->t_words[j].p, pl->t_amb[j].p, pl->t_len[j].p,
                                   pl->t_mult[j].p, pl->t_count.p + j);
                PMX_HIP(hipGetLastError());
                hipLaunchKernelGGL((l == 3 ? k_seed_histogram_ks<19, 8, 3> : k_seed_histogram_ks<19, 8, 1>), grid, block, lds_ks, st, pl->t_words[j].p, pl->t_amb[j].p,
                                   (const int64_t*)nullptr, (const int64_t*)nullptr, (int64_t)0, n_c, sp, pl->keys.p, pl->vals.p, pl->cap - 1, pl->counters.p,
                                   (const uint8_t*)nullptr, (const uint32_t*)nullptr, pl->t_len[j].p, pl->t_mult[j].p, pl->t_count.p + j);
            } else if (ks_path)
                hipLaunchKernelGGL((l == 3 ? k_seed_histogram_ks<19, 8, 3> : k_seed_histogram_ks<19, 8, 1>), grid, block, lds_ks, st, rs->words.p, rs->amb.p,
                                   rs->woff.p, rs->off.p, r0, r1, sp, pl->keys.p, pl->vals.p, pl->cap - 1, pl->counters.p, keep, perm,
                                   (const uint32_t*)nullptr, (const uint32_t*)nullptr, (const unsigned long long*)nullptr);
            else
                hipLaunchKernelGGL(k_seed_histogram, grid, block, lds, st, rs->words.p, rs->amb.p, rs->woff.p, rs->off.p, r0, r1, sp, pl->keys.p,
                                   pl->vals.p, pl->cap - 1, pl->counters.p, keep, quality_mode ? rs->qual.p : nullptr,
                                   quality_mode ? pp->min_seed_quality : 0, (uint64_t*)nullptr, (uint8_t*)nullptr, (uint32_t*)nullptr);
            PMX_HIP(hipGetLastError());
            if (j > 0) {
                PMX_HIP(hipEventRecord(ctx->seed_done[j - 1], st));
                PMX_HIP(hipStreamWaitEvent(ctx->stream, ctx->seed_done[j - 1], 0));
            }
            }
        }
        if (bound_div == 1) break;
        PMX_HIP(hipMemcpyAsync(pl->h_ctr, pl->counters.p, sizeof(pl->h_ctr), hipMemcpyDeviceToHost, ctx->stream));   // (finalize_histogram reuses them)
        PMX_HIP(hipStreamSynchronize(ctx->stream));
        pl->h_ctr_valid = true;
        const unsigned long long h_ovf = pl->h_ctr[PMX_CTR_OVERFLOW];
        if (pmx::opt_str(pmx::O_PLACE_PROF)) fprintf(stderr, "[pmx place] seeding with bound 1/%lld: table %llu slots, %llu failed inserts\n", (long long)bound_div, (unsigned long long)pl->cap, h_ovf);
        if (h_ovf == 0) break;
        // the optimistic table overflowed: start over with the safe bound
        PMX_HIP(hipMemsetAsync(pl->counters.p, 0, sizeof(unsigned long long) * PMX_CTR_N, ctx->stream));
        hipLaunchKernelGGL(k_fill_u64, dim3(grid_for((int64_t)pl->cap, 256, ctx->n_cu * 8)), dim3(256), 0, ctx->stream, pl->keys.p, PMX_EMPTY_KEY, pl->cap);
        PMX_HIP(hipMemsetAsync(pl->vals.p, 0, pl->cap * sizeof(unsigned long long), ctx->stream));
        bound_div = 1;
        }
        timer_end(ctx, "seed", 1);
    }
    pl->n_reads_added += rr1 - rr0;
    pl->bases_added += (double)(rr1 - rr0) * (double)rs->max_len;
    pl->hist_sorted = false;
    pl->table_dirty = true;
    return PMX_OK;
    PMX_CATCH
}

// --dedup over several ranks (pmx_dist_dedup_reads drives these): the local mask + the hash pairs of the kept reads, and
// the removal of the reads whose pair another rank already keeps
int64_t pmx_place_dedup_local(pmx_ctx* ctx, pmx_place* pl, const pmx_readset* rs, void* d_h1, void* d_h2, int64_t cap) {
    if (!ctx || !pl || !rs) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    dedup_local(ctx, pl, rs);
    pl->dd_count.ensure(1);
    PMX_HIP(hipMemsetAsync(pl->dd_count.p, 0, sizeof(unsigned long long), ctx->stream));
    if (!d_h1 || !d_h2) {   // count only: a reduction of the mask through the same kernel into scratch is not worth a variant
        unsigned long long kept = 0;
        std::vector<uint8_t> h((size_t)rs->n);
        if (rs->n > 0) PMX_HIP(hipMemcpyAsync(h.data(), pl->dd_keep.p, (size_t)rs->n, hipMemcpyDeviceToHost, ctx->stream));
        PMX_HIP(hipStreamSynchronize(ctx->stream));
        for (uint8_t v : h) kept += v;
        return (int64_t)kept;
    }
    if (cap < rs->n) return fail(PMX_ERR_CAPACITY, "dedup export buffers must hold one pair per read");
    if (rs->n > 0)
        hipLaunchKernelGGL(k_kept_read_hashes, dim3(grid_for(rs->n, 256, ctx->n_cu * 8)), dim3(256), 0, ctx->stream, pl->dd_h1.p, pl->dd_h2.p, pl->dd_keep.p, rs->n,
                           (uint64_t*)d_h1, (uint64_t*)d_h2, pl->dd_count.p);
    unsigned long long kept = 0;
    PMX_HIP(hipMemcpyAsync(&kept, pl->dd_count.p, sizeof(kept), hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return (int64_t)kept;
    PMX_CATCH
}

// reads the prepared mask still keeps (no recomputation)
int64_t pmx_place_dedup_local_count(pmx_ctx* ctx, pmx_place* pl, const pmx_readset* rs) {
    if (!ctx || !pl || !rs) return PMX_ERR_ARG;
    if (pl->dd_for != rs) return fail(PMX_ERR_ARG, "pmx_place_dedup_local must run on this read set first");
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    std::vector<uint8_t> h((size_t)rs->n);
    if (rs->n > 0) PMX_HIP(hipMemcpyAsync(h.data(), pl->dd_keep.p, (size_t)rs->n, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    int64_t kept = 0;
    for (uint8_t v : h) kept += v;
    return kept;
    PMX_CATCH
}

int pmx_place_dedup_drop_seen(pmx_ctx* ctx, pmx_place* pl, const pmx_readset* rs, void* d_seen_h1, void* d_seen_h2, int64_t n_seen) {
    if (!ctx || !pl || !rs || n_seen < 0 || (n_seen > 0 && (!d_seen_h1 || !d_seen_h2))) return PMX_ERR_ARG;
    if (pl->dd_for != rs) return fail(PMX_ERR_ARG, "pmx_place_dedup_local must run on this read set first");
    if (n_seen == 0 || rs->n == 0) return PMX_OK;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    DevBuf<uint64_t> s1, s2;
    s1.alloc((size_t)n_seen); s2.alloc((size_t)n_seen);
    size_t bytes = 0;
    PMX_HIP(rocprim::radix_sort_pairs(nullptr, bytes, (uint64_t*)d_seen_h1, s1.p, (uint64_t*)d_seen_h2, s2.p, (size_t)n_seen, 0, 64, ctx->stream));
    pl->tmp.ensure(bytes);
    PMX_HIP(rocprim::radix_sort_pairs(pl->tmp.p, bytes, (uint64_t*)d_seen_h1, s1.p, (uint64_t*)d_seen_h2, s2.p, (size_t)n_seen, 0, 64, ctx->stream));
    hipLaunchKernelGGL(k_drop_seen_reads, dim3(grid_for(rs->n, 256, ctx->n_cu * 8)), dim3(256), 0, ctx->stream, pl->dd_h1.p, pl->dd_h2.p, rs->n, s1.p, s2.p, n_seen,
                       pl->dd_keep.p);
    PMX_HIP(hipGetLastError());
    PMX_HIP(hipStreamSynchronize(ctx->stream));   // (s1 / s2 go out of scope)
    return PMX_OK;
    PMX_CATCH
}

int pmx_place_add_reads(pmx_ctx* ctx, pmx_place* pl, const pmx_readset* rs, const pmx_place_params* pp) {
    if (!rs) return PMX_ERR_ARG;
    return add_reads_impl(ctx, pl, rs, 0, rs->n, pp);
}

int pmx_place_add_reads_range(pmx_ctx* ctx, pmx_place* pl, const pmx_readset* rs, int64_t r0, int64_t r1, const pmx_place_params* pp) {
    return add_reads_impl(ctx, pl, rs, r0, r1, pp);
}

int64_t pmx_place_histogram_size(pmx_ctx* ctx, pmx_place* pl) {
    if (!ctx || !pl) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    finalize_histogram(ctx, pl);
    return pl->n_hist;
    PMX_CATCH
}

int pmx_place_histogram_export(pmx_ctx* ctx, pmx_place* pl, uint64_t* hash, int64_t* count, int64_t cap) {
    if (!ctx || !pl || (cap > 0 && (!hash || !count))) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    finalize_histogram(ctx, pl);
    if (cap < pl->n_hist) return fail(PMX_ERR_CAPACITY, "histogram export buffer too small");
    if (pl->n_hist > 0) {
        PMX_HIP(hipMemcpyAsync(hash, pl->hist_hash.p, sizeof(uint64_t) * pl->n_hist, hipMemcpyDeviceToHost, ctx->stream));
        PMX_HIP(hipMemcpyAsync(count, pl->hist_count.p, sizeof(int64_t) * pl->n_hist, hipMemcpyDeviceToHost, ctx->stream));
    }
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
    PMX_CATCH
}

int pmx_place_histogram_export_device(pmx_ctx* ctx, pmx_place* pl, void* d_hash, void* d_count, int64_t cap) {
    if (!ctx || !pl || (cap > 0 && (!d_hash || !d_count))) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    finalize_histogram(ctx, pl);
    if (cap < pl->n_hist) return fail(PMX_ERR_CAPACITY, "histogram export buffer too small");
    if (pl->n_hist > 0) {
        PMX_HIP(hipMemcpyAsync(d_hash, pl->hist_hash.p, sizeof(uint64_t) * pl->n_hist, hipMemcpyDeviceToDevice, ctx->stream));
        PMX_HIP(hipMemcpyAsync(d_count, pl->hist_count.p, sizeof(int64_t) * pl->n_hist, hipMemcpyDeviceToDevice, ctx->stream));
    }
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
    PMX_CATCH
}

// The multi-GPU exchange needs every rank's (hash, count) pairs, not their order: the distinct seeds of the table, counted
// without the compaction + sort of finalize_histogram, and written -- in table order -- straight into the caller's buffers.
// The one sorted histogram is made after the other ranks' parts were merged (pmx_place_score).
int64_t pmx_place_histogram_entries(pmx_ctx* ctx, pmx_place* pl) {
    if (!ctx || !pl) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    if (pl->hist_sorted) return pl->n_hist;
    if (!pl->h_ctr_valid) {
        PMX_HIP(hipMemcpyAsync(pl->h_ctr, pl->counters.p, sizeof(pl->h_ctr), hipMemcpyDeviceToHost, ctx->stream));
        PMX_HIP(hipStreamSynchronize(ctx->stream));
        pl->h_ctr_valid = true;
    }
    if (pl->h_ctr[PMX_CTR_OVERFLOW]) throw std::runtime_error("seed table overflow (internal sizing error)");
    int64_t n = 0;
    for (int i = 0; i < PMX_CTR_NSHARD; ++i) n += (int64_t)pl->h_ctr[PMX_CTR_SHARD0 + i];
    return n;
    PMX_CATCH
}

int pmx_place_histogram_export_device_unsorted(pmx_ctx* ctx, pmx_place* pl, void* d_hash, void* d_count, int64_t cap) {
    if (!ctx || !pl || (cap > 0 && (!d_hash || !d_count))) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    const int64_t n = pmx_place_histogram_entries(ctx, pl);
    if (n < 0) return (int)n;
    if (cap < n) return fail(PMX_ERR_CAPACITY, "histogram export buffer too small");
    if (n == 0) return PMX_OK;
    if (pl->hist_sorted) {   // already finalized: the sorted arrays are as good
        PMX_HIP(hipMemcpyAsync(d_hash, pl->hist_hash.p, sizeof(uint64_t) * n, hipMemcpyDeviceToDevice, ctx->stream));
        PMX_HIP(hipMemcpyAsync(d_count, pl->hist_count.p, sizeof(int64_t) * n, hipMemcpyDeviceToDevice, ctx->stream));
        return PMX_OK;
    }
    PMX_HIP(hipMemsetAsync(pl->counters.p + PMX_CTR_COMPACT, 0, sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(k_table_compact, dim3(grid_for((int64_t)pl->cap, 256, ctx->n_cu * 8)), dim3(256), 0, ctx->stream, pl->keys.p, pl->vals.p, pl->cap,
                       (uint64_t*)d_hash, (int64_t*)d_count, pl->counters.p + PMX_CTR_COMPACT);
    PMX_HIP(hipGetLastError());
    return PMX_OK;   // (stream-ordered: the caller synchronizes before another stream reads the buffers)
    PMX_CATCH
}

int pmx_place_histogram_merge_device(pmx_ctx* ctx, pmx_place* pl, const void* d_hash, const void* d_count, int64_t n) {
    if (!ctx || !pl || n < 0 || (n > 0 && (!d_hash || !d_count))) return PMX_ERR_ARG;
    if (n == 0) return PMX_OK;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    table_reserve(ctx, pl, (uint64_t)n);
    hipLaunchKernelGGL(k_table_merge, dim3(grid_for(n, 256, ctx->n_cu * 8)), dim3(256), 0, ctx->stream, (const uint64_t*)d_hash,
                       (const int64_t*)d_count, n, pl->keys.p, pl->vals.p, pl->cap - 1, pl->counters.p);
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    pl->hist_sorted = false;
    pl->table_dirty = true;
    pl->h_ctr_valid = false;
    return PMX_OK;
    PMX_CATCH
}

int pmx_place_histogram_merge_device_parts(pmx_ctx* ctx, pmx_place* pl, const void* d_hash, const void* d_count, int64_t part_stride,
                                           const int64_t* sizes, int n_parts, int skip_part) {
    if (!ctx || !pl || !sizes || n_parts < 0 || part_stride < 0) return PMX_ERR_ARG;
    int64_t total = 0;
    for (int p = 0; p < n_parts; ++p) {
        if (sizes[p] < 0 || sizes[p] > part_stride) return PMX_ERR_ARG;
        if (p != skip_part) total += sizes[p];
    }
    if (total == 0) return PMX_OK;
    if (!d_hash || !d_count) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    table_reserve(ctx, pl, (uint64_t)total);   // one capacity check for every part, then the merges run back to back
    for (int p = 0; p < n_parts; ++p) {
        if (p == skip_part || sizes[p] == 0) continue;
        hipLaunchKernelGGL(k_table_merge, dim3(grid_for(sizes[p], 256, ctx->n_cu * 8)), dim3(256), 0, ctx->stream,
                           (const uint64_t*)d_hash + (size_t)p * (size_t)part_stride, (const int64_t*)d_count + (size_t)p * (size_t)part_stride, sizes[p],
                           pl->keys.p, pl->vals.p, pl->cap - 1, pl->counters.p);
    }
    PMX_HIP(hipGetLastError());
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    pl->hist_sorted = false;
    pl->table_dirty = true;
    pl->h_ctr_valid = false;
    return PMX_OK;
    PMX_CATCH
}

int pmx_place_histogram_merge(pmx_ctx* ctx, pmx_place* pl, const uint64_t* hash, const int64_t* count, int64_t n) {
    if (!ctx || !pl || n < 0 || (n > 0 && (!hash || !count))) return PMX_ERR_ARG;
    if (n == 0) return PMX_OK;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    table_reserve(ctx, pl, (uint64_t)n);
    DevBuf<uint64_t> dh;
    DevBuf<int64_t> dc;
    dh.alloc(n); dc.alloc(n);
    PMX_HIP(hipMemcpyAsync(dh.p, hash, sizeof(uint64_t) * n, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(dc.p, count, sizeof(int64_t) * n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_table_merge, dim3(grid_for(n, 256, ctx->n_cu * 8)), dim3(256), 0, ctx->stream, dh.p, dc.p, n, pl->keys.p, pl->vals.p,
                       pl->cap - 1, pl->counters.p);
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    pl->hist_sorted = false;
    pl->table_dirty = true;
    pl->h_ctr_valid = false;
    return PMX_OK;
    PMX_CATCH
}

int pmx_place_score(pmx_ctx* ctx, pmx_place* pl, const pmx_place_params* pp, int64_t n_reads_total, pmx_place_result* res) {
    if (!ctx || !pl || !pp || !res) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    std::memset(res, 0, sizeof(*res));
    for (int m = 0; m < 5; ++m) res->best_index[m] = UINT32_MAX;
    // PMX_PLACE_PROF=1: host wall time of the sections of this call (each mark synchronises the stream first)
    const bool prof = pmx::opt_str(pmx::O_PLACE_PROF) != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!prof) return;
        (void)hipStreamSynchronize(ctx->stream);
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[pmx place score] %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    finalize_histogram(ctx, pl);
    mark("finalize_histogram (compact + sort)");
    const int64_t n = pl->n_hist;
    const int G = ctx->n_cu * 8;
    hipStream_t st = ctx->stream;
    res->n_reads = n_reads_total;

    // ---- read-side filters (src/placement.cpp:1703-1856)
    pl->dead.ensure(n);
    pl->flag.ensure(n + 1);
    pl->pos.ensure(n + 1);
    pl->kept_hash.ensure(n);
    pl->kept_log.ensure(n);
    int64_t n_kept = 0, min_support = pp->min_read_support;
    unsigned long long h_stats[4] = {0, 0, 0, 0};
    if (n > 0) {
        hipLaunchKernelGGL(k_mark_homopolymer, dim3(grid_for(n, 256, G)), dim3(256), 0, st, pl->hist_hash.p, n, homopolymer_hash(0, pl->params.k),
                           homopolymer_hash(1, pl->params.k), homopolymer_hash(2, pl->params.k), homopolymer_hash(3, pl->params.k), pl->dead.p);
        if (pp->seed_mask_fraction > 0.0) {
            // unique seeds after homopolymer erase
            PMX_HIP(hipMemsetAsync(pl->stats.p, 0, 4 * sizeof(unsigned long long), st));
            hipLaunchKernelGGL(k_hist_stats, dim3(grid_for(n, 256, ctx->n_cu)), dim3(256), 0, st, pl->hist_count.p, pl->dead.p, n, pl->stats.p);
            PMX_HIP(hipMemcpyAsync(h_stats, pl->stats.p, sizeof(h_stats), hipMemcpyDeviceToHost, st));
            PMX_HIP(hipStreamSynchronize(st));
            const int64_t n_mask = (int64_t)(pp->seed_mask_fraction * (double)h_stats[3]);   // :1775
            if (n_mask > 0) {
                DevBuf<uint64_t> key, key2;
                DevBuf<uint32_t> idx, idx2;
                key.alloc(n); key2.alloc(n); idx.alloc(n); idx2.alloc(n);
                hipLaunchKernelGGL(k_mask_keys, dim3(grid_for(n, 256, G)), dim3(256), 0, st, pl->hist_count.p, pl->dead.p, n, key.p, idx.p);
                size_t bytes = 0;
                PMX_HIP(rocprim::radix_sort_pairs(nullptr, bytes, key.p, key2.p, idx.p, idx2.p, (size_t)n, 0, 64, st));
                pl->tmp.ensure(bytes);
                PMX_HIP(rocprim::radix_sort_pairs(pl->tmp.p, bytes, key.p, key2.p, idx.p, idx2.p, (size_t)n, 0, 64, st));
                const int64_t nm = std::min<int64_t>(n_mask, (int64_t)h_stats[3]);
                hipLaunchKernelGGL(k_mask_apply, dim3(grid_for(nm, 256, G)), dim3(256), 0, st, idx2.p, nm, pl->dead.p);
                PMX_HIP(hipStreamSynchronize(st));
            }
        }
        PMX_HIP(hipMemsetAsync(pl->stats.p, 0, 4 * sizeof(unsigned long long), st));
        hipLaunchKernelGGL(k_hist_stats, dim3(grid_for(n, 256, ctx->n_cu)), dim3(256), 0, st, pl->hist_count.p, pl->dead.p, n, pl->stats.p);
        PMX_HIP(hipMemcpyAsync(h_stats, pl->stats.p, sizeof(h_stats), hipMemcpyDeviceToHost, st));
        PMX_HIP(hipStreamSynchronize(st));
        if (min_support < 0) {   // resolveMinReadSupport, src/placement.cpp:931-955
            const double est = h_stats[1] > 0 ? (double)h_stats[0] / (double)h_stats[1] : 0.0;
            min_support = est > 3.0 ? 2 : 1;
        }
        hipLaunchKernelGGL(k_keep_flags, dim3(grid_for(n, 256, G)), dim3(256), 0, st, pl->hist_count.p, pl->dead.p, n, min_support, pl->flag.p);
        size_t bytes = 0;
        PMX_HIP(rocprim::exclusive_scan(nullptr, bytes, pl->flag.p, pl->pos.p, 0u, (size_t)n, rocprim::plus<uint32_t>(), st));
        pl->tmp.ensure(bytes);
        PMX_HIP(rocprim::exclusive_scan(pl->tmp.p, bytes, pl->flag.p, pl->pos.p, 0u, (size_t)n, rocprim::plus<uint32_t>(), st));
        uint32_t last_pos = 0, last_flag = 0;
        PMX_HIP(hipMemcpyAsync(&last_pos, pl->pos.p + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        PMX_HIP(hipMemcpyAsync(&last_flag, pl->flag.p + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        PMX_HIP(hipStreamSynchronize(st));
        n_kept = (int64_t)last_pos + last_flag;
        hipLaunchKernelGGL(k_keep_scatter, dim3(grid_for(n, 256, G)), dim3(256), 0, st, pl->hist_hash.p, pl->hist_count.p, pl->flag.p, pl->pos.p, n,
                           pl->kept_hash.p, pl->kept_log.p);
    } else if (min_support < 0) min_support = 1;
    pl->n_kept = n_kept;
    mark("read-side filters (homopolymer, stats, keep scan/scatter)");
    // canonical-order sums (src/placement.cpp:957-984)
    {
        const int64_t nb = (n_kept + PMX_SUM_BLOCK - 1) / PMX_SUM_BLOCK;
        pl->partial.ensure((size_t)(2 * nb + 2));
        if (nb > 0) hipLaunchKernelGGL(k_block_sums, dim3((unsigned)std::min<int64_t>(nb, (int64_t)G * 4)), dim3(64), 0, st, pl->kept_log.p, n_kept, pl->partial.p);
        hipLaunchKernelGGL(k_sequential_sums, dim3(1), dim3(64), 0, st, pl->partial.p, nb, pl->scalars.p);
    }
    // probe table for the kept seeds
    pl->tcap = next_pow2((uint64_t)std::max<int64_t>(n_kept, 1) * 2 + 64);
    pl->tkeys.ensure(pl->tcap);
    pl->tvals.ensure(pl->tcap);
    hipLaunchKernelGGL(k_fill_u64, dim3(grid_for((int64_t)pl->tcap, 256, G)), dim3(256), 0, st, pl->tkeys.p, PMX_EMPTY_KEY, pl->tcap);
    if (n_kept > 0)
        hipLaunchKernelGGL(k_kept_table_build, dim3(grid_for(n_kept, 256, G)), dim3(256), 0, st, pl->kept_hash.p, pl->kept_log.p, n_kept, pl->tkeys.p,
                           pl->tvals.p, pl->tcap - 1);
    hipLaunchKernelGGL(k_wc_denominator, dim3(1), dim3(1024), 0, st, pl->ch_hash.p, pl->ch_child.p, pl->root_beg, pl->root_end, pl->tkeys.p, pl->tvals.p,
                       pl->tcap - 1, n_kept > 0 ? 1 : 0, pl->scalars.p + 2);
    double h_scal[3] = {0, 0, 0};   // fetched with the scores below (the device reads the scalars itself)
    mark("sums, probe table, denominators");

    // ---- node scoring, one launch per BFS level (src/placement.cpp:701-918)
    const int n_levels = (int)pl->level_off.size() - 1;
    timer_begin(ctx, "score");
    const int64_t n_ch = pl->n_changes;
    pl->terms.ensure((size_t)std::max<int64_t>(n_ch, 1) * 5);
    pl->term_meta.ensure((size_t)std::max<int64_t>(n_ch, 1));
    double* t_mag = pl->terms.p;
    double *t_raw = t_mag + n_ch, *t_cos = t_mag + 2 * n_ch, *t_wc = t_mag + 3 * n_ch, *t_lc = t_mag + 4 * n_ch;
    if (n_ch > 0)
        hipLaunchKernelGGL(k_score_terms, dim3(grid_for(n_ch, 256, G)), dim3(256), 0, st, pl->ch_hash.p, pl->ch_par.p, pl->ch_child.p, n_ch,
                           pl->tkeys.p, pl->tvals.p, pl->tcap - 1, n_kept > 0 ? 1 : 0, t_mag, t_raw, t_cos, t_wc, t_lc, pl->term_meta.p);
    // The level launches are a fixed, launch-bound chain (122 dependent launches for the SARS tree, none of
    // whose arguments change between calls): captured once into a HIP graph and replayed.
    auto launch_levels = [&]() {
        for (int lv = 0; lv < n_levels; ++lv) {
            const int64_t beg = pl->level_off[lv], cnt = pl->level_off[lv + 1] - beg;
            if (cnt <= 0) continue;
            hipLaunchKernelGGL(k_score_level, dim3((unsigned)((cnt + 3) / 4)), dim3(256), 0, st, pl->level_nodes.p + beg, cnt, pl->parent.p,
                               pl->offsets.p, t_mag, t_raw, t_cos, t_wc, t_lc, pl->term_meta.p, pl->metrics5.p, pl->counts2.p);
        }
    };
    const void* sig[3] = {(const void*)t_mag, (const void*)pl->metrics5.p, (const void*)pl->term_meta.p};
    bool tree_kernel = false;
    if (!pmx::opt_str(pmx::O_PLACE_LEVEL_KERNELS)) {
        // one persistent launch, parent -> child through per-node flags (k_score_tree); one workgroup per CU so that
        // every wave is resident
        if (!pl->tree_done.p) {
            pl->tree_done.alloc((size_t)pl->n_nodes + 1);
            PMX_HIP(hipMemsetAsync(pl->tree_done.p, 0, sizeof(uint32_t) * ((size_t)pl->n_nodes + 1), st));
            pl->tree_epoch = 0;
        }
        if (++pl->tree_epoch == 0) {   // wrapped: start over
            PMX_HIP(hipMemsetAsync(pl->tree_done.p, 0, sizeof(uint32_t) * ((size_t)pl->n_nodes + 1), st));
            pl->tree_epoch = 1;
        }
        if (pmx::opt_str(pmx::O_PLACE_TREE_KERNEL)) {   // per-node flags in BFS order (kept for comparison)
            const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ctx->n_cu, (pl->n_nodes + 3) / 4));
            hipLaunchKernelGGL(k_score_tree, dim3(grid), dim3(256), 0, st, pl->level_nodes.p, pl->n_nodes, pl->parent.p, pl->offsets.p, t_mag, t_raw,
                               t_cos, t_wc, t_lc, pl->term_meta.p, pl->metrics5.p, pl->counts2.p, pl->tree_done.p, pl->tree_epoch,
                               pl->tree_done.p + pl->n_nodes);
        } else {
            const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ctx->n_cu, (pl->n_chains + 3) / 4));
            hipLaunchKernelGGL(k_score_chains, dim3(grid), dim3(256), 0, st, pl->chain_off.p, pl->n_chains, pl->chain_nodes.p, pl->chain_beg.p,
                               pl->chain_end.p, pl->parent.p, t_mag, t_raw, t_cos, t_wc, t_lc, pl->term_meta.p, pl->metrics5.p, pl->counts2.p,
                               pl->tree_done.p, pl->tree_epoch, pl->tree_done.p + pl->n_nodes);
        }
        tree_kernel = true;
        if (pmx::opt_str(pmx::O_PLACE_TEST_STARVED)) {   // tests: pretend a wave gave up, so that the level-kernel redo runs
            const uint32_t one = 1;
            PMX_HIP(hipMemcpyAsync(pl->tree_done.p + pl->n_nodes, &one, sizeof(one), hipMemcpyHostToDevice, st));
        }
    } else if (pmx::opt_str(pmx::O_PLACE_NO_GRAPH)) launch_levels();
    else {
        if (!pl->level_graph_exec || pl->level_graph_sig[0] != sig[0] || pl->level_graph_sig[1] != sig[1] || pl->level_graph_sig[2] != sig[2]) {
            if (pl->level_graph_exec) { (void)hipGraphExecDestroy(pl->level_graph_exec); pl->level_graph_exec = nullptr; }
            hipGraph_t graph = nullptr;
            PMX_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            launch_levels();
            PMX_HIP(hipStreamEndCapture(st, &graph));
            PMX_HIP(hipGraphInstantiate(&pl->level_graph_exec, graph, nullptr, nullptr, 0));
            (void)hipGraphDestroy(graph);
            for (int q = 0; q < 3; ++q) pl->level_graph_sig[q] = sig[q];
        }
        PMX_HIP(hipGraphLaunch(pl->level_graph_exec, st));
    }
    timer_end(ctx, "score", 1);
    uint32_t tree_status = 0;
    auto finish = [&]() {   // node scores from the accumulators, everything the host needs back in one synchronisation
        hipLaunchKernelGGL(k_score_getters, dim3(grid_for(pl->n_nodes, 256, G)), dim3(256), 0, st, pl->metrics5.p, pl->counts2.p, pl->n_nodes, pl->scalars.p,
                           n_kept, pl->scores5.p, pl->level_nodes.p, pl->scores_bfs.p);
        PMX_HIP(hipGetLastError());
        pl->h_scores.resize(5 * (size_t)pl->n_nodes);
        PMX_HIP(hipMemcpyAsync(pl->h_scores.data(), pl->scores_bfs.p, sizeof(double) * 5 * (size_t)pl->n_nodes, hipMemcpyDeviceToHost, st));
        PMX_HIP(hipMemcpyAsync(h_scal, pl->scalars.p, sizeof(h_scal), hipMemcpyDeviceToHost, st));
        if (tree_kernel) PMX_HIP(hipMemcpyAsync(&tree_status, pl->tree_done.p + pl->n_nodes, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        PMX_HIP(hipStreamSynchronize(st));
    };
    finish();
    mark("terms + tree scoring + getters + D2H of the scores");
    if (tree_status != 0) {
        // The persistent kernel is only correct while all of its workgroups are resident; a wave that polled 2^24 times
        // without seeing its parent's flag gave up (the GPU is shared with another process, or with collectives in
        // flight).  The level kernels need no co-residency and add the same terms in the same order: redo with them.
        PMX_HIP(hipMemsetAsync(pl->tree_done.p + pl->n_nodes, 0, sizeof(uint32_t), st));
        tree_kernel = false;
        tree_status = 0;
        launch_levels();
        PMX_HIP(hipGetLastError());
        finish();
        mark("level-kernel redo after a starved persistent launch");
    }

    // ---- sequential best/tie rule in BFS visit order (src/placement.cpp:355-401)
    // Every improvement of `best` resets the tie list to the improving node, so only the visit positions after the
    // LAST improvement can contribute ties: pass 1 replays just the scalar part of the rule (best, idx, position of
    // the last improvement), pass 2 runs the full rule from that position on.  Same result as one pass with the
    // vectors, without pushing and clearing tens of thousands of transient ties.
    Best best[5];
    {
        const int64_t nn = pl->n_nodes;
        std::vector<uint8_t> flag((size_t)nn);
        for (int m = 0; m < 5; ++m) {
            const double* sc = pl->h_scores.data() + (size_t)m * (size_t)nn;   // this metric, in visit order
            double b = 0.0, thr = 0.0 + std::max(0.0 * 0.0001, 1e-9);   // thr = best + tol, recomputed only when best moves
            int64_t last = -1;
            if (pp->force_leaf) {
                for (int64_t j = 0; j < nn; ++j) {
                    if (pl->h_has_child[pl->h_order[j]]) continue;
                    if (sc[j] > thr) { b = sc[j]; thr = b + std::max(b * 0.0001, 1e-9); last = j; }
                }
            } else {
                for (int64_t j = 0; j < nn; ++j)
                    if (sc[j] > thr) { b = sc[j]; thr = b + std::max(b * 0.0001, 1e-9); last = j; }
            }
            if (last < 0) {   // never improved (all scores ~0): the plain rule, start to end
                for (int64_t j = 0; j < nn; ++j) {
                    const uint32_t nd = pl->h_order[j];
                    if (pp->force_leaf && pl->h_has_child[nd]) continue;
                    best[m].update(nd, sc[j]);
                }
                continue;
            }
            // after the last improvement `best` is fixed, so a later node ties iff score >= best - tol (and > 0); the
            // rule's list, once sorted and de-duplicated, is {idx} + those nodes: flagged and read back in id order
            best[m].best = b;
            best[m].idx = pl->h_order[last];
            const double tol = std::max(b * 0.0001, 1e-9);
            std::fill(flag.begin(), flag.end(), (uint8_t)0);
            flag[best[m].idx] = 1;
            for (int64_t j = last + 1; j < nn; ++j) {
                if (!(sc[j] >= b - tol && sc[j] > 0)) continue;
                const uint32_t nd = pl->h_order[j];
                if (pp->force_leaf && pl->h_has_child[nd]) continue;
                flag[nd] = 1;
            }
            for (int64_t nd = 0; nd < nn; ++nd)
                if (flag[(size_t)nd]) best[m].tied.push_back((uint32_t)nd);
        }
    }
    for (int m = 0; m < 5; ++m) {
        std::vector<uint32_t>& t = best[m].tied;
        if (!t.empty()) {
            std::sort(t.begin(), t.end());
            t.erase(std::unique(t.begin(), t.end()), t.end());
            best[m].idx = t.front();
        }
        pl->tied[m] = t;
        res->best_score[m] = best[m].best;
        res->best_index[m] = best[m].idx;
        res->n_tied[m] = (int64_t)t.size();
    }
    mark("best / tie rule on the host");
    res->n_unique_seeds = (int64_t)h_stats[3];
    res->n_kept_seeds = n_kept;
    res->total_seed_freq = (int64_t)h_stats[2];
    res->min_support = min_support;
    res->log_read_magnitude = std::sqrt(h_scal[0]);
    res->log_containment_den = h_scal[1];
    res->weighted_containment_den = h_scal[2];
    return PMX_OK;
    PMX_CATCH
}

int pmx_place_tied(const pmx_place* pl, int metric, uint32_t* out, int64_t cap) {
    if (!pl || metric < 0 || metric >= 5) return PMX_ERR_ARG;
    const std::vector<uint32_t>& t = pl->tied[metric];
    if ((int64_t)t.size() > cap) return fail(PMX_ERR_CAPACITY, "tie buffer too small");
    if (!t.empty()) std::memcpy(out, t.data(), sizeof(uint32_t) * t.size());
    return PMX_OK;
}

int pmx_place_node_outputs(pmx_ctx* ctx, pmx_place* pl, double* scores5, double* metrics5, int64_t* counts2) {
    if (!ctx || !pl) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)pl->n_nodes;
    if (scores5) PMX_HIP(hipMemcpyAsync(scores5, pl->scores5.p, sizeof(double) * 5 * n, hipMemcpyDeviceToHost, ctx->stream));
    if (metrics5) PMX_HIP(hipMemcpyAsync(metrics5, pl->metrics5.p, sizeof(double) * 5 * n, hipMemcpyDeviceToHost, ctx->stream));
    if (counts2) PMX_HIP(hipMemcpyAsync(counts2, pl->counts2.p, sizeof(int64_t) * 2 * n, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
    PMX_CATCH
}

int64_t pmx_place_kept_seeds(pmx_ctx* ctx, pmx_place* pl, uint64_t* hash, double* logc, int64_t cap) {
    if (!ctx || !pl) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    if (cap < pl->n_kept) return pl->n_kept;
    if (pl->n_kept > 0) {
        if (hash) PMX_HIP(hipMemcpyAsync(hash, pl->kept_hash.p, sizeof(uint64_t) * pl->n_kept, hipMemcpyDeviceToHost, ctx->stream));
        if (logc) PMX_HIP(hipMemcpyAsync(logc, pl->kept_log.p, sizeof(double) * pl->n_kept, hipMemcpyDeviceToHost, ctx->stream));
    }
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return pl->n_kept;
    PMX_CATCH
}

}  // extern "C"
