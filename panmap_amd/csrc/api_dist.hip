// C ABI, device part 3: the multi-GPU exchange steps (see include/panmap_amd.h, "multi-GPU").
// One rank per GPU (a process, or a thread with its own context).  The reference is a single process: what is exchanged
// follows SURVEY.md section 8e -- the per-rank seed histograms (all-gather, merged on every rank, so that node scoring is
// replicated and deterministic: integer sums, no floating-point reduction) and the alignment records + CIGAR arenas
// (gathered to one rank, cigar_off rebased, for the BAM writer).
//
// Transport: RCCL (librccl.so.1, loaded with dlopen on first use: a one-GPU run never maps its 300 MB), all operations
// stream-ordered on the context's stream.  PMX_DIST_HOST_DIR=<dir> selects a file-based transport through host memory
// instead -- the functional test on a box with fewer GPUs than ranks (RCCL refuses two ranks on one device); it is never
// used for a reported number.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "align_kernel.h"
#include "device/dev_util.hpp"

using namespace pmx;

// pieces of the place / align objects this unit needs (defined next to them)
int64_t pmx_place_histogram_entries(pmx_ctx* ctx, pmx_place* pl);
extern "C" int64_t pmx_place_dedup_local_count(pmx_ctx* ctx, pmx_place* pl, const pmx_readset* rs);

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

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

const RcclApi& rccl() {
    static RcclApi api = [] {
        RcclApi a;
        // (a process that already mapped a librccl.so.1 -- torch ships one -- gets that copy: same SONAME)
        a.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!a.handle) a.handle = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!a.handle) throw std::runtime_error(std::string("RCCL is not available: ") + dlerror());
        auto sym = [&](const char* n) {
            void* p = dlsym(a.handle, n);
            if (!p) throw std::runtime_error(std::string("librccl lacks ") + n);
            return p;
        };
        a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
        a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
        a.Send = (decltype(a.Send))sym("ncclSend");
        a.Recv = (decltype(a.Recv))sym("ncclRecv");
        a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
        a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
        a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
        return a;
    }();
    return api;
}

#define PMX_NCCL(expr)                                                                                          \
    do {                                                                                                        \
        ncclResult_t _r = (expr);                                                                               \
        if (_r != ncclSuccess) throw std::runtime_error(std::string(#expr) + ": " + rccl().GetErrorString(_r)); \
    } while (0)

// ---- what the exchange steps need from a transport -----------------------------------------------------------------------
struct Transport {
    int rank = 0, world = 1;
    virtual ~Transport() {}
    // every rank's `bytes` bytes of DEVICE memory, rank-major, into d_all (world * bytes); stream-ordered
    virtual void all_gather(hipStream_t st, const void* d_mine, size_t bytes, void* d_all) = 0;
    // rank r's sizes[r] bytes to offset offs[r] of d_all on `root` (device memory; root's own part included); stream-ordered
    virtual void gather_v(hipStream_t st, const void* d_mine, const std::vector<size_t>& sizes, const std::vector<size_t>& offs, void* d_all, int root) = 0;
};

struct RcclTransport : Transport {
    ncclComm_t comm = nullptr;
    ~RcclTransport() override {
        if (comm) (void)rccl().CommDestroy(comm);
    }
    void all_gather(hipStream_t st, const void* d_mine, size_t bytes, void* d_all) override {
        PMX_NCCL(rccl().AllGather(d_mine, d_all, bytes, ncclChar, comm, st));
    }
    void gather_v(hipStream_t st, const void* d_mine, const std::vector<size_t>& sizes, const std::vector<size_t>& offs, void* d_all, int root) override {
        // point-to-point inside one group: exact sizes, no padding (RCCL has no gather-v)
        PMX_NCCL(rccl().GroupStart());
        if (rank == root) {
            for (int r = 0; r < world; ++r) {
                if (r == root || sizes[r] == 0) continue;
                PMX_NCCL(rccl().Recv((char*)d_all + offs[r], sizes[r], ncclChar, r, comm, st));
            }
        } else if (sizes[rank] > 0) {
            PMX_NCCL(rccl().Send(d_mine, sizes[rank], ncclChar, root, comm, st));
        }
        PMX_NCCL(rccl().GroupEnd());
        if (rank == root && sizes[root] > 0)
            PMX_HIP(hipMemcpyAsync((char*)d_all + offs[root], d_mine, sizes[root], hipMemcpyDeviceToDevice, st));
    }
};

// Test transport: every operation is a round of files <dir>/x<nonce>.<seq>.<rank> (written under a temporary name, then
// renamed).  A rank removes its file of round k once round k+1 is complete (everybody has read round k by then).  The file
// of the LAST round is never removed by the transport: only rounds before it are known to have been read by every peer (a
// rank that returns from the final round first would otherwise unlink its file under a peer that still polls for it).
// Whoever made the directory removes it.  `nonce` = the first bytes of the communicator id the ranks were given, so that a
// reused directory with files of another (crashed) run is not taken for this run's round 0 when the caller ships an id.
struct HostDirTransport : Transport {
    std::string dir, nonce;
    uint64_t seq = 0;
    std::string prev_file;
    std::string name(uint64_t s, int r) const { return dir + "/x" + nonce + "." + std::to_string(s) + "." + std::to_string(r); }
    void publish(const std::vector<char>& buf) {
        const std::string fin = name(seq, rank), tmp = fin + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb");
        if (!f) throw std::runtime_error("host transport: cannot write " + tmp);
        if (!buf.empty() && fwrite(buf.data(), 1, buf.size(), f) != buf.size()) { fclose(f); throw std::runtime_error("host transport: short write"); }
        fclose(f);
        if (rename(tmp.c_str(), fin.c_str()) != 0) throw std::runtime_error("host transport: rename failed");
    }
    std::vector<char> take(int r) {
        const std::string fn = name(seq, r);
        const auto t0 = std::chrono::steady_clock::now();
        struct stat sb;
        while (stat(fn.c_str(), &sb) != 0) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0) throw std::runtime_error("host transport: rank " + std::to_string(r) + " did not arrive");
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        std::vector<char> buf((size_t)sb.st_size);
        FILE* f = fopen(fn.c_str(), "rb");
        if (!f || (buf.size() && fread(buf.data(), 1, buf.size(), f) != buf.size())) { if (f) fclose(f); throw std::runtime_error("host transport: cannot read " + fn); }
        fclose(f);
        return buf;
    }
    void round_done() {
        if (!prev_file.empty()) (void)unlink(prev_file.c_str());
        prev_file = name(seq, rank);
        ++seq;
    }
    void all_gather(hipStream_t st, const void* d_mine, size_t bytes, void* d_all) override {
        std::vector<char> mine(bytes);
        if (bytes) PMX_HIP(hipMemcpyAsync(mine.data(), d_mine, bytes, hipMemcpyDeviceToHost, st));
        PMX_HIP(hipStreamSynchronize(st));
        publish(mine);
        for (int r = 0; r < world; ++r) {
            std::vector<char> b = r == rank ? mine : take(r);
            if (b.size() != bytes) throw std::runtime_error("host transport: size mismatch in all_gather");
            if (bytes) PMX_HIP(hipMemcpy((char*)d_all + (size_t)r * bytes, b.data(), bytes, hipMemcpyHostToDevice));
        }
        round_done();
    }
    void gather_v(hipStream_t st, const void* d_mine, const std::vector<size_t>& sizes, const std::vector<size_t>& offs, void* d_all, int root) override {
        std::vector<char> mine(sizes[rank]);
        if (!mine.empty()) PMX_HIP(hipMemcpyAsync(mine.data(), d_mine, mine.size(), hipMemcpyDeviceToHost, st));
        PMX_HIP(hipStreamSynchronize(st));
        publish(rank == root ? std::vector<char>() : mine);
        for (int r = 0; r < world; ++r) {
            if (r == rank) continue;
            std::vector<char> b = take(r);                 // (every rank reads every file: the round doubles as a barrier)
            if (rank == root) {
                if (b.size() != sizes[r]) throw std::runtime_error("host transport: size mismatch in gather");
                if (!b.empty()) PMX_HIP(hipMemcpy((char*)d_all + offs[r], b.data(), b.size(), hipMemcpyHostToDevice));
            }
        }
        if (rank == root && !mine.empty()) PMX_HIP(hipMemcpy((char*)d_all + offs[root], mine.data(), mine.size(), hipMemcpyHostToDevice));
        round_done();
    }
};

// records of the ranks behind the first one point into their own arenas: add the rank's base in the merged arena
__global__ void k_rebase_cigars(pmx::aln::AlnRecord* recs, int64_t n, uint32_t base) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (recs[i].flags & PMX_REC_HAS_ALN) recs[i].cigar_off += base;
}
}  // namespace

struct pmx_dist {
    pmx_ctx* ctx = nullptr;
    std::unique_ptr<Transport> tp;
    DevBuf<int64_t> d_meta;                 // small all-gathers of counts
    DevBuf<uint64_t> hist_mine, hist_all;   // [2][max] per rank: hash plane, count plane
    DevBuf<pmx_aln_record> g_records;       // on the root: every rank's records, rank order
    DevBuf<uint32_t> g_cigars;              // ... and arenas back to back
    DevBuf<pmx_aln_record> s_records;       // pmx_dist_plan_alignments: this rank's records with cigar_off rebased
    int64_t g_n_records = 0, g_n_words = 0;
    int64_t shard_records = 0, shard_words = 0, shard_record_base = 0, shard_word_base = 0;   // pmx_dist_plan_alignments
    std::vector<int64_t> rank_records, rank_words;
    hipEvent_t ev_gathered = nullptr, ev_fetched = nullptr;   // pmx_dist_fetch_gathered_async
    bool fetch_pending = false;

    // every rank's k int64 values -> host, rank-major (one tiny all-gather + one read-back)
    std::vector<int64_t> exchange_counts(const int64_t* mine, int k) {
        const int world = tp->world;
        d_meta.ensure((size_t)k * (size_t)(world + 1));
        PMX_HIP(hipMemcpyAsync(d_meta.p, mine, sizeof(int64_t) * (size_t)k, hipMemcpyHostToDevice, ctx->stream));
        tp->all_gather(ctx->stream, d_meta.p, sizeof(int64_t) * (size_t)k, d_meta.p + k);
        std::vector<int64_t> all((size_t)k * (size_t)world);
        PMX_HIP(hipMemcpyAsync(all.data(), d_meta.p + k, sizeof(int64_t) * all.size(), hipMemcpyDeviceToHost, ctx->stream));
        PMX_HIP(hipStreamSynchronize(ctx->stream));
        return all;
    }
};

extern "C" {

int pmx_dist_unique_id(char id[PMX_DIST_ID_BYTES]) {
    if (!id) return PMX_ERR_ARG;
    memset(id, 0, PMX_DIST_ID_BYTES);
    if (pmx::opt_str(pmx::O_DIST_HOST_DIR)) {
        // the test transport needs no communicator, but its file names carry the id's first bytes as the run's nonce: a
        // directory that still holds the last-round files of an earlier run (they are never unlinked, see HostDirTransport)
        // can be used again
        FILE* f = fopen("/dev/urandom", "rb");
        const size_t got = f ? fread(id, 1, 16, f) : 0;
        if (f) fclose(f);
        if (got != 16) {
            const uint64_t a = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count(), b = (uint64_t)getpid();
            memcpy(id, &a, 8); memcpy(id + 8, &b, 8);
        }
        return PMX_OK;
    }
    PMX_TRY
    static_assert(sizeof(ncclUniqueId) == PMX_DIST_ID_BYTES, "unique id size");
    ncclUniqueId u;
    PMX_NCCL(rccl().GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return PMX_OK;
    PMX_CATCH
}

int pmx_dist_init(pmx_ctx* ctx, const char id[PMX_DIST_ID_BYTES], int rank, int world, pmx_dist** out) {
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<pmx_dist> d(new pmx_dist());
    d->ctx = ctx;
    if (const char* dir = pmx::opt_str(pmx::O_DIST_HOST_DIR)) {
        auto* t = new HostDirTransport();
        t->dir = dir;
        char hex[17];
        uint64_t n8;
        memcpy(&n8, id, sizeof(n8));
        snprintf(hex, sizeof(hex), "%016llx", (unsigned long long)n8);
        t->nonce = hex;
        d->tp.reset(t);
    } else {
        auto* t = new RcclTransport();
        d->tp.reset(t);
        ncclUniqueId u;
        memcpy(&u, id, sizeof(u));
        PMX_NCCL(rccl().CommInitRank(&t->comm, world, u, rank));
    }
    d->tp->rank = rank;
    d->tp->world = world;
    *out = d.release();
    return PMX_OK;
    PMX_CATCH
}

void pmx_dist_free(pmx_dist* d) {
    if (d && d->ctx) (void)hipSetDevice(d->ctx->device);
    if (d && d->ev_gathered) (void)hipEventDestroy(d->ev_gathered);
    if (d && d->ev_fetched) (void)hipEventDestroy(d->ev_fetched);
    delete d;
}

int pmx_dist_rank(const pmx_dist* d) { return d ? d->tp->rank : -1; }
int pmx_dist_world(const pmx_dist* d) { return d ? d->tp->world : 0; }

int pmx_dist_barrier(pmx_dist* d) {
    if (!d) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(d->ctx->device));
    const int64_t one = 1;
    (void)d->exchange_counts(&one, 1);
    return PMX_OK;
    PMX_CATCH
}

// element-wise sum of every rank's n int64 values, on every rank (--refine: a candidate's score is the sum over the shards)
int pmx_dist_sum_i64(pmx_dist* d, int64_t* vals, int64_t n) {
    if (!d || n < 0 || (n > 0 && !vals)) return PMX_ERR_ARG;
    PMX_TRY
    PMX_HIP(hipSetDevice(d->ctx->device));
    if (n == 0 || d->tp->world == 1) return PMX_OK;
    const std::vector<int64_t> all = d->exchange_counts(vals, (int)n);
    for (int64_t i = 0; i < n; ++i) {
        int64_t sum = 0;
        for (int r = 0; r < d->tp->world; ++r) sum += all[(size_t)r * (size_t)n + (size_t)i];
        vals[i] = sum;
    }
    return PMX_OK;
    PMX_CATCH
}

// --dedup over the whole sample: every distinct read string is counted once, by the lowest rank that holds a copy.  Each
// rank thins its own shard first (exact byte comparison), then the ranks exchange the 128-bit hash pairs of the reads they
// keep and a rank drops every read whose pair a LOWER rank keeps.  (Across ranks the identity of two reads is the equality of
// two independent 64-bit hashes of their ASCII: a false merge needs a 128-bit collision among the sample's reads.)
// Follow with pmx_place_add_reads(..., dedup_reads = 1) on the same read set: it seeds through the prepared mask.
int pmx_dist_dedup_reads(pmx_dist* d, pmx_place* pl, const pmx_readset* rs, int64_t* n_kept_local) {
    if (!d || !pl || !rs) return PMX_ERR_ARG;
    PMX_TRY
    pmx_ctx* ctx = d->ctx;
    PMX_HIP(hipSetDevice(ctx->device));
    const int world = d->tp->world, rank = d->tp->rank;
    const int64_t n = pmx_readset_num_reads(rs);
    DevBuf<uint64_t> mine;
    // sizes first: the padded all-gather needs the largest count
    mine.alloc(2 * (size_t)std::max<int64_t>(n, 1));
    int64_t kept = pmx_place_dedup_local(ctx, pl, rs, mine.p, mine.p + std::max<int64_t>(n, 1), std::max<int64_t>(n, 1));
    if (kept < 0) return (int)kept;
    if (world > 1) {
        const std::vector<int64_t> sizes = d->exchange_counts(&kept, 1);
        int64_t mx = 1, lower = 0;
        for (int r = 0; r < world; ++r) { mx = std::max(mx, sizes[r]); if (r < rank) lower += sizes[r]; }
        DevBuf<uint64_t> padded, all, seen;
        padded.alloc(2 * (size_t)mx);
        all.alloc(2 * (size_t)mx * (size_t)world);
        PMX_HIP(hipMemsetAsync(padded.p, 0, sizeof(uint64_t) * 2 * (size_t)mx, ctx->stream));
        if (kept > 0) {
            PMX_HIP(hipMemcpyAsync(padded.p, mine.p, sizeof(uint64_t) * (size_t)kept, hipMemcpyDeviceToDevice, ctx->stream));
            PMX_HIP(hipMemcpyAsync(padded.p + mx, mine.p + std::max<int64_t>(n, 1), sizeof(uint64_t) * (size_t)kept, hipMemcpyDeviceToDevice, ctx->stream));
        }
        d->tp->all_gather(ctx->stream, padded.p, sizeof(uint64_t) * 2 * (size_t)mx, all.p);
        if (lower > 0) {
            seen.alloc(2 * (size_t)lower);
            int64_t at = 0;
            for (int r = 0; r < rank; ++r) {
                if (sizes[r] == 0) continue;
                PMX_HIP(hipMemcpyAsync(seen.p + at, all.p + (size_t)r * 2 * (size_t)mx, sizeof(uint64_t) * (size_t)sizes[r], hipMemcpyDeviceToDevice, ctx->stream));
                PMX_HIP(hipMemcpyAsync(seen.p + lower + at, all.p + (size_t)r * 2 * (size_t)mx + (size_t)mx, sizeof(uint64_t) * (size_t)sizes[r], hipMemcpyDeviceToDevice, ctx->stream));
                at += sizes[r];
            }
            const int rc = pmx_place_dedup_drop_seen(ctx, pl, rs, seen.p, seen.p + lower, lower);
            if (rc != PMX_OK) return rc;
            kept = pmx_place_dedup_local_count(ctx, pl, rs);
            if (kept < 0) return (int)kept;
        }
        PMX_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (n_kept_local) *n_kept_local = kept;
    return PMX_OK;
    PMX_CATCH
}

// Exchange step 1: afterwards every rank's placer holds the histogram of the whole sample.
int pmx_dist_merge_histograms(pmx_dist* d, pmx_place* pl) {
    if (!d || !pl) return PMX_ERR_ARG;
    PMX_TRY
    pmx_ctx* ctx = d->ctx;
    PMX_HIP(hipSetDevice(ctx->device));
    const int world = d->tp->world, rank = d->tp->rank;
    if (world == 1) return PMX_OK;
    const int64_t n_loc = pmx_place_histogram_entries(ctx, pl);   // (no sort: the one sorted histogram is made after the merge)
    if (n_loc < 0) return (int)n_loc;
    const std::vector<int64_t> sizes = d->exchange_counts(&n_loc, 1);
    int64_t mx = 1;
    for (int64_t s : sizes) mx = std::max(mx, s);
    // max-padded (hash, count) planes per rank (RCCL has no all-gather-v): rank p's run sits 2 * mx elements after rank p-1's
    d->hist_mine.ensure(2 * (size_t)mx);
    d->hist_all.ensure(2 * (size_t)mx * (size_t)world);
    int rc = pmx_place_histogram_export_device_unsorted(ctx, pl, d->hist_mine.p, d->hist_mine.p + mx, mx);
    if (rc != PMX_OK) return rc;
    d->tp->all_gather(ctx->stream, d->hist_mine.p, 2 * (size_t)mx * sizeof(uint64_t), d->hist_all.p);
    return pmx_place_histogram_merge_device_parts(ctx, pl, d->hist_all.p, d->hist_all.p + mx, 2 * mx, sizes.data(), world, rank);
    PMX_CATCH
}

// Exchange step 2: the records and the CIGAR arena of the aligner's last call, from every rank to `root`.
int pmx_dist_gather_alignments(pmx_dist* d, pmx_aligner* al, int root, int64_t* n_records, int64_t* n_words) {
    if (!d || !al || root < 0 || root >= d->tp->world) return PMX_ERR_ARG;
    PMX_TRY
    pmx_ctx* ctx = d->ctx;
    PMX_HIP(hipSetDevice(ctx->device));
    const int world = d->tp->world, rank = d->tp->rank;
    if (d->fetch_pending) {   // a download of the previous gather on another stream still reads the buffers
        PMX_HIP(hipStreamWaitEvent(ctx->stream, d->ev_fetched, 0));
        d->fetch_pending = false;
    }
    const int64_t mine[2] = {pmx_align_num_records(al), pmx_align_cigar_words(ctx, al)};
    if (mine[1] < 0) return (int)mine[1];
    const std::vector<int64_t> all = d->exchange_counts(mine, 2);
    d->rank_records.assign((size_t)world, 0);
    d->rank_words.assign((size_t)world, 0);
    std::vector<size_t> rec_sizes((size_t)world), rec_offs((size_t)world), cig_sizes((size_t)world), cig_offs((size_t)world);
    int64_t tot_r = 0, tot_w = 0;
    for (int r = 0; r < world; ++r) {
        d->rank_records[r] = all[2 * r];
        d->rank_words[r] = all[2 * r + 1];
        rec_offs[r] = (size_t)tot_r * sizeof(pmx_aln_record);
        rec_sizes[r] = (size_t)all[2 * r] * sizeof(pmx_aln_record);
        cig_offs[r] = (size_t)tot_w * sizeof(uint32_t);
        cig_sizes[r] = (size_t)all[2 * r + 1] * sizeof(uint32_t);
        tot_r += all[2 * r];
        tot_w += all[2 * r + 1];
    }
    if (tot_w >= (int64_t)1 << 32) return fail(PMX_ERR_CAPACITY, "merged CIGAR arena exceeds the 32-bit cigar_off of pmx_aln_record");
    if (rank == root) {
        d->g_records.ensure((size_t)std::max<int64_t>(tot_r, 1));
        d->g_cigars.ensure((size_t)std::max<int64_t>(tot_w, 1));
    }
    d->tp->gather_v(ctx->stream, pmx_align_device_records(al), rec_sizes, rec_offs, d->g_records.p, root);
    d->tp->gather_v(ctx->stream, pmx_align_device_cigars(al), cig_sizes, cig_offs, d->g_cigars.p, root);
    if (rank == root) {
        int64_t r_at = 0, w_at = 0;
        for (int r = 0; r < world; ++r) {
            if (w_at > 0 && d->rank_records[r] > 0)
                hipLaunchKernelGGL(k_rebase_cigars, dim3(grid_for(d->rank_records[r], 256, ctx->n_cu * 8)), dim3(256), 0, ctx->stream,
                                   reinterpret_cast<pmx::aln::AlnRecord*>(d->g_records.p) + r_at, d->rank_records[r], (uint32_t)w_at);
            r_at += d->rank_records[r];
            w_at += d->rank_words[r];
        }
        PMX_HIP(hipGetLastError());
        d->g_n_records = tot_r;
        d->g_n_words = tot_w;
    } else {
        d->g_n_records = d->g_n_words = 0;
    }
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    if (n_records) *n_records = d->g_n_records;
    if (n_words) *n_words = d->g_n_words;
    return PMX_OK;
    PMX_CATCH
}

// Exchange step 2, one-node form: nothing crosses between the GPUs.  The ranks agree (one small all-gather of their record
// and CIGAR-word counts) on where every rank's part lies in ONE result set -- records in rank order, arenas back to back --
// each rank copies its records aside with cigar_off rebased onto the merged arena, and downloads ITS OWN part over ITS OWN
// PCIe link straight to its place in host buffers that all the ranks of the node map (a shared-memory segment).  The
// gather above sends everything through the root's device and the root's link: 360 MB per 10M reads at ~57 GB/s = 6.3 ms,
// a ceiling of ~1.6 G reads/s whatever the number of GPUs.
int pmx_dist_plan_alignments(pmx_dist* d, pmx_aligner* al, int64_t* record_base, int64_t* word_base, int64_t* total_records, int64_t* total_words) {
    if (!d || !al) return PMX_ERR_ARG;
    PMX_TRY
    pmx_ctx* ctx = d->ctx;
    PMX_HIP(hipSetDevice(ctx->device));
    const int world = d->tp->world, rank = d->tp->rank;
    if (d->fetch_pending) {   // a download of the previous plan on another stream still reads the staging copy
        PMX_HIP(hipStreamWaitEvent(ctx->stream, d->ev_fetched, 0));
        d->fetch_pending = false;
    }
    const int64_t mine[2] = {pmx_align_num_records(al), pmx_align_cigar_words(ctx, al)};
    if (mine[1] < 0) return (int)mine[1];
    const std::vector<int64_t> all = d->exchange_counts(mine, 2);
    d->rank_records.assign((size_t)world, 0);
    d->rank_words.assign((size_t)world, 0);
    int64_t tot_r = 0, tot_w = 0, my_r = 0, my_w = 0;
    for (int r = 0; r < world; ++r) {
        d->rank_records[r] = all[2 * r];
        d->rank_words[r] = all[2 * r + 1];
        if (r == rank) { my_r = tot_r; my_w = tot_w; }
        tot_r += all[2 * r];
        tot_w += all[2 * r + 1];
    }
    if (tot_w >= (int64_t)1 << 32) return fail(PMX_ERR_CAPACITY, "merged CIGAR arena exceeds the 32-bit cigar_off of pmx_aln_record");
    d->shard_records = mine[0]; d->shard_words = mine[1]; d->shard_record_base = my_r; d->shard_word_base = my_w;
    d->s_records.ensure((size_t)std::max<int64_t>(mine[0], 1));
    if (mine[0] > 0) {
        PMX_HIP(hipMemcpyAsync(d->s_records.p, pmx_align_device_records(al), sizeof(pmx_aln_record) * (size_t)mine[0], hipMemcpyDeviceToDevice, ctx->stream));
        if (my_w > 0)
            hipLaunchKernelGGL(k_rebase_cigars, dim3(grid_for(mine[0], 256, ctx->n_cu * 8)), dim3(256), 0, ctx->stream,
                               reinterpret_cast<pmx::aln::AlnRecord*>(d->s_records.p), mine[0], (uint32_t)my_w);
        PMX_HIP(hipGetLastError());
    }
    if (record_base) *record_base = my_r;
    if (word_base) *word_base = my_w;
    if (total_records) *total_records = tot_r;
    if (total_words) *total_words = tot_w;
    return PMX_OK;
    PMX_CATCH
}

// ... and the download of this rank's part to records_all + record_base / cigars_all + word_base (the WHOLE result set's
// buffers, e.g. a shared-memory mapping; pinned or registered memory for the copies to be asynchronous) on `stream`
// (NULL: the context's), without waiting for it.  The next plan waits for it before it reuses the staging copy.
int pmx_dist_fetch_shard_async(pmx_dist* d, pmx_aligner* al, pmx_aln_record* records_all, uint32_t* cigars_all, void* stream) {
    if (!d || !al || (d->shard_records > 0 && !records_all) || (d->shard_words > 0 && !cigars_all)) return PMX_ERR_ARG;
    PMX_TRY
    pmx_ctx* ctx = d->ctx;
    PMX_HIP(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    if (!d->ev_gathered) {
        PMX_HIP(hipEventCreateWithFlags(&d->ev_gathered, hipEventDisableTiming));
        PMX_HIP(hipEventCreateWithFlags(&d->ev_fetched, hipEventDisableTiming));
    }
    if (st != ctx->stream) {
        PMX_HIP(hipEventRecord(d->ev_gathered, ctx->stream));
        PMX_HIP(hipStreamWaitEvent(st, d->ev_gathered, 0));
    }
    if (d->shard_records > 0)
        PMX_HIP(hipMemcpyAsync(records_all + d->shard_record_base, d->s_records.p, sizeof(pmx_aln_record) * (size_t)d->shard_records, hipMemcpyDeviceToHost, st));
    if (d->shard_words > 0)
        PMX_HIP(hipMemcpyAsync(cigars_all + d->shard_word_base, pmx_align_device_cigars(al), sizeof(uint32_t) * (size_t)d->shard_words, hipMemcpyDeviceToHost, st));
    if (st != ctx->stream) {
        PMX_HIP(hipEventRecord(d->ev_fetched, st));
        d->fetch_pending = true;
    }
    return PMX_OK;
    PMX_CATCH
}

const void* pmx_dist_gathered_records(const pmx_dist* d) { return d ? d->g_records.p : nullptr; }
const void* pmx_dist_gathered_cigars(const pmx_dist* d) { return d ? d->g_cigars.p : nullptr; }

int pmx_dist_rank_counts(const pmx_dist* d, int64_t* records_per_rank, int64_t* words_per_rank) {
    if (!d) return PMX_ERR_ARG;
    for (size_t r = 0; r < d->rank_records.size(); ++r) {
        if (records_per_rank) records_per_rank[r] = d->rank_records[r];
        if (words_per_rank) words_per_rank[r] = d->rank_words[r];
    }
    return PMX_OK;
}

int pmx_dist_fetch_gathered(pmx_dist* d, pmx_aln_record* records, int64_t n_records, uint32_t* cigar_arena, int64_t arena_cap) {
    if (!d || n_records < d->g_n_records || arena_cap < d->g_n_words || (d->g_n_records > 0 && !records) || (d->g_n_words > 0 && !cigar_arena)) return PMX_ERR_ARG;
    PMX_TRY
    pmx_ctx* ctx = d->ctx;
    PMX_HIP(hipSetDevice(ctx->device));
    if (d->g_n_records > 0) PMX_HIP(hipMemcpyAsync(records, d->g_records.p, sizeof(pmx_aln_record) * (size_t)d->g_n_records, hipMemcpyDeviceToHost, ctx->stream));
    if (d->g_n_words > 0) PMX_HIP(hipMemcpyAsync(cigar_arena, d->g_cigars.p, sizeof(uint32_t) * (size_t)d->g_n_words, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
    PMX_CATCH
}

// the same download on a stream of the caller's choice, without waiting for it (see pmx_align_fetch_async)
int pmx_dist_fetch_gathered_async(pmx_dist* d, pmx_aln_record* records, int64_t n_records, uint32_t* cigar_arena, int64_t arena_cap, void* stream) {
    if (!d || n_records < d->g_n_records || arena_cap < d->g_n_words || (d->g_n_records > 0 && !records) || (d->g_n_words > 0 && !cigar_arena)) return PMX_ERR_ARG;
    PMX_TRY
    pmx_ctx* ctx = d->ctx;
    PMX_HIP(hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    if (!d->ev_gathered) {
        PMX_HIP(hipEventCreateWithFlags(&d->ev_gathered, hipEventDisableTiming));
        PMX_HIP(hipEventCreateWithFlags(&d->ev_fetched, hipEventDisableTiming));
    }
    if (st != ctx->stream) {
        PMX_HIP(hipEventRecord(d->ev_gathered, ctx->stream));
        PMX_HIP(hipStreamWaitEvent(st, d->ev_gathered, 0));
    }
    if (d->g_n_records > 0) PMX_HIP(hipMemcpyAsync(records, d->g_records.p, sizeof(pmx_aln_record) * (size_t)d->g_n_records, hipMemcpyDeviceToHost, st));
    if (d->g_n_words > 0) PMX_HIP(hipMemcpyAsync(cigar_arena, d->g_cigars.p, sizeof(uint32_t) * (size_t)d->g_n_words, hipMemcpyDeviceToHost, st));
    if (st != ctx->stream) {
        PMX_HIP(hipEventRecord(d->ev_fetched, st));
        d->fetch_pending = true;
    }
    return PMX_OK;
    PMX_CATCH
}

}  // extern "C"
