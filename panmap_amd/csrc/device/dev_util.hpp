// Host-side helpers for the device translation units: error plumbing, RAII device buffers, context.
#pragma once
#include "pmx_options.hpp"
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "../api_internal.hpp"

namespace pmx {

struct HipError {
    std::string msg;
};

#define PMX_HIP(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess)                                                                              \
            throw pmx::HipError{std::string(#expr) + ": " + hipGetErrorString(_e)};                        \
    } while (0)

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    bool owned = true;
    DevBuf() {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p && owned) (void)hipFree(p);
        p = nullptr;
        n = 0;
        owned = true;
    }
    void alloc(size_t count) {
        release();
        if (count == 0) count = 1;
        PMX_HIP(hipMalloc((void**)&p, count * sizeof(T)));
        n = count;
    }
    void ensure(size_t count) {
        if (count > n) alloc(count);
    }
    void wrap(T* ptr, size_t count) {
        release();
        p = ptr;
        n = count;
        owned = false;
    }
    void swap(DevBuf& o) {
        std::swap(p, o.p);
        std::swap(n, o.n);
        std::swap(owned, o.owned);
    }
};

// A stream with a HARDWARE queue of its own.  HIP multiplexes ordinary streams onto a few hardware queues
// (GPU_MAX_HW_QUEUES, 4 by default); streams that land on one queue execute strictly in submission order: kernels of two
// contexts then never overlap, and a kernel queued behind the barrier packet of a long host-to-device copy of ANOTHER stream
// waits for that copy.  A stream created with a CU mask gets its own queue; the mask enables every CU.
// (PMX_CTX_POOLED_QUEUE=1: ordinary pooled streams.)
inline hipStream_t create_dedicated_stream(int n_cu) {
    hipStream_t st = nullptr;
    if (!pmx::opt_str(pmx::O_CTX_POOLED_QUEUE) && n_cu > 0) {
        std::vector<uint32_t> mask((size_t)(n_cu + 31) / 32, 0xffffffffu);
        if (n_cu % 32) mask.back() = (1u << (n_cu % 32)) - 1u;
        if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
            (void)hipGetLastError();
            st = nullptr;
        }
    }
    if (!st) PMX_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    return st;
}

struct KernelTimer {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int launches = 0;
    bool pending = false;
};

}  // namespace pmx

struct pmx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::map<std::string, pmx::KernelTimer> timers;
    int n_cu = 256;
    // Side streams of the seeding stage (hardware queues of their own, see create_dedicated_stream) and the events that order
    // them against the context's stream.  They belong to the CONTEXT, made on first use and kept until it is destroyed: a
    // placer that made and destroyed its own left the next placer's kernels on a recycled queue hanging (round 4).
    hipStream_t pair_stream = nullptr;       // pmx_readset_order_pairs
    hipStream_t seed_streams[3] = {nullptr, nullptr, nullptr};
    hipEvent_t seed_go = nullptr, seed_done[3] = {nullptr, nullptr, nullptr};
    // align stage: the general tiers of the pairs the compact tier's seeds kernel gave up on run on pair_stream beside the
    // compact chain kernel (api_align.hip); these two events order that stream against the context's
    hipEvent_t tail_go = nullptr, tail_done = nullptr;
};

namespace pmx {
inline void timer_begin(pmx_ctx* ctx, const char* name) {
    KernelTimer& t = ctx->timers[name];
    if (!t.e0) { PMX_HIP(hipEventCreate(&t.e0)); PMX_HIP(hipEventCreate(&t.e1)); }
    PMX_HIP(hipEventRecord(t.e0, ctx->stream));
}
inline void timer_end(pmx_ctx* ctx, const char* name, int launches) {
    KernelTimer& t = ctx->timers[name];
    PMX_HIP(hipEventRecord(t.e1, ctx->stream));
    t.launches = launches;
    t.pending = true;
}
inline int grid_for(int64_t n, int block, int max_blocks) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}
}  // namespace pmx
