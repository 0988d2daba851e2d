// ALIGN stage: shared types of the per-read(-pair) mapping pipeline.
//
// The pipeline restates, for one wave64 per read pair, what the reference's aligner boundary does per
// pair (src/mm_align.c:304-354 -> mm_map_frag, src/3rdparty/minimap2/map.c:236-390): sketch -> seed
// lookup -> chaining DP -> region bookkeeping -> ksw2 dual-affine extension -> mapq -> pairing.
// All code in align/*.hpp is PMX_HD: it is compiled by hipcc for gfx950 (PMX_W = 64 lanes of one wave
// execute it; scalar bookkeeping runs redundantly-uniform on every lane, the DP anti-diagonals are
// spread over the lanes) and by g++ with PMX_W = 1 for the CPU unit tests of the host logic
// (tests/hostsim; never part of libpanmap_amd.so).
#pragma once
#include <stdint.h>

#include <type_traits>

#include "../device/pmx_math.h"

// PMX_W = lanes that cooperate on ONE read pair: 64 in the wave-per-pair kernels, 1 on the host and in the
// thread-per-pair kernel (PMX_THREAD_PER_PAIR: every lane runs the whole pipeline for its own pair).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PMX_THREAD_PER_PAIR)
#define PMX_W 64
#else
#define PMX_W 1
#endif

namespace pmx {
namespace aln {

#if PMX_W > 1
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ void wave_sync() { __syncthreads(); }   // one wave per workgroup
#else
PMX_HD int lane_id() { return 0; }
PMX_HD void wave_sync() {}
#endif

// Tier-1 translation unit (PMX_ALL_LDS): every work array except the traceback matrix is in LDS, and the
// compiler is told so pointer by pointer -- otherwise generic pointers compile to flat_load/flat_store,
// whose latency dominated the kernel (measured: ~9k flat ops per pair, 56% of wave cycles waiting).
// the same statement about ONE pointer in any device translation unit
#if defined(__HIP_DEVICE_COMPILE__)
#define PMX_LDS_HERE(p) __builtin_assume(__builtin_amdgcn_is_shared((const void*)(p)))
#else
#define PMX_LDS_HERE(p) ((void)0)
#endif
#if defined(__HIP_DEVICE_COMPILE__) && defined(PMX_ALL_LDS)
#define PMX_LDS(p) __builtin_assume(__builtin_amdgcn_is_shared((const void*)(p)))
#elif defined(__HIP_DEVICE_COMPILE__) && defined(PMX_THREAD_PER_PAIR)
// Thread-per-pair kernel: the Work descriptor is a local object of the kernel (private memory) that every function of the
// pipeline receives by reference, i.e. through a generic pointer -- 1,400 flat loads and 950 flat stores in the kernel, most
// of them fields of W.  The statement every function already makes about W (PMX_LDS(&W): "in LDS" in the all-LDS tier)
// says "private" here; everything else the macro is applied to is an IPtr (an arena offset, not a pointer) or a pointer
// into global memory, and gets no statement.
struct Work;
__device__ __forceinline__ void pmx_private_hint(const Work* w) { __builtin_assume(__builtin_amdgcn_is_private((const void*)w)); }
__device__ __forceinline__ void pmx_private_hint(Work* w) { __builtin_assume(__builtin_amdgcn_is_private((const void*)w)); }
template <class T> __device__ __forceinline__ void pmx_private_hint(const T&) {}
#define PMX_LDS(p) (pmx_private_hint(p))
#else
#define PMX_LDS(p) ((void)0)
#endif

// Traceback bytes of the DP kernels.  A store through a generic pointer compiles to flat_store, which counts on
// lgkmcnt as well as vmcnt: the next LDS access of the anti-diagonal loop then waits for the HBM round trip of
// the previous diagonal's store (measured: ~5k cycles per diagonal in every DP kernel).  The wave-per-pair DP
// kernels therefore pick the address space explicitly (uniform branch): ds_write_b8 or global_store_byte.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PMX_THREAD_PER_PAIR)
typedef __attribute__((address_space(1))) uint8_t pmx_gbyte;
typedef __attribute__((address_space(3))) uint8_t pmx_lbyte;
#define PMX_TB_STORE(p, is_lds, idx, val)                         \
    do {                                                         \
        if (is_lds) ((pmx_lbyte*)(p))[(idx)] = (val);            \
        else ((pmx_gbyte*)(p))[(idx)] = (val);                   \
    } while (0)
#define PMX_TB_IS_LDS(p) __builtin_amdgcn_is_shared((const void*)(p))
#else
#define PMX_TB_STORE(p, is_lds, idx, val) ((p)[(idx)] = (val))
#define PMX_TB_IS_LDS(p) false
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define PMX_STAMP(W, k)                                                              \
    do {                                                                            \
        if ((W).prof) {                                                             \
            const unsigned long long t_ = (unsigned long long)clock64();            \
            (W).prof_acc[k] += t_ - (W).prof_t;                                     \
            (W).prof_t = t_;                                                        \
        }                                                                           \
    } while (0)
#else
#define PMX_STAMP(W, k) ((void)0)
#endif

struct A128 {
    uint64_t x, y;
};

// ---------------------------------------------------------------------------------------- work pointers
// Ptr<T> is how every per-pair work array is addressed.  In the wave-per-pair kernels and on the host it is
// a plain T*.  In the thread-per-pair kernel it is IPtr<T>: a 32-bit LOGICAL byte offset into the thread's
// work arena, translated on access to a granule-interleaved address in the wave's HBM slab
//     physical = wave_base + (offset / G) * (64 * G) + lane * G + offset % G
// so the 64 lanes of a wave touching the same logical offset share cache lines instead of owning one line
// each (measured with per-thread contiguous slabs: 4450 line fills per pair, ~4.2 TB/s of HBM traffic; the
// kernel was bound by exactly that).  The granule G follows the element type: 16 bytes for 8/16-byte
// elements (A128, uint64), 4 bytes for byte / 32-bit arrays (a wave-wide byte access then spans 256 B
// instead of 1 KB).  Either way a 16-byte-aligned logical block maps into the same 1 KB physical block,
// so regions of different granule never collide as long as every region is 16-byte aligned and sized and
// is always accessed through one granule -- the thread-per-pair layout therefore has NO overlaid regions,
// and ptr_cast only converts between types of equal granule.  Objects with natural alignment never
// straddle a granule, so operator[] can hand out real references; larger structs (Reg, Seed) stay in a
// small per-thread contiguous region behind plain pointers.
// The thread-per-pair arena of a launch: FIRST member of the kernel argument struct (AlignArgs), so that IPtr::phys
// can read it from the kernarg segment at a fixed offset (scalar loads, hoisted like a __constant__ would be).  It is
// per launch, not per module: two host threads aligning on one device never see each other's arena (the aligner
// boundary is called concurrently from TBB workers in --batch mode, src/main.cpp:1581-1611).
struct TppArena {
    uint8_t* base;          // first wave's slab
    uint32_t wave_stride;   // bytes per wave slab = 64 * per-thread logical arena size
    uint32_t pad;
};
#if defined(PMX_THREAD_PER_PAIR) && defined(__HIP_DEVICE_COMPILE__)
#define PMX_INTERLEAVED 1
// interleave granule of an element type (log2 bytes); a type may opt into a 64-byte granule (Wide64)
template <class T, class = void> struct IGranule { static constexpr uint32_t LG = sizeof(T) >= 8 ? 4u : 2u; };
template <class T> struct IGranule<T, typename std::enable_if<std::remove_cv<T>::type::kWide64>::type> { static constexpr uint32_t LG = 6u; };
template <class T>
struct IPtr {
    uint32_t o;
    IPtr() = default;
    __device__ explicit IPtr(uint32_t off) : o(off) {}
    template <class U, class = typename std::enable_if<std::is_convertible<U*, T*>::value>::type>
    __device__ IPtr(const IPtr<U>& q) : o(q.o) {}
    static constexpr uint32_t LG = IGranule<T>::LG;   // log2 of the granule
    __device__ __forceinline__ static T* phys(uint32_t off) {
        // The kernarg segment of this dispatch, found through the AQL dispatch packet (hsa_kernel_dispatch_packet_t:
        // kernarg_address at byte 40).  llvm.amdgcn.kernarg.segment.ptr itself is only defined in the kernel function, the
        // dispatch pointer is handed down to callees.  Both loads are uniform and from the constant address space.
        typedef const __attribute__((address_space(4))) uint8_t* CPtr;
        typedef const __attribute__((address_space(4))) TppArena* KernargArena;
        const CPtr pkt = (CPtr)__builtin_amdgcn_dispatch_ptr();
        const KernargArena ka = *(const KernargArena __attribute__((address_space(4)))*)(pkt + 40);   // AlignArgs::tpp
        // (the arena is global memory, and the address is built in that address space: handed out as a generic pointer the
        //  optimizer still sees where it came from and emits global_load / global_store -- through a plain generic pointer
        //  every arena access was a flat instruction, 2,400 of them in the kernel, each counted on the LDS queue as well)
        typedef __attribute__((address_space(1))) uint8_t GByte;
        GByte* wave_base = (GByte*)ka->base + (size_t)blockIdx.x * ka->wave_stride;   // uniform
        const uint32_t vo = ((threadIdx.x & 63u) << LG) + ((off >> LG) << (LG + 6)) + (off & ((1u << LG) - 1u));
        return reinterpret_cast<T*>((uint8_t*)(wave_base + vo));
    }
    __device__ __forceinline__ T& operator*() const { return *phys(o); }
    __device__ __forceinline__ T* operator->() const { return phys(o); }
    template <class I> __device__ __forceinline__ T& operator[](I i) const { return *phys(o + (uint32_t)i * (uint32_t)sizeof(T)); }
    template <class I> __device__ __forceinline__ IPtr operator+(I n) const { return IPtr(o + (uint32_t)n * (uint32_t)sizeof(T)); }
    template <class I> __device__ __forceinline__ IPtr operator-(I n) const { return IPtr(o - (uint32_t)n * (uint32_t)sizeof(T)); }
    __device__ __forceinline__ int64_t operator-(const IPtr& q) const { return (int64_t)((int32_t)(o - q.o) / (int32_t)sizeof(T)); }
    template <class I> __device__ __forceinline__ IPtr& operator+=(I n) { o += (uint32_t)n * (uint32_t)sizeof(T); return *this; }
    template <class I> __device__ __forceinline__ IPtr& operator-=(I n) { o -= (uint32_t)n * (uint32_t)sizeof(T); return *this; }
    __device__ __forceinline__ IPtr& operator++() { o += (uint32_t)sizeof(T); return *this; }
    __device__ __forceinline__ IPtr& operator--() { o -= (uint32_t)sizeof(T); return *this; }
    __device__ __forceinline__ bool operator==(const IPtr& q) const { return o == q.o; }
    __device__ __forceinline__ bool operator!=(const IPtr& q) const { return o != q.o; }
    __device__ __forceinline__ bool operator<(const IPtr& q) const { return o < q.o; }
    __device__ __forceinline__ bool operator>(const IPtr& q) const { return o > q.o; }
    __device__ __forceinline__ bool operator<=(const IPtr& q) const { return o <= q.o; }
    __device__ __forceinline__ bool operator>=(const IPtr& q) const { return o >= q.o; }
};
template <class T> using Ptr = IPtr<T>;
template <class U, class T> __device__ __forceinline__ IPtr<U> ptr_cast(IPtr<T> q) {
    static_assert(IPtr<U>::LG == IPtr<T>::LG, "ptr_cast between element types of different interleave granule");
    return IPtr<U>(q.o);
}
// A whole region viewed through another granule (e.g. 64-byte blocks of four chain cells: one lane's block is
// one contiguous 64 bytes, so a batch of four cells costs one line instead of four).  Every 64-byte aligned
// logical block maps into the same 4 KB of the wave slab under any granule, so this is safe as long as the region
// starts 64-byte aligned (plan_layout_tpp aligns every region) and is accessed ONLY through the new view until
// it is handed back (nothing cached is carried across).
template <class U, class T> __device__ __forceinline__ IPtr<U> ptr_region_cast(IPtr<T> q) { return IPtr<U>(q.o); }
#else
template <class T> using Ptr = T*;
template <class U, class T> PMX_HD U* ptr_cast(T* q) { return reinterpret_cast<U*>(q); }
template <class U, class T> PMX_HD U* ptr_region_cast(T* q) { return reinterpret_cast<U*>(q); }
#endif

// Sequential byte reads through ALIGNED 32-bit loads: the per-base loops of the pipeline (sketch, mismatch
// scans, z-drop test, alignment statistics, reference copies) issue one memory access per four bases instead
// of one per base.  Works on either pointer flavour: the aligned word that contains byte i is loaded once
// and kept until the scan leaves it (arrays are padded, so the word never leaves the allocation).
#ifdef PMX_INTERLEAVED
struct ByteReader {
    uint32_t base, tag, w;
    __device__ __forceinline__ explicit ByteReader(IPtr<const uint8_t> q) : base(q.o), tag(0xffffffffu), w(0) {}
    __device__ __forceinline__ uint32_t operator[](int i) {
        const uint32_t o = base + (uint32_t)i;
        if ((o >> 2) != tag) { tag = o >> 2; w = *IPtr<const uint32_t>::phys(o & ~3u); }
        return (w >> ((o & 3u) * 8u)) & 0xffu;
    }
};
#endif
struct RawByteReader {   // wave-per-pair kernels / host: plain byte reads (lanes stride the loops, LDS pointers stay LDS)
    const uint8_t* base;
    PMX_HD explicit RawByteReader(const uint8_t* q) : base(q) {}
    PMX_HD uint32_t operator[](int i) const { return base[i]; }
};
struct GlobalByteReader {   // reference sequence in global memory, any mode
    const uint8_t* base;
    uintptr_t tag;
    uint32_t w;
    PMX_HD explicit GlobalByteReader(const uint8_t* q) : base(q), tag(~(uintptr_t)0), w(0) {}
    PMX_HD uint32_t operator[](int i) {
        const uintptr_t a = (uintptr_t)(base + i);
        if ((a >> 2) != tag) { tag = a >> 2; w = *reinterpret_cast<const uint32_t*>(a & ~(uintptr_t)3); }
        return (w >> ((a & 3u) * 8u)) & 0xffu;
    }
};
#ifndef PMX_INTERLEAVED
typedef RawByteReader ByteReader;
#endif

// anchor flag bits in A128::y (mmpriv.h:17-23)
#define PMX_SEED_LONG_JOIN (1ULL << 40)
#define PMX_SEED_IGNORE (1ULL << 41)
#define PMX_SEED_TANDEM (1ULL << 42)
#define PMX_SEED_SELF (1ULL << 43)
#define PMX_SEED_SEG_SHIFT 48
#define PMX_SEED_SEG_MASK (0xffULL << PMX_SEED_SEG_SHIFT)

#define PMX_PARENT_UNSET (-1)
#define PMX_PARENT_TMP_PRI (-2)

// ksw2 flags (ksw2.h:8-19)
#define PMX_EZ_SCORE_ONLY 0x01
#define PMX_EZ_RIGHT 0x02
#define PMX_EZ_APPROX_MAX 0x08
#define PMX_EZ_APPROX_DROP 0x10
#define PMX_EZ_EXTZ_ONLY 0x40
#define PMX_EZ_REV_CIGAR 0x80
#define PMX_KSW_NEG_INF (-0x40000000)
// the small LDS copy of the DP arrays of the wave-per-read kernels (Work::dp_fast): DPs up to this size; a compile-time
// constant so that the nine arrays are immediate offsets from one base address in the inner loop
#define PMX_DP_FAST_TLEN 608

// Mapping options actually read on this path: the option block of setup_minimap2(for_scoring=1)
// (src/mm_align.c:118-188) on top of mm_mapopt_init (options.c:14-64) and mm_mapopt_update (:66-81).
struct Opt {
    int k, w;                       // minimizer index parameters
    int is_sr_like;                 // mean read length < 500 branch (FRAG_MODE | HEAP_SORT)
    int seed;
    int bw, bw_long, max_gap, max_gap_ref, max_frag_len, max_chain_skip, max_chain_iter, min_cnt, min_chain_score;
    float chain_gap_scale, chain_skip_scale;
    float chn_pen_gap, chn_pen_skip;  // chain_gap_scale * 0.01 * k, evaluated on the host (map.c:282-283)
    float mask_level;
    int mask_len;
    float pri_ratio;
    int best_n;
    float alt_drop;
    int a, b, q, e, q2, e2, sc_ambi, zdrop, zdrop_inv, end_bonus, min_dp_max, min_ksw_len;
    float max_clip_ratio;
    int rank_min_len;
    float rank_frac;
    int pe_ori, pe_bonus;
    int mid_occ, max_occ, max_max_occ, occ_dist;
    float q_occ_frac;
    int64_t max_sw_mat;
    int rmq_rescue_size, rmq_inner_dist, rmq_size_cap;
    float rmq_rescue_ratio;
    int8_t mat[25];
    int ref_len;
};

// Device-resident minimizer index of ONE reference sequence (replaces mm_idx_str / mm_idx_get,
// index.c:81-99, 408-451): open-addressing table keyed by the minimizer value, occurrence lists sorted
// ascending (index.c:252 sorts by position).
struct HtEnt {              // one probe = one 16-byte load
    uint64_t key;           // minimizer (x >> 8); UINT64_MAX = empty
    uint32_t off, cnt;      // first occurrence in pos[], number of occurrences
};
struct RefIndex {
    const uint8_t* seq;     // nt4 codes 0..4, ref_len bytes
    int32_t len;
    uint32_t ht_mask;       // table size - 1
    const HtEnt* ht;
    const uint64_t* pos;    // y values: rid<<32 | lastPos<<1 | strand
    // logf tables computed by the host libm (hit.c:440-457, pe.c:160): bit-identical mapq without
    // relying on a device logf.   logf_ratio[d] = logf((float)d / a) ; logf_int[n] = logf((float)n)
    const float* logf_ratio;
    const float* logf_int;
    int32_t n_logf;
    // the same sequence 2 bit/base (base j of word k at bits 2(j % 32); ambiguous bases as 0) and its ambiguity mask
    // (bit 2(j % 32) of word k set when base j is not A/C/G/T): what the compact tier XORs packed reads against
    const uint64_t* pk;
    const uint64_t* pk_amb;
    // per table slot: the low 32 bits of the ONLY occurrence's position word (cnt == 1), so that a probe of a unique
    // minimizer needs no second, dependent load from pos[]
    const uint32_t* ht_pv;
};

// chain-DP state of one anchor (f, p, t, v of mg_lchain_dp, lchain.c:148-230) as ONE 16-byte cell: the inner
// loop needs f, p and t of anchor j together -- one load instead of three
struct ChainCell {
    int32_t f, p, t, v;
};

// mm_seed_t (mmpriv.h:42-49) as two 16-byte halves kept in parallel arrays (one interleave granule each in the
// thread-per-pair arena): the merge loop reads both, the occurrence filter and the list staging only the first
struct SeedA {
    uint32_t n, q_pos, off, flt;
};
struct SeedB {
    uint32_t q_span, seg_id, is_tandem, pad;
};

// mm_reg1_t + mm_extra_t (minimap.h:98-128).  In the thread-per-pair kernel the struct is STRIDED: every 4-byte
// slot is followed by 252 bytes of padding, so that with a base pointer of `wave region + lane * 4` the fields of
// the 64 lanes' regions interleave dword by dword exactly like the IPtr arrays (a wave touching the same field of
// its 64 regions touches 256 contiguous bytes instead of 64 separate lines), while the code keeps plain struct
// syntax.  Copies and clears are field-wise (the padding belongs to the other lanes).
#define PMX_REG_SLOTS 27
#define PMX_REG_STRIDED_BYTES (PMX_REG_SLOTS * 256)
#ifdef PMX_INTERLEAVED
#define PMX_RP(n) char pad_##n[252];
#else
#define PMX_RP(n)
#endif
#define PMX_REG_FIELDS(X)                                                                                              \
    X(id) X(cnt) X(rid) X(score) X(qs) X(qe) X(rs) X(re) X(parent) X(subsc) X(as) X(mlen) X(blen) X(n_sub) X(score0)   \
    X(hash) X(div) X(mapq) X(split) X(rev) X(inv) X(sam_pri) X(proper_frag) X(pe_thru) X(seg_split) X(seg_id)          \
    X(split_inv) X(is_alt) X(strand_retained) X(has_p) X(dp_score) X(dp_max) X(dp_max2) X(n_ambi) X(n_cigar) X(cig_slot)
struct Reg {
    int32_t id; PMX_RP(0) int32_t cnt; PMX_RP(1) int32_t rid; PMX_RP(2) int32_t score; PMX_RP(3) int32_t qs; PMX_RP(4)
    int32_t qe; PMX_RP(5) int32_t rs; PMX_RP(6) int32_t re; PMX_RP(7) int32_t parent; PMX_RP(8) int32_t subsc; PMX_RP(9)
    int32_t as; PMX_RP(10) int32_t mlen; PMX_RP(11) int32_t blen; PMX_RP(12) int32_t n_sub; PMX_RP(13) int32_t score0; PMX_RP(14)
    uint32_t hash; PMX_RP(15)
    float div; PMX_RP(16)
    uint8_t mapq, split, rev, inv; PMX_RP(17)
    uint8_t sam_pri, proper_frag, pe_thru, seg_split; PMX_RP(18)
    uint8_t seg_id, split_inv, is_alt, strand_retained; PMX_RP(19)
    // extra
    uint8_t has_p, pad_b[3]; PMX_RP(20)
    int32_t dp_score; PMX_RP(21) int32_t dp_max; PMX_RP(22) int32_t dp_max2; PMX_RP(23)
    uint32_t n_ambi; PMX_RP(24)
    uint32_t n_cigar; PMX_RP(25)
    uint32_t cig_slot; PMX_RP(26)   // index of this region's CIGAR buffer in the per-wave CIGAR pool
#ifdef PMX_INTERLEAVED
    Reg() = default;
#define PMX_X(f) f = o.f;
    __device__ __forceinline__ Reg(const Reg& o) { PMX_REG_FIELDS(PMX_X) }
    __device__ __forceinline__ Reg& operator=(const Reg& o) { PMX_REG_FIELDS(PMX_X) return *this; }
#undef PMX_X
#endif
};
#ifdef PMX_INTERLEAVED
static_assert(sizeof(Reg) == PMX_REG_STRIDED_BYTES, "strided Reg layout");
#else
static_assert(sizeof(Reg) == 4 * PMX_REG_SLOTS, "Reg has 27 four-byte slots");
#endif

struct Ez {             // ksw_extz_t (ksw2.h:27-36)
    uint32_t max;
    int zdropped;
    int max_q, max_t, mqe, mqe_t, mte, mte_q, score, n_cigar, reach_end;
};

// status bits reported per record
#define PMX_ST_OVERFLOW 0x1
#define PMX_ST_UNSUPPORTED 0x2
#define PMX_ST_NEED_WAVE 0x4   // thread-per-pair kernel: hand this pair to the wave-per-pair kernels
#define PMX_ST_NEED_DP 0x8     // thread-per-pair kernel: a DP request was posted; re-run once it is served
#define PMX_ST_ABORT (PMX_ST_NEED_WAVE | PMX_ST_NEED_DP)

// DP service of the thread-per-pair kernel (align_kernel_tpp.hip): a pair whose extension / gap fill is
// not covered by ksw_shortcut posts the DP's inputs as a request, a wave-per-request kernel runs
// ksw_extd2 on it, and the pair is replayed with the result served from its per-pair result list (the
// pipeline is deterministic, so the c-th DP call of the replay is the c-th call of the first run).
#define PMX_DP_SEQ_BYTES 480
#define PMX_DP_MAX_CIGAR 20
#define PMX_DP_MAX_CALLS 8
#define PMX_DP_REQ_PER_PASS 4   // DP requests a pair may post in one thread-per-pair pass (its request slots)
struct DpRes {   // 128 bytes
    Ez ez;
    uint32_t key;                    // must equal the request's key; 0xffffffff = the DP overflowed
    uint32_t cigar[PMX_DP_MAX_CIGAR];
};

struct DpReq {   // 512 bytes
    int32_t qlen, tlen, w, zdrop, end_bonus, flag;
    uint32_t call, key;   // call: index of the DP call within the pair (its place in the result list); 0xffffffff = no request in this entry
    uint8_t seq[PMX_DP_SEQ_BYTES];   // query, then target at ((qlen + 15) & ~15)
};

// Capacities of the per-wave work memory (chosen by the host from the read-length regime).
struct Caps {
    int max_qlen;      // per segment
    int max_mini;      // minimizers per fragment
    int max_anchor;    // anchors per fragment
    int max_reg;       // regions per list
    int max_cigar;     // CIGAR ops per region
    int max_tlen;      // DP target length
    int n_cig_slots;   // CIGAR buffers in the pool
    int dp_fast_tlen;  // wave-per-read kernels, long reads: DPs up to this size run on the small LDS copy of the DP arrays (0 = none)
};

// Per-wave work memory: plain pointers into the LDS arena ("fast") or the global scratch slab ("slow");
// which array lives where is decided by the host planner (align_layout()).
struct Work {
    Caps caps;
    // sequences
    Ptr<uint8_t> qseq[2][2];   // [segment][strand] nt4 codes; strand 1 = reverse complement
    int qlen[2];
    int n_segs;
    // sketch / seeds
    Ptr<A128> mv;
    int n_mv;
    Ptr<A128> sk_buf;      // minimizer window ring (w entries)
    // thread-per-pair kernel: the ring lives in LDS instead, [slot][lane], 8-byte x and 4-byte low half of y
    // (the high half of y is the segment id); NULL elsewhere
    uint64_t* sk_lds_x;
    uint32_t* sk_lds_y;
    Ptr<SeedA> seeds;
    Ptr<SeedB> seeds_b;
    int n_seeds;
    Ptr<uint64_t> mini_pos;
    int n_mini_pos;
    Ptr<A128> heap;
    // anchors + chaining
    Ptr<A128> a;
    Ptr<A128> a2;
    int64_t n_a;
    Ptr<ChainCell> cc;     // chain-DP cells (max_anchor)
    Ptr<int32_t> kidx;     // filter_bad_seeds: indices of long gaps (max_anchor)
    Ptr<A128> z;
    Ptr<uint64_t> u;
    Ptr<uint64_t> u2;
    int n_u;
    // regions
    Reg* regs0;
    Reg* regs[2];
    Reg* reg_tmp;
    int n_regs0, n_regs[2];
    Ptr<A128> seg_a[2];
    Ptr<uint64_t> seg_u[2];
    int seg_n_a[2], seg_n_u[2];
    Ptr<uint64_t> aux64;   // small sort scratch (max_reg * 8)
    Ptr<int32_t> aux32;    // small index scratch (max_reg * 4)
    Ptr<A128> aux128;
    // DP
    Ptr<int8_t> du, dv, dx, dy, dx2, dy2, ds;
    Ptr<uint8_t> sf, qr;
    Ptr<int32_t> H;
    Ptr<int32_t> off, off_end;
    Ptr<uint8_t> tb;       // traceback matrix (global)
    int8_t* dp_fast;       // u,v,x,y,x2,y2,s,sf,qr for DPs of at most caps.dp_fast_tlen (LDS; NULL = none)
    size_t tb_cap;
    Ptr<uint8_t> tseq;
    Ptr<uint32_t> cig_tmp;   // ez->cigar
    Ptr<uint32_t> cig_pool;  // n_cig_slots * max_cigar
    int cig_next;
    uint32_t status;
    int rep_len;
    int frag_gap;
    uint64_t tmp64;        // lane-0 -> wave broadcast slot
    // DP service (thread-per-pair kernel only; all NULL / 0 elsewhere)
    uint8_t* dp_req_base;             // request slots, sizeof(DpReq) each
    const struct DpRes* dp_res;       // this pair's served results (dp_n_cached of them)
    unsigned long long* dp_slot_ctr;  // round 0: slot allocator
    int64_t dp_slot;                  // this pair's slot, -1 = none yet
    uint32_t dp_slot_cap;
    int dp_n_cached, dp_calls;
    int dp_post_end;        // calls [dp_n_cached, dp_post_end) were posted in this pass (contiguous: the next pass may rely on them)
    uint32_t status_pre;    // status when the first request of this pass was posted (what follows runs on neutral dummy results)
    int last_dp_shortcut;   // the last align_pair call was answered by ksw_shortcut
    int skip_shortcut;      // align1 already tried the shortcut on the reference bases directly
    // optional phase profile (diagnostic runs only: AlignArgs::prof != NULL)
    unsigned long long* prof;
    unsigned long long prof_t;
    unsigned long long prof_acc[24];
    // DP work actually run for this pair by the wave models / host (statistics only: GCUPS in bench.py)
    uint32_t dp_run_calls;
    uint64_t dp_run_cells;
    int sk_no_lane_ring;   // wave models: 1 = sketch with the ring in LDS (comparison / fallback switch)
    int no_rows_dp;        // wave-per-read kernels: 1 = never the row-by-row DP (aln_ksw_rows.hpp; comparison switch)
    int mv_ready;          // mv[] / n_mv already hold this pair's minimizers (handed over by the thread-per-pair kernel)   // [0..11] phases, [16..23] sub-phases of seeding / chaining / align1 (PMX_ALIGN_PROF)
};

}  // namespace aln
}  // namespace pmx
