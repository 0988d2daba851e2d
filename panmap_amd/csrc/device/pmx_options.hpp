// Every tuning / testing / diagnostic switch of the library, in ONE table.
//
// The switches are environment variables PMX_<NAME>.  The environment is read ONCE, at the first use, into a table
// (pmx::opt_str looks a switch up there: nullptr when it is not set, its value otherwise -- what getenv would have returned
// at that moment); pmx_options_reload() reads it again (tests that change a switch inside one process call it through
// panmap_amd.reload_options()), pmx_options_describe() lists the table with the current values.  Nothing else in the library
// calls getenv.  What ships is the behaviour with NO switch set; a switch exists for one of these reasons (its class):
//   env   deployment: which device, how many host threads, the test transport's directory
//   test  forces a rare path so that a test can reach it (capacity overflows, redo loops, fall-back tiers)
//   ab    keeps the previous form of a design decision selectable, for same-box A/B measurements of the two forms
//   tune  a resource size (waves per CU, chunk sizes); the default is the measured optimum on MI355X
//   diag  prints / profiles; never changes results
#pragma once
#include <cstddef>

#define PMX_OPTION_TABLE(X)                                                                                                                   \
    X(DEVICE, "env", "device ordinal the --refine worker contexts (and the command line) open; default 0")                                    \
    X(INDEX_THREADS, "env", "host threads of the index producer (default: hardware threads, at most 16)")                                      \
    X(FASTX_THREADS, "env", "host threads of the FASTQ scan")                                                                                  \
    X(BAM_THREADS, "env", "host threads of the BAM writer (record building + BGZF deflate)")                                                   \
    X(DIST_HOST_DIR, "env", "multi-rank exchange through files in this directory instead of RCCL: functional tests on a box with fewer GPUs than ranks; never for a number") \
    X(CTX_POOLED_QUEUE, "ab", "context streams from HIP's pool of hardware queues instead of a queue of their own")                            \
    X(SEED_NO_COLLAPSE, "ab", "seeding: every read through the seeding kernel, no collapse of identical reads (k_collapse_reads)")             \
    X(SEED_GENERIC, "ab", "seeding: the generic kernel (LDS rings) also for the default k=19 s=8 parameters")                                  \
    X(SEED_NO_SORT, "ab", "seeding: reads in input order, not in locality order")                                                              \
    X(SEED_SAFE_BOUND, "ab", "seeding: size the seed table by one key per base from the start (no optimistic table + redo)")                   \
    X(SEED_NO_HINT, "ab", "seeding: do not size the optimistic table by the previous histogram's density")                                     \
    X(SEED_BOUND_DIV, "test", "seeding: optimistic table = safe bound / N (a large N forces the overflow redo)")                               \
    X(SEED_CHUNK_MB, "tune", "seeding: bases per launch, in MB (default: a third of the range, 64..512)")                                      \
    X(SEED_PAR, "tune", "seeding: concurrent launches per group, 1..4 (default 3)")                                                            \
    X(SEED_BATCHES, "tune", "seeding: batches of 128 reads per block (default 1)")                                                             \
    X(PLACE_PROF, "diag", "place stage: print table sizes / failed inserts")                                                                   \
    X(PLACE_LEVEL_KERNELS, "ab", "node scoring: one launch per BFS level instead of the heavy-path chains kernel")                             \
    X(PLACE_TREE_KERNEL, "ab", "node scoring: the flag-per-node persistent kernel instead of the chains kernel")                               \
    X(PLACE_NO_GRAPH, "ab", "node scoring by levels: plain launches instead of the captured HIP graph")                                        \
    X(PLACE_TEST_STARVED, "test", "node scoring: make the persistent kernel report a starved grid (exercises the level-kernel fall-back)")     \
    X(ALIGN_NO_COMPACT, "ab", "align: skip the compact tier (every pair through the general tiers)")                                           \
    X(ALIGN_COMPACT_FUSED, "ab", "align: compact tier as one kernel (sketch + probes inside k_align_compact)")                                 \
    X(ALIGN_COMPACT_POS32, "test", "align: compact tier with 32-bit position words whatever the reference length")                             \
    X(ALIGN_EARLY_TAIL, "ab", "align: the pairs the compact tier's seeds kernel gave up on run on a second stream beside the chain kernels")   \
    X(ALIGN_NO_MULTI, "ab", "align: compact tier without its second form (several regions per mate): every bail to the thread-per-pair tier")  \
    X(ALIGN_COMPACT_WAVES, "tune", "align: compact chain kernel on a resident grid of N waves per CU (default: one workgroup per 64 pairs)")    \
    X(ALIGN_CSEED_WAVES, "tune", "align: compact seeds kernel on a resident grid of N waves per CU")                                           \
    X(ALIGN_NO_TPP, "ab", "align: no thread-per-pair tier (bails straight to the wave tiers)")                                                 \
    X(ALIGN_NO_TIER1, "ab", "align: no compact-layout wave tier")                                                                              \
    X(ALIGN_NO_DP_SERVICE, "ab", "align: thread-per-pair tier posts no DP requests")                                                           \
    X(ALIGN_NO_DP_GROUP, "ab", "align: DP requests through the wave-per-request service only (no eight-lane groups)")                          \
    X(ALIGN_DP_ONE_CLASS, "ab", "align: wave-per-request DP service in one size class")                                                        \
    X(ALIGN_NO_ROWS_DP, "ab", "align: anti-diagonal DPs in the wave tiers (no row-by-row form)")                                               \
    X(ALIGN_NO_DP_FAST, "ab", "align, long reads: no LDS copy of small DPs' arrays")                                                           \
    X(ALIGN_NO_WORK_QUEUE, "ab", "align, long reads: reads strided over the waves instead of drawn from a counter")                            \
    X(ALIGN_NO_PAIR_SORT, "ab", "align: pairs in input order")                                                                                 \
    X(ALIGN_PAIR_KEY1, "ab", "align: pair order by mate 1's locality key alone")                                                               \
    X(ALIGN_NO_MV_HANDOVER, "ab", "align: wave tier sketches again instead of taking the minimizers a DP-posting pair left")                   \
    X(ALIGN_NO_LDS_RING, "ab", "align: thread-per-pair minimizer window ring in the arena instead of LDS")                                     \
    X(ALIGN_NO_LANE_RING, "ab", "align: wave tier window ring in LDS instead of across the lanes")                                             \
    X(ALIGN_RESIDENT_GRID, "ab", "align: wave tiers on a resident strided grid even for few pairs")                                            \
    X(ALIGN_HOST_INDEX, "ab", "align: reference minimizer index built on the host and uploaded (the device build's checker)")                  \
    X(ALIGN_BAIL_TPP_MIN, "tune", "align: fewer compact-tier bails than this go straight to the wave tier (default 4096)")                     \
    X(ALIGN_TPP_MIN, "tune", "align: fewer DP requests than this end the service rounds (default 8192)")                                       \
    X(ALIGN_TPP_WAVES, "tune", "align: thread-per-pair waves per CU (default 16)")                                                             \
    X(ALIGN_TPP_TB, "tune", "align: thread-per-pair in-lane traceback bytes (default 0: DPs are requests)")                                    \
    X(ALIGN_DPG_WAVES, "tune", "align: grouped DP service waves per CU (default 8)")                                                           \
    X(ALIGN_WAVES, "tune", "align: wave tiers' target waves per SIMD (default 4)")                                                             \
    X(ALIGN_LDS_KB, "tune", "align: LDS budget of a wave-tier wave, KB")                                                                       \
    X(ALIGN_SLAB_MB, "test", "align: cap of the wave tiers' HBM slabs, MB (forces a small grid)")                                              \
    X(ALIGN_TB_KB, "test", "align, long reads: first-launch traceback per wave, KB (forces the retry launch)")                                 \
    X(ALIGN_CIGAR_CAP, "test", "align: CIGAR arena words (forces the overflow redo)")                                                          \
    X(ALIGN_TEST_MAX_CIGAR, "test", "align: cap the CIGAR operations per region in every tier (invalid records at the boundary)")              \
    X(ALIGN_PROF, "diag", "align: per-phase cycle stamps of lane 0, printed per call")                                                         \
    X(ALIGN_VERBOSE, "diag", "align: print the wave-tier launches")                                                                            \
    X(DP_HIST, "diag", "align: print the shapes of the posted DP requests")                                                                    \
    X(DPG_PROF, "diag", "align: grouped DP service phase cycles")                                                                              \
    X(DPG_SHADOW, "diag", "align: run both DP services and compare their results")                                                             \
    X(DPG_CHECK_LIST, "diag", "align: check the grouped service's sorted request list")                                                        \
    X(DPG_NO_SERVE, "test", "align: grouped DP service collects but serves nothing")                                                           \
    X(DPG_LEFT_TO_WAVE_TIER, "tune", "align: at most this many requests the grouped service leaves go to the wave tier (default 256)")

namespace pmx {
enum OptId {
#define PMX_OPT_ENUM(name, cls, doc) O_##name,
    PMX_OPTION_TABLE(PMX_OPT_ENUM)
#undef PMX_OPT_ENUM
    O_COUNT
};
// the value of PMX_<name> when the environment was last read, nullptr when it was not set (a drop-in for getenv)
const char* opt_str(OptId id);
void options_reload();
// "PMX_NAME [class] doc (= value)" lines into buf; returns the bytes needed (including the terminator)
size_t options_describe(char* buf, size_t cap);
}  // namespace pmx
