// Declarations shared between place_kernels.hip and the host orchestration in api_device.hip.
#pragma once
#include <stdint.h>

#define PMX_EMPTY_KEY 0xFFFFFFFFFFFFFFFFULL
#define PMX_SEED_BLOCK 128
#define PMX_SEED_CACHE 512   // entries of a block's (seed, count) cache in LDS (k_seed_histogram_ks)
#define PMX_SEED_QCAP 256   // seeds a wave queues in LDS before it inserts them 64 at a time (k_seed_histogram)
#define PMX_SEED_QCAP_KS 1024   // k_seed_histogram_ks: drained once per block of k-s+1 bases
#define PMX_DEDUP_BLOCK 1024   // reads a block of k_collapse_reads compares with each other
#define PMX_DEDUP_LDS_BYTES (20 * PMX_DEDUP_BLOCK * 4)   // 16 record dwords + hash + count + two table slots per read
#define PMX_SUM_BLOCK 1024
#define PMX_CTR_NSHARD 256
enum { PMX_CTR_ENTRIES = 0, PMX_CTR_OVERFLOW = 1, PMX_CTR_SEEDS = 2, PMX_CTR_COMPACT = 3, PMX_CTR_SHARD0 = 8, PMX_CTR_N = 8 + PMX_CTR_NSHARD };

namespace pmx {

struct SeedParams {
    int k, s, t, l, open;
    int trim_start, trim_end;
};

__global__ void k_pack_reads(const uint8_t* ascii, const int64_t* off, const int64_t* woff, int64_t n_reads, int64_t n_words,
                             uint64_t* words, uint32_t* amb, int64_t r0, int64_t r1, uint8_t* recs);
__global__ void k_pack_reads_fixed(const uint8_t* ascii, int64_t off0, int len, int64_t w_lo, int64_t w_hi, uint64_t* words, uint32_t* amb, uint8_t* recs);
// read -> its number of 32-base words, plus the checks a wrapped read set needs (stats: [0] max length, [1] 1 if offsets are
// not monotone or leave [0, total_bytes])
__global__ void k_read_word_counts(const int64_t* off, int64_t n_reads, int64_t total_bytes, int64_t* nwords, unsigned long long* stats);
__global__ void k_seed_histogram(const uint64_t* words, const uint32_t* amb, const int64_t* woff, const int64_t* off, int64_t r_begin, int64_t n_reads,
                                 SeedParams sp, uint64_t* keys, unsigned long long* vals, uint64_t mask, unsigned long long* counters,
                                 const uint8_t* keep, const uint8_t* qual, int min_q, uint64_t* list_hash, uint8_t* list_rev, uint32_t* list_n);
template <int K, int S, int L>
__global__ void k_seed_histogram_ks(const uint64_t* words, const uint32_t* amb, const int64_t* woff, const int64_t* off, int64_t r_begin,
                                    int64_t n_reads, SeedParams sp, uint64_t* keys, unsigned long long* vals, uint64_t mask,
                                    unsigned long long* counters, const uint8_t* keep, const uint32_t* perm, const uint32_t* t_len,
                                    const uint32_t* t_mult, const unsigned long long* t_count);
__global__ void k_collapse_reads(const uint64_t* words, const uint32_t* amb, const int64_t* woff, const int64_t* off, int64_t r_begin, int64_t r_end,
                                 const uint8_t* keep, const uint32_t* perm, int min_len, int fixed_len, const uint8_t* recs, uint64_t* t_words, uint32_t* t_amb, uint32_t* t_len,
                                 uint32_t* t_mult, unsigned long long* n_out);
__global__ void k_read_prefix_keys(const uint64_t* words, const int64_t* woff, int64_t r_begin, int64_t n_reads, uint32_t* key, uint32_t* idx);
__global__ void k_read_hashes(const uint8_t* ascii, const int64_t* off, int64_t n_reads, uint64_t* h1, uint64_t* h2, uint32_t* idx);
__global__ void k_gather_u64(const uint64_t* src, const uint32_t* idx, int64_t n, uint64_t* dst);
__global__ void k_kept_read_hashes(const uint64_t* h1, const uint64_t* h2, const uint8_t* keep, int64_t n, uint64_t* out_h1, uint64_t* out_h2,
                                   unsigned long long* n_out);
__global__ void k_drop_seen_reads(const uint64_t* h1, const uint64_t* h2, int64_t n, const uint64_t* seen_h1, const uint64_t* seen_h2, int64_t n_seen,
                                  uint8_t* keep);
__global__ void k_mark_first_of_run(const uint8_t* ascii, const int64_t* off, const uint64_t* h1s, const uint64_t* h2, const uint32_t* perm,
                                    int64_t n, uint8_t* keep);
__global__ void k_table_merge(const uint64_t* hash, const int64_t* count, int64_t n, uint64_t* keys, unsigned long long* vals,
                              uint64_t mask, unsigned long long* counters);
__global__ void k_table_rehash(const uint64_t* okeys, const unsigned long long* ovals, uint64_t ocap, uint64_t* keys,
                               unsigned long long* vals, uint64_t mask, unsigned long long* counters);
__global__ void k_table_compact(const uint64_t* keys, const unsigned long long* vals, uint64_t cap, uint64_t* out_hash,
                                int64_t* out_count, unsigned long long* n_out);
__global__ void k_mark_homopolymer(const uint64_t* hash, int64_t n, uint64_t h0, uint64_t h1, uint64_t h2, uint64_t h3, uint8_t* dead);
__global__ void k_mask_keys(const int64_t* count, const uint8_t* dead, int64_t n, uint64_t* key, uint32_t* idx);
__global__ void k_mask_apply(const uint32_t* idx, int64_t n_mask, uint8_t* dead);
__global__ void k_hist_stats(const int64_t* count, const uint8_t* dead, int64_t n, unsigned long long* stats);
__global__ void k_keep_flags(const int64_t* count, const uint8_t* dead, int64_t n, int64_t min_support, uint32_t* flag);
__global__ void k_keep_scatter(const uint64_t* hash, const int64_t* count, const uint32_t* flag, const uint32_t* pos, int64_t n,
                               uint64_t* kept_hash, double* kept_log);
__global__ void k_block_sums(const double* kept_log, int64_t n, double* partial);
__global__ void k_sequential_sums(const double* partial, int64_t nb, double* out);
__global__ void k_kept_table_build(const uint64_t* kept_hash, const double* kept_log, int64_t n, uint64_t* tkeys, double* tvals,
                                   uint64_t mask);
__global__ void k_wc_denominator(const uint64_t* ch_hash, const int16_t* ch_child, uint64_t beg, uint64_t end, const uint64_t* tkeys,
                                 const double* tvals, uint64_t mask, int has_kept, double* out);
__global__ void k_score_terms(const uint64_t* ch_hash, const int16_t* ch_par, const int16_t* ch_child, int64_t n_changes,
                              const uint64_t* tkeys, const double* tvals, uint64_t mask, int has_kept, double* t_mag, double* t_raw,
                              double* t_cos, double* t_wc, double* t_lc, uint8_t* t_meta);
__global__ void k_score_level(const uint32_t* level_nodes, int64_t n_level, const uint32_t* parent, const uint64_t* offsets,
                              const double* t_mag, const double* t_raw, const double* t_cos, const double* t_wc, const double* t_lc,
                              const uint8_t* t_meta, double* metrics5, int64_t* counts2);
__global__ void k_score_tree(const uint32_t* order, int64_t n_nodes, const uint32_t* parent, const uint64_t* offsets, const double* t_mag,
                             const double* t_raw, const double* t_cos, const double* t_wc, const double* t_lc, const uint8_t* t_meta,
                             double* metrics5, int64_t* counts2, uint32_t* done, uint32_t epoch, uint32_t* status);
__global__ void k_score_chains(const uint32_t* chain_off, int64_t n_chains, const uint32_t* chain_nodes, const uint64_t* chain_beg,
                               const uint64_t* chain_end, const uint32_t* parent, const double* t_mag, const double* t_raw, const double* t_cos,
                               const double* t_wc, const double* t_lc, const uint8_t* t_meta, double* metrics5, int64_t* counts2,
                               uint32_t* done, uint32_t epoch, uint32_t* status);
__global__ void k_score_getters(const double* metrics5, const int64_t* counts2, int64_t n_nodes, const double* scalars, int64_t n_kept,
                                double* scores5, const uint32_t* order, double* scores_bfs);
__global__ void k_fill_u64(uint64_t* p, uint64_t v, uint64_t n);

}  // namespace pmx
