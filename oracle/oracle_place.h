/*
 * oracle_place.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's "place" stage (read seeding, k-min-mer
 * histogram, read-side magnitudes, per-node delta scoring, best/tie rule).  It is the
 * parity checker for the HIP path in panmap_amd/csrc; nothing under panmap_amd/ may
 * include, link or dlopen it.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it.
 *
 * Pinning (see DESIGN.md "Oracle"): known-answer vectors produced by the compiled
 * reference, the reference's own unit-test contracts
 * (src/test/test_seeding.cpp, src/test/test_placement.cpp) and the end-to-end golden
 * examples/expected/single_sample/isolate.placement.tsv.
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * reference checkout).
 */
#ifndef ORACLE_PLACE_H
#define ORACLE_PLACE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/seeding.hpp:100-112 */
uint64_t orc_chash(int c);
/* src/seeding.cpp:20-30 ; returns 0, or -1 if the k-mer holds a non-ACGT base */
int orc_hash_seq(const char *s, int k, uint64_t *fwd, uint64_t *rev);
/* src/placement.cpp:41-76 */
uint64_t orc_homopolymer_hash(int base, int k);

/* src/seeding.cpp:47-229.  return_all != 0 -> one entry per k-mer window (hash==UINT64_MAX
 * for non-syncmers); else syncmers only.  Output arrays must hold len-k+1 entries. */
int64_t orc_rolling_syncmers(const char *seq, int64_t len, int k, int s, int open, int t, int return_all,
                             uint64_t *hash, uint8_t *is_rev, uint8_t *is_sync, int64_t *pos);

/* src/placement.cpp:1611-1682 (l>=1) and :1339-1366 (l==0): the seeds one read contributes.
 * out must hold len entries.  Returns the number of seeds. */
int64_t orc_read_seeds(const char *seq, int64_t len, int k, int s, int l, int open, int t,
                       int trim_start, int trim_end, uint64_t *out);

/* seed histogram: src/placement.cpp:1663,1680 (localMap[h] += multiplicity), :922-929 (merge) */
typedef struct orc_hist orc_hist;
orc_hist *orc_hist_new(void);
void orc_hist_free(orc_hist *h);
void orc_hist_add(orc_hist *h, uint64_t key, int64_t mult);
/* quality-filtered variant (src/placement.cpp:1386-1527); multiplicity is always 1 there */
int64_t orc_read_seeds_q(const char *seq, const char *qual, int64_t len, int64_t qual_len, int k, int s, int l,
                         int open, int t, int trim_start, int trim_end, int min_q, uint64_t *out);
void orc_hist_add_read_q(orc_hist *h, const char *seq, const char *qual, int64_t len, int64_t qual_len, int k, int s, int l,
                         int open, int t, int trim_start, int trim_end, int min_q);
void orc_hist_add_read(orc_hist *h, const char *seq, int64_t len, int k, int s, int l, int open, int t,
                       int trim_start, int trim_end, int64_t multiplicity);
/* batch / multi-threaded entries for the CPU baseline (one C call; src/placement.cpp:1611-1686, :922-929) */
void orc_hist_add_reads(orc_hist *h, const char *concat, const int64_t *off, int64_t r0, int64_t r1, int k, int s, int l,
                        int open, int t, int trim_start, int trim_end);
void orc_hist_merge(orc_hist *dst, const orc_hist *src);
orc_hist *orc_hist_build_mt(const char *concat, const int64_t *off, int64_t n_reads, int k, int s, int l, int open, int t,
                            int trim_start, int trim_end, int n_threads);
int64_t orc_hist_size(const orc_hist *h);
/* ascending hash order */
void orc_hist_export_sorted(const orc_hist *h, uint64_t *hash, int64_t *count);

typedef struct {
    int64_t min_support;      /* resolved (placement.cpp:931-955) */
    int64_t n_unique_in;      /* unique seeds after homopolymer/mask erase */
    int64_t n_kept;           /* readUniqueSeedCount */
    int64_t total_freq;       /* totalReadSeedFrequency */
    double  log_magnitude;    /* sqrt(sum L^2) */
    double  log_cont_den;     /* sum L */
    double  est_coverage;
} orc_read_state;

/* src/placement.cpp:1703-1722 (homopolymer erase), :1774-1799 (mask), :931-984 (support, log1p, sums).
 * Input sorted by hash ascending.  FP sums run in ascending-hash order (canonical order, SURVEY
 * Appendix D-1).  kept_hash/kept_log must hold n entries.  Returns n_kept. */
int64_t orc_finalize_reads(const uint64_t *hash, const int64_t *count, int64_t n, int k,
                           double mask_fraction, int min_support_cfg,
                           uint64_t *kept_hash, double *kept_log, orc_read_state *st);

/* Per-node running metrics (src/placement.hpp:108-155), 7 per node:
 *   [0] logRawNumerator [1] logCosineNumerator [2] weightedContainmentNumerator
 *   [3] logContainmentNumerator [4] genomeMagnitudeSquared   (doubles)
 *   cnt[0] presenceIntersectionCount  cnt[1] genomeUniqueSeedCount (int64)
 * and the 5 scores in the TSV order {log_raw, log_cosine, containment, weighted_containment,
 * log_containment}.  Nodes are in DFS pre-order, parent[i] < i, parent[0] == 0.
 * src/placement.cpp:242-345 (deltas), :1863-1876 (weighted-containment denominator). */
void orc_score_nodes(int64_t n_nodes, const uint32_t *parent, const uint64_t *offsets,
                     const uint64_t *ch_hash, const int16_t *ch_par, const int16_t *ch_child,
                     int64_t n_kept, const uint64_t *kept_hash, const double *kept_log,
                     double log_magnitude, double log_cont_den,
                     double *wc_den_out, double *metrics5, int64_t *counts2, double *scores5);

/* src/placement.cpp:355-371 (update rule), :395-401 (finalize), visit order = single-thread BFS
 * (:742-878) = ascending (depth, DFS index).  tie_out: 5 rows of capacity tie_cap; n_tie_out[5].
 * best_idx is the lowest tied index (UINT32_MAX if no node scored > 0). */
void orc_best_ties(int64_t n_nodes, const uint32_t *parent, const double *scores5, int force_leaf,
                   double *best_score, uint32_t *best_idx, uint32_t *tie_out, int64_t tie_cap,
                   int64_t *n_tie_out);

#ifdef __cplusplus
}
#endif
#endif
