/*
 * panmap_amd.h -- C ABI of the MI355X-native panmap hot path (index -> place -> align).
 *
 * One shared library, libpanmap_amd.so: plain pointers and sizes, no C++/torch types.  Every
 * entry point cites the reference interface it stands in for (paths relative to amkram/panmap).
 * All functions returning int use 0 = success, negative = error (pmx_last_error() has the text);
 * there is NO CPU fallback: device entry points fail with PMX_ERR_NO_DEVICE when no gfx950 GPU is
 * usable.  Handles are thread-compatible, not thread-safe; one pmx_ctx per GPU.
 */
#ifndef PANMAP_AMD_H
#define PANMAP_AMD_H

#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PMX_OK 0
#define PMX_ERR_ARG (-1)
#define PMX_ERR_IO (-2)
#define PMX_ERR_FORMAT (-3)
#define PMX_ERR_NO_DEVICE (-4)
#define PMX_ERR_DEVICE (-5)
#define PMX_ERR_CAPACITY (-6)
#define PMX_ERR_UNSUPPORTED (-7)

const char *pmx_last_error(void);
const char *pmx_version(void);

/* ------------------------------------------------------------------------------------------
 * PanMAN + node genomes (host).  Replaces loadPanMAN (src/main.cpp:313-325, external panman
 * library) and panmapUtils::getStringFromReference (src/panmap_utils.cpp:182-190), which
 * runAlignment uses to materialise the placed node's genome (src/main.cpp:1757-1771).
 * ------------------------------------------------------------------------------------------ */
typedef struct pmx_panman pmx_panman;
int pmx_panman_open(const char *path, pmx_panman **out);
void pmx_panman_close(pmx_panman *pm);
int64_t pmx_panman_num_nodes(const pmx_panman *pm);
int64_t pmx_panman_num_blocks(const pmx_panman *pm);
int64_t pmx_panman_num_columns(const pmx_panman *pm);
const char *pmx_panman_node_id(const pmx_panman *pm, int64_t dfs_index);
int64_t pmx_panman_parent(const pmx_panman *pm, int64_t dfs_index);
int64_t pmx_panman_find_node(const pmx_panman *pm, const char *node_id); /* -1 if absent */
/* writes the ungapped genome (no NUL) into buf if cap suffices; always returns its length */
int64_t pmx_panman_node_genome(const pmx_panman *pm, int64_t dfs_index, char *buf, int64_t cap);
/* test hook (src/test/test_index.cpp:145-193 injects the same mutation into a loaded tree): appends a pure
 * inversion of the largest forward block with >= min_bases bases to the node's block mutations; returns the
 * block id, or -1 when the node has no such block */
int64_t pmx_panman_test_invert_block(pmx_panman *pm, int64_t dfs_index, int min_bases);

/* ------------------------------------------------------------------------------------------
 * Seed index (host).  Replaces IndexBuilder::buildIndexParallel (src/index_single_mode.hpp:207-214)
 * and the zero-copy SoA binding the place stage does on the loaded index
 * (src/placement.cpp:1054-1085; layout src/index_lite.capnp:36-70).
 * ------------------------------------------------------------------------------------------ */
typedef struct pmx_index pmx_index;
typedef struct {
    int32_t k, s, t, l;
    int32_t open_syncmer;
    int32_t hpc;
    int32_t flank_mask;
    int32_t reserved;
    int64_t n_nodes;
    int64_t n_changes;
} pmx_index_info;

int pmx_index_build(const pmx_panman *pm, int k, int s, int t, int l, int open_syncmer, int flank_mask,
                    pmx_index **out);
/* mode 0 = automatic (incremental DFS; from-scratch re-seeding of every node when the PanMAN has inverted
 * blocks), 1 = from scratch, 2 = incremental; max_nodes >= 0 stops after that many nodes in DFS order
 * (the remaining nodes carry no changes) -- both exist for the cross-check tests of the producer */
int pmx_index_build_ex(const pmx_panman *pm, int k, int s, int t, int l, int open_syncmer, int flank_mask,
                       int mode, int64_t max_nodes, pmx_index **out);
/* mode | PMX_INDEX_ORIENTED: the --meta form of the index.  A k-min-mer that reads right-to-left on the node's genome
 * (reverse rolled hash < forward rolled hash) is keyed hash ^ PMX_ORIENT_XOR: the per-node count changes then tell the
 * two orientations of one hash apart -- what the reference's MGSR index keeps as seedInfos[].isReverse
 * (src/mgsr.cpp:7225-7320 counts (forward, reverse) occurrences per hash).  Needs l >= 2. */
#define PMX_INDEX_ORIENTED 0x100
#define PMX_ORIENT_XOR 0x9e3779b97f4a7c15ULL
/* adopt caller-provided SoA arrays (copied): parent[n], offsets[n+1], hash/pc/cc[offsets[n]] */
int pmx_index_from_arrays(const pmx_index_info *info, const uint32_t *parent, const uint64_t *offsets,
                          const uint64_t *hash, const int16_t *parent_count, const int16_t *child_count,
                          pmx_index **out);
/* The `.idx` container the reference's place stage loads and its index stage writes (32-byte PMI1 header + Cap'n Proto
 * LiteIndex message, raw or as concatenated zstd frames; src/index_single_mode.cpp:1561-1640, src/placement.cpp:1009-1092,
 * src/index_lite.capnp:36-70).  Load rejects a stale format version / missing struct-of-arrays fields / short offsets with
 * the reference's messages (pmx_last_error).  zstd_level as --zstd-level; uncompressed != 0 writes the mmap-able form. */
int pmx_index_save(const pmx_index *idx, const char *path, int zstd_level, int uncompressed);
int pmx_index_load(const char *path, pmx_index **out);
/* the seeding parameters from the header alone (cache validation, src/main.cpp:371-396); PMX_ERR_FORMAT if absent */
int pmx_index_read_header(const char *path, pmx_index_info *info, int *uncompressed);
const char *pmx_index_node_id(const pmx_index *idx, int64_t dfs_index); /* "" when the index was adopted from arrays */
void pmx_index_close(pmx_index *idx);
int pmx_index_get_info(const pmx_index *idx, pmx_index_info *info);
const uint32_t *pmx_index_parents(const pmx_index *idx);      /* n_nodes   */
const uint64_t *pmx_index_offsets(const pmx_index *idx);      /* n_nodes+1 */
const uint64_t *pmx_index_hashes(const pmx_index *idx);       /* n_changes */
const int16_t *pmx_index_parent_counts(const pmx_index *idx); /* n_changes */
const int16_t *pmx_index_child_counts(const pmx_index *idx);  /* n_changes */

/* ------------------------------------------------------------------------------------------
 * Device context (one per GPU / per process rank).
 * ------------------------------------------------------------------------------------------ */
typedef struct pmx_ctx pmx_ctx;
int pmx_ctx_create(int device_ordinal, pmx_ctx **out);
void pmx_ctx_destroy(pmx_ctx *ctx);
int pmx_ctx_synchronize(pmx_ctx *ctx);
/* the HIP stream all kernels of this context are launched on (hipStream_t as void*).  Every context owns a hardware queue
   (a CU-masked stream with every CU enabled): contexts used from different host threads overlap their kernels. */
void *pmx_ctx_stream(pmx_ctx *ctx);

/* ------------------------------------------------------------------------------------------
 * Read sets.  The reference hands reads around as std::vector<std::string>
 * (extractReadSequences src/placement.cpp:164-197; readFastqPaired src/seeding.cpp:231-269).
 * Here: one concatenated ASCII buffer + n+1 offsets, uploaded once; pmx_readset_pack converts it
 * on the GPU to 2 bits/base + an ambiguity bitmask (every read starts on a 32-base word).
 * ------------------------------------------------------------------------------------------ */
typedef struct pmx_readset pmx_readset;
int pmx_readset_upload(pmx_ctx *ctx, const char *concat, const int64_t *offsets, int64_t n_reads,
                       pmx_readset **out);
/* wrap ASCII reads already resident in device memory (d_concat: total bytes, d_offsets: n+1 int64) */
int pmx_readset_wrap_device(pmx_ctx *ctx, const void *d_concat, const void *d_offsets, int64_t n_reads,
                            int64_t total_bytes, int64_t max_read_len, pmx_readset **out);
/* re-point an existing read set at another batch resident in device memory (streaming: the object's packed-read buffers
   are reused, the word offsets are computed and the offsets validated on the device; call pmx_readset_pack afterwards) */
int pmx_readset_rewrap_device(pmx_ctx *ctx, pmx_readset *rs, const void *d_concat, const void *d_offsets, int64_t n_reads,
                              int64_t total_bytes, int64_t max_read_len);
int pmx_readset_pack(pmx_ctx *ctx, pmx_readset *rs);
/* streaming form of the same: pack the reads [r0, r1) of a (re)wrapped read set as soon as THEIR bases have landed in the
   wrapped buffer (the offsets of the whole set are in place since the wrap); the set counts as packed once the ranges
   cover it.  Stream-ordered on the context's stream, no host round trip. */
int pmx_readset_pack_range(pmx_ctx *ctx, pmx_readset *rs, int64_t r0, int64_t r1);
/* FASTQ quality strings of the same reads (same offsets; one Phred+33 byte per base): only needed for
 * pmx_place_params.min_seed_quality > 0 (allReadQualities, src/placement.cpp:1386) */
int pmx_readset_set_qualities(pmx_ctx *ctx, pmx_readset *rs, const char *qual_concat);
/* Optional: enqueue the align stage's pair order (pairs by both mates' locality keys) of a packed, paired read set NOW, on a
 * side stream of the context.  It depends on the reads alone, so made here it runs beside the place stage instead of between
 * the placement and the first align kernel; an aligner that finds none makes it itself.  Same results either way. */
int pmx_readset_order_pairs(pmx_ctx *ctx, pmx_readset *rs);
void pmx_readset_free(pmx_ctx *ctx, pmx_readset *rs);
int64_t pmx_readset_num_reads(const pmx_readset *rs);

/* ------------------------------------------------------------------------------------------
 * PLACE stage on the GPU.  Together these replace placement::placeLite
 * (src/placement.hpp:237-244, src/placement.cpp:986-2018).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    /* TraversalParams fields that change results (src/placement.hpp:28-54); k/s/t/l/open come
       from the index (src/placement.cpp:1094-1101) */
    double seed_mask_fraction; /* CLI default 0 (src/main.cpp:1967) */
    int32_t min_read_support;  /* -1 = auto */
    int32_t trim_start, trim_end;
    int32_t dedup_reads;       /* --dedup: each distinct sequence counted once */
    int32_t force_leaf;
    int32_t min_seed_quality;  /* --min-seed-quality: > 0 with qualities attached selects the quality-filtered
                                  seeding branch (src/placement.cpp:1386-1527; it ignores dedup_reads) */
    int32_t reserved[2];
} pmx_place_params;

typedef struct {
    double best_score[5];     /* log_raw, log_cosine, containment, weighted_containment, log_containment */
    uint32_t best_index[5];   /* lowest tied DFS index; UINT32_MAX if none */
    int64_t n_tied[5];
    int64_t n_reads;
    int64_t n_unique_seeds;   /* histogram size after homopolymer/mask erase */
    int64_t n_kept_seeds;     /* readUniqueSeedCount */
    int64_t total_seed_freq;  /* totalReadSeedFrequency */
    int64_t min_support;      /* resolved */
    double log_read_magnitude;
    double log_containment_den;
    double weighted_containment_den;
} pmx_place_result;

typedef struct pmx_place pmx_place; /* device-resident index + working buffers for one sample */

/* uploads the index SoA (replicated per GPU) and precomputes BFS levels */
int pmx_place_create(pmx_ctx *ctx, const pmx_index *idx, pmx_place **out);
void pmx_place_free(pmx_ctx *ctx, pmx_place *pl);

/* [hot] reads -> syncmers -> k-min-mers -> seed histogram (src/placement.cpp:1611-1686).
   Accumulates into the place object's device histogram; call once per read shard. */
int pmx_place_reset(pmx_ctx *ctx, pmx_place *pl);
int pmx_place_add_reads(pmx_ctx *ctx, pmx_place *pl, const pmx_readset *rs, const pmx_place_params *pp);
/* the reads [r0, r1) of the set alone (packed by pmx_readset_pack or pmx_readset_pack_range): a batch is seeded range by
   range while the H2D copy of the next range is in flight; the sum over the ranges is the histogram of the set.
   --dedup needs the whole set (PMX_ERR_UNSUPPORTED on a proper sub-range). */
int pmx_place_add_reads_range(pmx_ctx *ctx, pmx_place *pl, const pmx_readset *rs, int64_t r0, int64_t r1,
                              const pmx_place_params *pp);
/* export / import of the (hash,count) histogram, ascending hash: the multi-GPU exchange step
   (all-gather of per-rank histograms, then merge) */
int64_t pmx_place_histogram_size(pmx_ctx *ctx, pmx_place *pl);
int pmx_place_histogram_export(pmx_ctx *ctx, pmx_place *pl, uint64_t *hash, int64_t *count, int64_t cap);
int pmx_place_histogram_merge(pmx_ctx *ctx, pmx_place *pl, const uint64_t *hash, const int64_t *count, int64_t n);
/* same with caller-owned DEVICE buffers (e.g. torch tensors fed to an RCCL all-gather): no host bounce */
int pmx_place_histogram_export_device(pmx_ctx *ctx, pmx_place *pl, void *d_hash, void *d_count, int64_t cap);
/* for the multi-GPU exchange: the number of distinct seeds without sorting them, and the (hash, count) pairs in table
 * order written into DEVICE buffers (stream-ordered on the context's stream; synchronize the context before another
 * stream reads them).  The ranks' parts are merged with pmx_place_histogram_merge_device_parts; order plays no role. */
int64_t pmx_place_histogram_entries(pmx_ctx *ctx, pmx_place *pl);
int pmx_place_histogram_export_device_unsorted(pmx_ctx *ctx, pmx_place *pl, void *d_hash, void *d_count, int64_t cap);
/* n_parts (hash,count) runs laid out `part_stride` elements apart (the all-gather buffer of the multi-GPU
 * exchange: rank p's run at d_hash + p * part_stride, sizes[p] valid entries); skip_part = this rank's own run */
int pmx_place_histogram_merge_device_parts(pmx_ctx *ctx, pmx_place *pl, const void *d_hash, const void *d_count,
                                           int64_t part_stride, const int64_t *sizes, int n_parts, int skip_part);
int pmx_place_histogram_merge_device(pmx_ctx *ctx, pmx_place *pl, const void *d_hash, const void *d_count, int64_t n);

/* read-side filters + magnitudes (src/placement.cpp:1703-1856), then [hot] per-node delta scoring
   down the tree (src/placement.cpp:242-345, 701-918) and the sequential best/tie rule (:355-401) */
int pmx_place_score(pmx_ctx *ctx, pmx_place *pl, const pmx_place_params *pp, int64_t n_reads_total,
                    pmx_place_result *res);
/* tied DFS indices of metric m (ascending), after pmx_place_score */
int pmx_place_tied(const pmx_place *pl, int metric, uint32_t *out, int64_t cap);
/* optional per-node outputs (host copies): scores [n_nodes][5] double, metrics [n_nodes][5] double,
   counts [n_nodes][2] int64; any pointer may be NULL */
int pmx_place_node_outputs(pmx_ctx *ctx, pmx_place *pl, double *scores5, double *metrics5, int64_t *counts2);
/* --refine (refineTopCandidates, src/placement.cpp:516-698): per metric the top refine_top_pct of the nodes (at most
 * max_top_n, always with the metric's winner) plus their neighbours within neighbor_radius branches (at most
 * max_neighbor_n each) are candidates; every candidate is scored once through `fn` (minus the total edit distance of the
 * reads against its genome: pmx_align_score_reads on pmx_panman_node_genome) and each metric keeps its best candidate
 * (ties: higher seed score, then lower DFS index).  Host logic; parent / scores5 / best_index as pmx_index_parents,
 * pmx_place_node_outputs and pmx_place_result give them.  fn returns 0 on success; anything else aborts with that code. */
typedef struct {
    double top_pct;          /* 0.01  (src/main.cpp:186-190) */
    int32_t max_top_n;       /* 150 */
    int32_t neighbor_radius; /* 2 */
    int32_t max_neighbor_n;  /* 150 */
    int32_t reserved;
} pmx_refine_params;
typedef struct {
    int32_t ran;             /* 0: no node had a positive score */
    int32_t n_candidates;
    int64_t score[5];        /* refined_<metric> score, metric order of the placement TSV */
    uint32_t node[5];        /* DFS index, 0xffffffff = none */
    uint32_t reserved;
} pmx_refine_result;
typedef int (*pmx_refine_score_fn)(void *user, uint32_t dfs_index, int64_t *score);
/* the candidate set alone (ascending DFS indices; returns its size): a caller may score the candidates concurrently and
 * hand pmx_refine_top_candidates a lookup */
int64_t pmx_refine_candidates(const uint32_t *parent, int64_t n_nodes, const double *scores5, const uint32_t best_index[5],
                              const pmx_refine_params *rp, uint32_t *cand_nodes, int64_t cand_cap);
int pmx_refine_top_candidates(const uint32_t *parent, int64_t n_nodes, const double *scores5, const uint32_t best_index[5],
                              const pmx_refine_params *rp, pmx_refine_score_fn fn, void *user, pmx_refine_result *out,
                              uint32_t *cand_nodes, int64_t *cand_scores, int64_t cand_cap);
/* kept read seeds after filtering: hash ascending + log1p(count) */
int64_t pmx_place_kept_seeds(pmx_ctx *ctx, pmx_place *pl, uint64_t *hash, double *logc, int64_t cap);

/* ------------------------------------------------------------------------------------------
 * ALIGN stage.  B-align boundary: identical to the reference's C ABI (src/mm_align.h:20-53).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t pos;    /* 1-based ref position (INT_MAX if unmapped) */
    int32_t rs, re; /* 0-based ref start/end */
    int32_t qs, qe; /* query start/end */
    uint8_t mapq;
    uint8_t rev;
    uint8_t proper_frag;
    int32_t n_cigar;
    uint32_t *cigar; /* BAM-encoded; malloc'd, caller frees */
    char *md;
} read_align_t;

typedef struct {
    read_align_t r1;
    read_align_t r2;
    int mapped;
} align_pair_result_t;

/* Drop-in for src/mm_align.h:44-53 (selected through the aligner function pointer at
   src/conversion.cpp:416-426).  Same argument meaning; n_threads is accepted and ignored (the GPU
   owns the parallelism).  On failure returns with `results` untouched, like the reference. */
void pmx_align_reads_direct(const char *reference, const char *refName, int n_reads, const char **reads,
                            const char **quality, const char **read_names, const int *r_lens,
                            align_pair_result_t *results, bool pairedEndReads, int n_threads);

/* ---- FASTA / FASTQ ingest (host): what seeding::readFastqPaired (src/seeding.cpp:231-269) and extractReadSequences
 * (src/placement.cpp:164-197) do with kseq.h, as flat arrays that upload to the device as they are.
 * kseq conventions: name = header up to the first white space; multi-line sequence / quality; CR stripped; a record
 * whose quality length differs from its sequence length ends the file.
 * pmx_fastx_read_paired: mate 2 reverse-complemented (upper-case ACGT only) with its qualities reversed, mates
 * interleaved (r1_0, r2_0, r1_1, ...), missing qualities 'I' x length; path2 NULL or "" = single-end; a mate-count
 * mismatch is PMX_ERR_ARG (the reference exits).  pmx_fastx_read: one file, records in file order, FASTA qualities
 * are zero bytes.  Views stay valid until pmx_fastx_free. */
typedef struct pmx_fastx pmx_fastx;
int pmx_fastx_read_paired(const char *path1, const char *path2, pmx_fastx **out);
int pmx_fastx_read(const char *path, pmx_fastx **out);
int64_t pmx_fastx_num_reads(const pmx_fastx *fx);
int pmx_fastx_views(const pmx_fastx *fx, const char **seq_concat, const char **qual_concat, const int64_t **offsets,
                    const char **names_concat, const int64_t **name_offsets);
void pmx_fastx_free(pmx_fastx *fx);

/* ---- BAM egress (host): what alignAndWriteBam does with the results of align_reads_direct
 * (src/conversion.cpp:288-538: build_bam_from_result, compute_sam_flags, compute_tlen, sort by pos, BAM + .bai).
 * reads / quality / read_names / r_lens are the arrays that were handed to the aligner (R2 already reverse-
 * complemented and its qualities reversed).  Writes `bam_path` and `bam_path`.bai.  Returns PMX_OK or PMX_ERR_IO. */
int pmx_write_bam(const char *bam_path, const char *ref_name, int64_t ref_len, int n_reads, const char **reads,
                  const char **quality, const char **read_names, const int *r_lens,
                  const align_pair_result_t *results, bool pairedEndReads);

/* Device-resident form used by the pipeline / benchmark: fixed 32-byte records + CIGAR arena. */
typedef struct {
    int32_t rs, re, qs, qe;
    uint8_t mapq, rev, proper_frag, mapped;
    uint16_t n_cigar;
    uint16_t flags;      /* PMX_ALN_* status bits */
    uint32_t cigar_off;  /* index into the CIGAR arena (uint32 ops) */
    int32_t score;       /* dp_max of the primary */
} pmx_aln_record;

#define PMX_ALN_OVERFLOW 0x1     /* a fixed-capacity work buffer overflowed: record is invalid */
#define PMX_ALN_UNSUPPORTED 0x2  /* hit a reference branch not implemented on the GPU yet */
#define PMX_ALN_HAS_ALN 0x4      /* the record carries an alignment (rs/re/qs/qe/mapq/CIGAR are set) */

typedef struct pmx_aligner pmx_aligner;
/* builds the minimizer index of one reference genome on the device; preset chosen from the mean
   read length exactly as setup_minimap2(for_scoring=1) does (src/mm_align.c:118-188) */
int pmx_aligner_create(pmx_ctx *ctx, const char *reference, int64_t ref_len, int mean_read_len, pmx_aligner **out);
/* re-target an existing aligner at another reference genome (keeps the work buffers) */
int pmx_aligner_set_reference(pmx_ctx *ctx, pmx_aligner *al, const char *reference, int64_t ref_len, int mean_read_len);
/* What the current reference index holds, for checks of the two index builders against each other (the device build of
 * ref_index_kernels.hip is the default; PMX_ALIGN_HOST_INDEX=1 selects the host restatement of mm_idx_str):
 * out[0] minimizer occurrences, out[1] distinct minimizers, out[2] mid_occ, out[3] a digest of every
 * (minimizer, occurrence list) pair, independent of the table's slot order; out[4] = 1 when the index was built on the device. */
int pmx_aligner_index_digest(pmx_ctx *ctx, pmx_aligner *al, uint64_t out[5]);
void pmx_aligner_free(pmx_ctx *ctx, pmx_aligner *al);
/* [hot] map + align every read (pair) of a packed read set.  revcomp_mate2 != 0: odd-indexed reads
   are reverse-complemented on the fly (what readFastqPaired does on the host, src/seeding.cpp:251).
   Results stay on the device until fetched. */
int pmx_align_readset(pmx_ctx *ctx, pmx_aligner *al, const pmx_readset *rs, int paired, int revcomp_mate2);
/* [hot, --refine] score_reads_vs_reference (src/mm_align.c:144-199): maps every read (pair) against the aligner's
 * reference and returns minus the summed count_read_errors (edit distance of the first region: block length - matches +
 * ambiguous bases; the read length without one).  Leaves the records of the run behind like pmx_align_readset.
 * Paired scoring needs an even number of reads. */
int pmx_align_score_reads(pmx_ctx *ctx, pmx_aligner *al, const pmx_readset *rs, int paired, int revcomp_mate2, int64_t *score);
/* Drop-in for the reference's scorer with its own signature (src/mm_align.h:13-17): host strings in (interleaved mates when
 * paired_end, mate 2 as sequenced), minus the total edit distance out, 0 on failure.  An odd read of a paired set is mapped
 * alone (src/mm_align.c:178-185); reads of flagged records count as unmapped (pmx_last_error() says how many). */
int64_t pmx_score_reads_vs_reference(const char *reference, int n_reads, const char **reads, const int *r_lens, int kmer_size,
                                     bool paired_end);
/* The DP kernel on its own (A9: ksw_extd2_sse, src/3rdparty/minimap2/ksw2_extd2_sse.c:28-400, with the aligner's scoring
 * parameters): a batch of (query, target) pairs of nt4 codes (0..3, 4 = N) through the GROUPED DP SERVICE of the short-read
 * tiers (align_kernel_dpg.hip) -- the entry point of its parity tests against the reference function and of its throughput
 * measurement (bench.py `dp_service`).  Sequence i is seqs[q_off[i] .. q_off[i+1]) / seqs[t_off[i] .. t_off[i+1]); w,
 * zdrop, end_bonus, flag (KSW_EZ_* bits) per request as the reference takes them.  out[i].served = 0 for a request the
 * service does not take (a side longer than 128 bases, a band that cuts the matrix, other flag combinations, more than 20
 * CIGAR operations): the align tiers run those on the wave-per-request kernels.  `reps` > 1 repeats the launch (timing);
 * *kernel_ms (may be NULL) receives the duration of one launch of the service kernel. */
typedef struct {
    int32_t served;
    uint32_t max;
    int32_t zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score, n_cigar, reach_end;
    uint32_t cigar[20];
} pmx_dp_result;
/* the aligner's DP scoring: out[0..8] = a, b, q, e, q2, e2, sc_ambi, zdrop, end_bonus (mm_mapopt_t of the preset the mean
 * read length selected, src/mm_align.c:140-166) */
int pmx_align_scoring(const pmx_aligner *al, int32_t out[9]);
int pmx_align_dp_batch(pmx_ctx *ctx, pmx_aligner *al, const uint8_t *seqs, const int64_t *q_off, const int64_t *t_off, int64_t n,
                       const int32_t *w, const int32_t *zdrop, const int32_t *end_bonus, const int32_t *flag, pmx_dp_result *out,
                       int reps, double *kernel_ms);
int64_t pmx_align_num_records(const pmx_aligner *al);
int64_t pmx_align_cigar_words(pmx_ctx *ctx, pmx_aligner *al);
int pmx_align_fetch(pmx_ctx *ctx, pmx_aligner *al, pmx_aln_record *records, int64_t n_records, uint32_t *cigar_arena,
                    int64_t arena_cap);
/* the same download without waiting for it, on a stream of the caller's choice (hipStream_t as void*; NULL = the context's):
 * the copies start once the results are complete, and the next pmx_align_readset on this aligner waits for them before it
 * overwrites its buffers.  The caller synchronises `stream` before reading the (pinned) host buffers. */
int pmx_align_fetch_async(pmx_ctx *ctx, pmx_aligner *al, pmx_aln_record *records, int64_t n_records, uint32_t *cigar_arena,
                          int64_t arena_cap, void *stream);
/* copy the fixed-size records into a caller-owned DEVICE buffer (for RCCL gathers) */
int pmx_align_copy_records_device(pmx_ctx *ctx, pmx_aligner *al, void *d_records, int64_t n_records);
/* the same for the CIGAR arena (pmx_align_cigar_words() words): records + arena are what rank 0 needs to write the BAM */
int pmx_align_copy_cigars_device(pmx_ctx *ctx, pmx_aligner *al, void *d_cigars, int64_t n_words);
/* Work statistics of the last pmx_align_readset call (no reference counterpart; bench.py reports GCUPS from them):
   DP cells are counted as q * min(t, 2w+1) per ksw2 call. */
typedef struct pmx_align_stats {
    int64_t n_items;        /* pairs (paired) or reads aligned */
    int64_t dp_pairs;       /* items that needed at least one ksw2 DP (the rest were answered by proved shortcuts) */
    int64_t dp_calls;       /* ksw2 DPs run */
    int64_t dp_cells;       /* their cells */
    int64_t dp_rounds;      /* DP-service rounds of the thread-per-pair tier */
    int64_t wave_tier_items;     /* items handed to the wave-per-pair tiers */
    int64_t general_tier_items;  /* ... of which re-run with the general capacities */
    int64_t compact_tier_items;  /* items finished by the compact LDS tier */
    int64_t reserved[8];
} pmx_align_stats;
int pmx_align_get_stats(pmx_ctx *ctx, pmx_aligner *al, pmx_align_stats *out);
/* device pointers of the last result (for RCCL gathers without a host bounce) */
const void *pmx_align_device_records(const pmx_aligner *al);
const void *pmx_align_device_cigars(const pmx_aligner *al);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU: one rank per GPU (a process, or a host thread with its own context), RCCL over xGMI.
 * The reference is a single process; the two exchange steps follow SURVEY.md section 8e.  Reads are sharded by the
 * caller (contiguous, pair-aligned shards), the seed index and the placed genome are replicated per GPU.
 *   rank 0: pmx_dist_unique_id(id); ship the 128 bytes to every rank (a file, a pipe, MPI, a torch store ...)
 *   every rank: pmx_dist_init(ctx, id, rank, world, &d)            -- collective (ncclCommInitRank)
 *   per sample: seed the shard, pmx_dist_merge_histograms(d, pl), pmx_place_score (replicated: same result on every
 *               rank), align the shard, pmx_dist_gather_alignments(d, al, 0, ...), rank 0 writes the BAM.
 * Every call is collective over the ranks of `d` and stream-ordered on the context's stream.  librccl.so.1 is loaded on
 * first use.  PMX_DIST_HOST_DIR=<shared directory>: a file-based transport through host memory for functional tests on a
 * box with fewer GPUs than ranks (RCCL refuses two ranks on one device).
 * ------------------------------------------------------------------------------------------ */
#define PMX_DIST_ID_BYTES 128
typedef struct pmx_dist pmx_dist;
int pmx_dist_unique_id(char id[PMX_DIST_ID_BYTES]);
int pmx_dist_init(pmx_ctx *ctx, const char id[PMX_DIST_ID_BYTES], int rank, int world, pmx_dist **out);
void pmx_dist_free(pmx_dist *d);
int pmx_dist_rank(const pmx_dist *d);
int pmx_dist_world(const pmx_dist *d);
int pmx_dist_barrier(pmx_dist *d);
/* element-wise sum of every rank's n values, on every rank (--refine over shards: candidate scores add up) */
int pmx_dist_sum_i64(pmx_dist *d, int64_t *vals, int64_t n);
/* --dedup over the whole sample (src/placement.cpp:1550-1620 collapses duplicates of ALL reads): every rank thins its shard
 * (exact comparison), the ranks exchange the 128-bit hash pairs of the reads they keep, and a read whose pair a lower rank
 * keeps is dropped.  Then pmx_place_add_reads(..., dedup_reads = 1) on the same read set seeds through that mask.
 * Building blocks (also usable with another transport): pmx_place_dedup_local (mask + the kept reads' hash pairs into two
 * device arrays of n_reads entries; returns their number), pmx_place_dedup_drop_seen, pmx_place_dedup_local_count. */
int pmx_dist_dedup_reads(pmx_dist *d, pmx_place *pl, const pmx_readset *rs, int64_t *n_kept_local);
int64_t pmx_place_dedup_local(pmx_ctx *ctx, pmx_place *pl, const pmx_readset *rs, void *d_h1, void *d_h2, int64_t cap);
int pmx_place_dedup_drop_seen(pmx_ctx *ctx, pmx_place *pl, const pmx_readset *rs, void *d_seen_h1, void *d_seen_h2, int64_t n_seen);
int64_t pmx_place_dedup_local_count(pmx_ctx *ctx, pmx_place *pl, const pmx_readset *rs);
/* all-gather of the ranks' (hash, count) histograms + integer merge: afterwards every rank's placer holds the histogram of
 * the whole sample and pmx_place_score gives the same result on every rank (no floating-point reduction anywhere) */
int pmx_dist_merge_histograms(pmx_dist *d, pmx_place *pl);
/* the records and the CIGAR arena of the aligner's last call, from every rank to `root` (exact sizes, point to point); on
 * the root the records stand in rank order with cigar_off rebased onto the arenas laid back to back.  n_records / n_words:
 * totals on the root, 0 elsewhere. */
int pmx_dist_gather_alignments(pmx_dist *d, pmx_aligner *al, int root, int64_t *n_records, int64_t *n_words);
/* One-node form of step 2 (no traffic between the GPUs, no funnel through the root's PCIe link): the ranks agree on the
 * place of every rank's records / CIGAR words in ONE result set (records in rank order, arenas back to back; one small
 * all-gather of the counts) and each rank downloads its own part -- cigar_off rebased onto the merged arena on the device --
 * to that place in host buffers all ranks map (shared memory; pinned / registered for the copies to be asynchronous).
 * pmx_dist_rank_counts reports the sizes the plan exchanged. */
int pmx_dist_plan_alignments(pmx_dist *d, pmx_aligner *al, int64_t *record_base, int64_t *word_base, int64_t *total_records,
                             int64_t *total_words);
int pmx_dist_fetch_shard_async(pmx_dist *d, pmx_aligner *al, pmx_aln_record *records_all, uint32_t *cigars_all, void *stream);
const void *pmx_dist_gathered_records(const pmx_dist *d);   /* device pointers of the last gather (root) */
const void *pmx_dist_gathered_cigars(const pmx_dist *d);
int pmx_dist_rank_counts(const pmx_dist *d, int64_t *records_per_rank, int64_t *words_per_rank);   /* world entries each */
int pmx_dist_fetch_gathered(pmx_dist *d, pmx_aln_record *records, int64_t n_records, uint32_t *cigar_arena, int64_t arena_cap);
int pmx_dist_fetch_gathered_async(pmx_dist *d, pmx_aln_record *records, int64_t n_records, uint32_t *cigar_arena,
                                  int64_t arena_cap, void *stream);   /* see pmx_align_fetch_async */

/* ------------------------------------------------------------------------------------------
 * --meta: haplotype deconvolution of a mixed sample (BASELINE config 5; src/main.cpp:1192-1313 runDeconvolution,
 * src/mgsr.cpp).  Reads -> seedmer lists (k-min-mer hash + orientation), merged by list; every node's overlap coefficient
 * (:5685-5790); the nodes of the best top_oc distinct coefficients are the candidates (:8010-8060); a parsimony score per
 * (read, candidate) = max(seedmers the node's genome holds in the read's orientation, in the other one) (:7225-7455);
 * candidates with equal score columns merge; P(read | node) = err^(n - s) (1 - err)^s; SQUAREM EM (:4341-4443), nodes under
 * prop_threshold dropped, again (<= em_max_rounds, :4445-4490).  csrc/api_meta.hip says how each step runs on the device.
 * The node side is an ORIENTED index over the same tree as the place stage's index (PMX_INDEX_ORIENTED).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    double error_rate;          /* 0.005 (src/mgsr.hpp:1109) */
    double em_convergence;      /* 1e-5  --em-convergence-threshold: |log-likelihood change| that ends the EM */
    double em_delta_threshold;  /* 0     --em-delta-threshold: > 0 = stop on the largest proportion change instead */
    double prop_threshold;      /* 0.005 propThresholdToRemove (src/mgsr.hpp:1108) */
    double discard;             /* 0     --discard: reads whose best score < discard * seedmers carry no weight */
    int32_t em_max_iterations;  /* 1000 */
    int32_t em_max_rounds;      /* 5 */
    int64_t reserved[2];
} pmx_meta_params;
typedef struct pmx_meta pmx_meta;
int pmx_meta_create(pmx_ctx *ctx, const pmx_index *idx, const pmx_index *idx_oriented, pmx_meta **out);
void pmx_meta_free(pmx_ctx *ctx, pmx_meta *m);
/* reads (concatenated ASCII + n+1 offsets, mates as sequenced): seedmer lists, merge, overlap coefficients of all nodes */
int pmx_meta_set_reads(pmx_ctx *ctx, pmx_meta *m, const char *concat, const int64_t *offsets, int64_t n_reads);
/* candidates = the nodes of the top_oc best distinct overlap coefficients (1000: --top-oc), or exactly the given nodes when
 * n_override > 0; then every (merged read, candidate) parsimony score on the device */
int pmx_meta_score(pmx_ctx *ctx, pmx_meta *m, int64_t top_oc, const uint32_t *cand_override, int64_t n_override);
int pmx_meta_em(pmx_ctx *ctx, pmx_meta *m, const pmx_meta_params *mp);
int64_t pmx_meta_num_reads(const pmx_meta *m);        /* merged reads that have seedmers */
int64_t pmx_meta_num_candidates(const pmx_meta *m);
int pmx_meta_candidates(const pmx_meta *m, uint32_t *dfs_index, int64_t cap);          /* ascending */
int pmx_meta_overlap_coefficients(const pmx_meta *m, double *oc, int64_t cap);         /* per node */
int pmx_meta_read_info(const pmx_meta *m, int64_t *n_seedmers, int64_t *multiplicity, int64_t cap);
int pmx_meta_read_seedmers(const pmx_meta *m, int64_t *offsets, uint64_t *hash, uint8_t *rev, int64_t cap_seedmers);
int pmx_meta_scores(pmx_ctx *ctx, pmx_meta *m, uint16_t *scores, int64_t cap);         /* [reads][candidates] */
/* the estimated haplotypes, by proportion (descending): representative node, proportion, the candidates merged into it */
int64_t pmx_meta_num_haplotypes(const pmx_meta *m);
int pmx_meta_haplotype(const pmx_meta *m, int64_t i, uint32_t *node, double *prop, int64_t *n_members, uint32_t *members,
                       int64_t cap);
int pmx_meta_em_info(const pmx_meta *m, int32_t *rounds, int32_t *iterations, double *log_likelihood);
/* --dust T (src/main.cpp:2059-2061, default 100 = off): pmx_meta_set_reads drops every read whose DUST score is non-zero and
 * greater than T (src/mgsr.cpp:1593-1594, 1833-1834).  pmx_read_dust = mgsr::getDust (src/mgsr.cpp:1505-1568; host, no
 * device): Prinseq-scaled score over base triplets in a sliding window of `window` triplets (the reference uses 64). */
int pmx_meta_set_dust(pmx_meta *m, double threshold);
double pmx_read_dust(const char *seq, int64_t len, int32_t window);

/* The library's tuning / testing / diagnostic switches are environment variables PMX_<NAME>, all of them listed with their
 * class and meaning in ONE table (csrc/device/pmx_options.hpp).  The environment is read once, at the first use;
 * pmx_options_reload reads it again (a test that changes a switch inside one process), pmx_options_describe writes the
 * table with the current values ("PMX_NAME [class] meaning (= value)" lines) and returns the bytes it needs. */
void pmx_options_reload(void);
int64_t pmx_options_describe(char *buf, int64_t cap);

/* kernel timing: average duration (ms) of the dominant kernel of the last call, measured with HIP
   events on the context stream; name selects a stage ("pack", "seed", "score", "align") or a kernel of the align stage
   ("align_cseeds" = k_compact_seeds*, "align_dom" = the mapping kernel over every pair, "align_cmulti" = the compact
   tier's second form, k_align_compact*_multi); < 0: no such span in the last call */
double pmx_last_kernel_ms(pmx_ctx *ctx, const char *name);

#ifdef __cplusplus
}
#endif
#endif
