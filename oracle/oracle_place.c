/*
 * oracle_place.c -- TEST INFRASTRUCTURE ONLY (see oracle_place.h).
 * CPU restatement of the reference's place stage.  Written from the behaviour of the
 * reference (file:line cited per function); shares no code with it.
 */
#include "oracle_place.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define ORC_SUM_BLOCK 1024

/* ------------------------------------------------------------------ hashes */

static inline uint64_t rotl64(uint64_t h, unsigned r) { r &= 63u; return r ? (h << r) | (h >> (64u - r)) : h; }
static inline uint64_t rotr64(uint64_t h, unsigned r) { r &= 63u; return r ? (h >> r) | (h << (64u - r)) : h; }

/* src/seeding.hpp:100-112 : per-base constants, case-insensitive, anything else 0 */
uint64_t orc_chash(int c)
{
    switch (c) {
    case 'A': case 'a': return 0x3c8bfbb395c60474ULL;
    case 'C': case 'c': return 0x3193c18562a02b4cULL;
    case 'G': case 'g': return 0x20323ed082572324ULL;
    case 'T': case 't': return 0x295549f54be24456ULL;
    default: return 0;
    }
}

/* complement constant: chash(comp(c)); src/seeding.hpp:86-98 maps non-ACGT to 'N' -> 0 */
static inline uint64_t chash_comp(int c)
{
    switch (c) {
    case 'A': case 'a': return 0x295549f54be24456ULL;
    case 'C': case 'c': return 0x20323ed082572324ULL;
    case 'G': case 'g': return 0x3193c18562a02b4cULL;
    case 'T': case 't': return 0x3c8bfbb395c60474ULL;
    default: return 0;
    }
}

/* src/seeding.cpp:20-30 : F = xor_p rol(c(w[p]), n-1-p) ; R = xor_p rol(c(comp(w[p])), p) */
int orc_hash_seq(const char *s, int k, uint64_t *fwd, uint64_t *rev)
{
    uint64_t f = 0, r = 0;
    for (int p = 0; p < k; ++p) {
        if (orc_chash(s[p]) == 0) return -1;
        f ^= rotl64(orc_chash(s[p]), (unsigned)(k - 1 - p));
        r ^= rotl64(chash_comp(s[p]), (unsigned)p);
    }
    *fwd = f; *rev = r;
    return 0;
}

/* src/placement.cpp:41-76 : canonical hash of b^k */
uint64_t orc_homopolymer_hash(int base, int k)
{
    uint64_t b = orc_chash(base), c = chash_comp(base), f = 0, r = 0;
    if (b == 0) return 0;
    for (int i = 0; i < k; ++i) { f ^= rotl64(b, (unsigned)(k - i - 1)); r ^= rotl64(c, (unsigned)(k - i - 1)); }
    return f < r ? f : r;
}

/* ---------------------------------------------------------------- syncmers */

/*
 * src/seeding.cpp:47-229.  For the k-mer starting at i let fS[j], rS[j] be the forward /
 * reverse-complement hashes of the s-mers starting at j in [i, i+k-s].  The k-mer is a
 * syncmer iff (closed) fS[i+t]==min fS or fS[i+k-s-t]==min fS, or the same test on rS
 * (:205-211); open syncmers test fS[i+t] and rS[i+k-s-t] only (:202-204, the reverse ring is
 * indexed from the newest s-mer).  Windows holding a non-ACGT base (:196-197) or with
 * F(kmer)==R(kmer) (:219-221) emit nothing.
 */
int64_t orc_rolling_syncmers(const char *seq, int64_t len, int k, int s, int open, int t, int return_all,
                             uint64_t *hash, uint8_t *is_rev, uint8_t *is_sync, int64_t *pos)
{
    if (len < k || k <= 0 || s <= 0 || s > k) return 0;
    const int64_t ns = len - s + 1;   /* number of s-mers */
    const int64_t nk = len - k + 1;   /* number of k-mers */
    const int w = k - s + 1;
    uint64_t *fS = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)ns * 2);
    uint64_t *rS = fS + ns;
    int64_t n = 0;

    /* s-mer hashes by rolling (src/seeding.cpp:182-183) */
    {
        uint64_t f = 0, r = 0;
        for (int p = 0; p < s; ++p) {
            f ^= rotl64(orc_chash(seq[p]), (unsigned)(s - 1 - p));
            r ^= rotl64(chash_comp(seq[p]), (unsigned)p);
        }
        fS[0] = f; rS[0] = r;
        for (int64_t j = 1; j < ns; ++j) {
            int out = seq[j - 1], in = seq[j + s - 1];
            f = rotl64(f, 1) ^ rotl64(orc_chash(out), (unsigned)s) ^ orc_chash(in);
            r = rotr64(r, 1) ^ rotr64(chash_comp(out), 1) ^ rotl64(chash_comp(in), (unsigned)(s - 1));
            fS[j] = f; rS[j] = r;
        }
    }

    uint64_t fK = 0, rK = 0;
    int64_t last_amb = -1;
    for (int p = 0; p < k; ++p) {
        fK ^= rotl64(orc_chash(seq[p]), (unsigned)(k - 1 - p));
        rK ^= rotl64(chash_comp(seq[p]), (unsigned)p);
        if (orc_chash(seq[p]) == 0) last_amb = p;
    }
    for (int64_t i = 0; i < nk; ++i) {
        if (i > 0) { /* src/seeding.cpp:180-181 */
            int out = seq[i - 1], in = seq[i + k - 1];
            fK = rotl64(fK, 1) ^ rotl64(orc_chash(out), (unsigned)k) ^ orc_chash(in);
            rK = rotr64(rK, 1) ^ rotr64(chash_comp(out), 1) ^ rotl64(chash_comp(in), (unsigned)(k - 1));
            if (orc_chash(in) == 0) last_amb = i + k - 1;
        }
        int emit = 0;
        if (last_amb < i) { /* no ambiguous base inside [i, i+k) */
            uint64_t fmin = UINT64_MAX, rmin = UINT64_MAX;
            for (int j = 0; j < w; ++j) {
                if (fS[i + j] < fmin) fmin = fS[i + j];
                if (rS[i + j] < rmin) rmin = rS[i + j];
            }
            int fsync, rsync;
            if (open) {
                fsync = fS[i + t] == fmin;
                rsync = rS[i + k - s - t] == rmin;
            } else {
                fsync = fS[i + t] == fmin || fS[i + k - s - t] == fmin;
                rsync = rS[i + k - s - t] == rmin || rS[i + t] == rmin;
            }
            if ((fsync || rsync) && fK != rK) emit = 1;
        }
        if (emit) {
            hash[n] = fK < rK ? fK : rK;
            if (is_rev) is_rev[n] = rK < fK;
            if (is_sync) is_sync[n] = 1;
            if (pos) pos[n] = i;
            ++n;
        } else if (return_all) {
            hash[n] = UINT64_MAX;
            if (is_rev) is_rev[n] = 0;
            if (is_sync) is_sync[n] = 0;
            if (pos) pos[n] = i;
            ++n;
        }
    }
    free(fS);
    return n;
}

/* ------------------------------------------------------------- read seeds */

/*
 * src/placement.cpp:1611-1682.  Syncmers restricted to k-mer starts in
 * [trim_start, len-trim_end-k] (a contiguous sub-range, :1643-1648); l==1 (and the l==0 raw
 * syncmer mode, :1339-1366): seed = syncmer hash; l>1: every window of l consecutive syncmers
 * gives F = xor_q rol(h[j+q], k*(l-1-q)), R = xor_q rol(h[j+q], k*q), seed = min(F,R) if F != R.
 * The reference skips the read when it has fewer than l syncmers before trimming (:1626).
 */
int64_t orc_read_seeds(const char *seq, int64_t len, int k, int s, int l, int open, int t,
                       int trim_start, int trim_end, uint64_t *out)
{
    if (len < k) return 0;
    const int64_t nk = len - k + 1;
    uint64_t *h = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nk);
    int64_t *p = (int64_t *)malloc(sizeof(int64_t) * (size_t)nk);
    int64_t m = orc_rolling_syncmers(seq, len, k, s, open, t, 0, h, NULL, NULL, p);
    int64_t n = 0;
    const int64_t valid_start = trim_start, valid_end = len - trim_end - k;
    if (l <= 1) {
        if (l == 1 && m < 1) { free(h); free(p); return 0; }
        for (int64_t j = 0; j < m; ++j)
            if ((int)p[j] >= valid_start && (int)p[j] <= valid_end) out[n++] = h[j];
    } else if (m >= l) {
        int64_t lo = 0, hi = m;
        while (lo < hi && (int)p[lo] < valid_start) ++lo;
        while (hi > lo && (int)p[hi - 1] > valid_end) --hi;
        for (int64_t j = lo; j + l <= hi; ++j) {
            uint64_t F = 0, R = 0;
            for (int q = 0; q < l; ++q) {
                F ^= rotl64(h[j + q], (unsigned)(k * (l - 1 - q)));
                R ^= rotl64(h[j + q], (unsigned)(k * q));
            }
            if (F != R) out[n++] = F < R ? F : R;
        }
    }
    free(h); free(p);
    return n;
}

/*
 * Quality-filtered seeds of one read (src/placement.cpp:1386-1527, the `minSeedQuality > 0` branch; no read
 * dedup there).  avgPhredQuality (:79-88) = sum(qual - 33) / k over the k-mer, 0 when the quality string does not
 * cover it; a syncmer passes when its start is inside the trim window AND avg >= min_q.  l == 1: passing
 * syncmers are the seeds.  l > 1: windows run over the ORIGINAL syncmer list (trimmed / low-quality syncmers are
 * not removed first, unlike the unfiltered path) and a window yields a seed only if all its l syncmers pass.
 */
int64_t orc_read_seeds_q(const char *seq, const char *qual, int64_t len, int64_t qual_len, int k, int s, int l,
                         int open, int t, int trim_start, int trim_end, int min_q, uint64_t *out)
{
    if (len < k) return 0;
    const int64_t nk = len - k + 1;
    uint64_t *h = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nk);
    int64_t *p = (int64_t *)malloc(sizeof(int64_t) * (size_t)nk);
    uint8_t *pass = (uint8_t *)malloc((size_t)nk);
    int64_t m = orc_rolling_syncmers(seq, len, k, s, open, t, 0, h, NULL, NULL, p);
    int64_t n = 0;
    const int valid_start = trim_start, valid_end = (int)len - trim_end - k;
    if (l < 1) l = 1;
    if (m >= l) {
        for (int64_t j = 0; j < m; ++j) {
            double avg = 0.0;
            if (qual_len > 0 && p[j] >= 0 && p[j] + k <= qual_len) {
                int64_t sum = 0;
                for (int i = 0; i < k; ++i) sum += (int)qual[p[j] + i] - 33;
                avg = (double)sum / k;
            }
            const int in_range = (int)p[j] >= valid_start && (int)p[j] <= valid_end;
            pass[j] = in_range && avg >= (double)min_q;
        }
        if (l == 1) {
            for (int64_t j = 0; j < m; ++j)
                if (pass[j]) out[n++] = h[j];
        } else {
            for (int64_t j = 0; j + l <= m; ++j) {
                int ok = 1;
                for (int q = 0; q < l; ++q) ok &= pass[j + q];
                if (!ok) continue;
                uint64_t F = 0, R = 0;
                for (int q = 0; q < l; ++q) {
                    F = rotl64(F, (unsigned)k) ^ h[j + q];
                    R = rotl64(R, (unsigned)k) ^ h[j + l - q - 1];
                }
                if (F != R) out[n++] = F < R ? F : R;
            }
        }
    }
    free(h); free(p); free(pass);
    return n;
}

/* --------------------------------------------------------------- histogram */

struct orc_hist {
    uint64_t *key;
    int64_t *val;
    uint8_t *used;
    uint64_t cap, n;
    uint64_t *tmp;
    int64_t tmp_cap;
};

static inline uint64_t mix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

orc_hist *orc_hist_new(void)
{
    orc_hist *h = (orc_hist *)calloc(1, sizeof(*h));
    h->cap = 1u << 16;
    h->key = (uint64_t *)malloc(h->cap * sizeof(uint64_t));
    h->val = (int64_t *)malloc(h->cap * sizeof(int64_t));
    h->used = (uint8_t *)calloc(h->cap, 1);
    return h;
}

void orc_hist_free(orc_hist *h)
{
    if (!h) return;
    free(h->key); free(h->val); free(h->used); free(h->tmp); free(h);
}

static void hist_grow(orc_hist *h)
{
    uint64_t ocap = h->cap;
    uint64_t *ok = h->key; int64_t *ov = h->val; uint8_t *ou = h->used;
    h->cap = ocap * 2;
    h->key = (uint64_t *)malloc(h->cap * sizeof(uint64_t));
    h->val = (int64_t *)malloc(h->cap * sizeof(int64_t));
    h->used = (uint8_t *)calloc(h->cap, 1);
    for (uint64_t i = 0; i < ocap; ++i) {
        if (!ou[i]) continue;
        uint64_t j = mix64(ok[i]) & (h->cap - 1);
        while (h->used[j]) j = (j + 1) & (h->cap - 1);
        h->used[j] = 1; h->key[j] = ok[i]; h->val[j] = ov[i];
    }
    free(ok); free(ov); free(ou);
}

void orc_hist_add(orc_hist *h, uint64_t key, int64_t mult)
{
    if ((h->n + 1) * 10 > h->cap * 6) hist_grow(h);
    uint64_t j = mix64(key) & (h->cap - 1);
    while (h->used[j]) {
        if (h->key[j] == key) { h->val[j] += mult; return; }
        j = (j + 1) & (h->cap - 1);
    }
    h->used[j] = 1; h->key[j] = key; h->val[j] = mult; ++h->n;
}

void orc_hist_add_read(orc_hist *h, const char *seq, int64_t len, int k, int s, int l, int open, int t,
                       int trim_start, int trim_end, int64_t multiplicity)
{
    if (len <= 0) return;
    if (h->tmp_cap < len) {
        free(h->tmp);
        h->tmp_cap = len * 2 + 64;
        h->tmp = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)h->tmp_cap);
    }
    int64_t n = orc_read_seeds(seq, len, k, s, l, open, t, trim_start, trim_end, h->tmp);
    for (int64_t i = 0; i < n; ++i) orc_hist_add(h, h->tmp[i], multiplicity);
}

void orc_hist_add_read_q(orc_hist *h, const char *seq, const char *qual, int64_t len, int64_t qual_len, int k, int s, int l,
                         int open, int t, int trim_start, int trim_end, int min_q)
{
    if (h->tmp_cap < len + 1) {
        free(h->tmp);
        h->tmp_cap = len * 2 + 64;
        h->tmp = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)h->tmp_cap);
    }
    int64_t n = orc_read_seeds_q(seq, qual, len, qual_len, k, s, l, open, t, trim_start, trim_end, min_q, h->tmp);
    for (int64_t i = 0; i < n; ++i) orc_hist_add(h, h->tmp[i], 1);
}

/* Batch entry for the CPU baseline (bench.py): reads [r0, r1) of one flat buffer (concat + n+1 offsets) go into
 * the histogram in ONE C call, so a timing around it measures the seeding, not a Python loop. */
void orc_hist_add_reads(orc_hist *h, const char *concat, const int64_t *off, int64_t r0, int64_t r1, int k, int s, int l,
                        int open, int t, int trim_start, int trim_end)
{
    for (int64_t r = r0; r < r1; ++r)
        orc_hist_add_read(h, concat + off[r], off[r + 1] - off[r], k, s, l, open, t, trim_start, trim_end, 1);
}

/* mergeSeedMaps (src/placement.cpp:922-929): add every (key, count) of src into dst */
void orc_hist_merge(orc_hist *dst, const orc_hist *src)
{
    for (uint64_t j = 0; j < src->cap; ++j)
        if (src->used[j]) orc_hist_add(dst, src->key[j], src->val[j]);
}

/* The reference seeds the reads on all cores with per-thread maps merged afterwards (src/placement.cpp:1611-1686:
 * tbb::parallel_for over the unique reads, then mergeSeedMaps): the same shape with pthreads, one call. */
typedef struct { orc_hist *h; const char *concat; const int64_t *off; int64_t r0, r1; int k, s, l, open, t, ts, te; } hist_job;
static void *hist_worker(void *p)
{
    hist_job *j = (hist_job *)p;
    orc_hist_add_reads(j->h, j->concat, j->off, j->r0, j->r1, j->k, j->s, j->l, j->open, j->t, j->ts, j->te);
    return NULL;
}
orc_hist *orc_hist_build_mt(const char *concat, const int64_t *off, int64_t n_reads, int k, int s, int l, int open, int t,
                            int trim_start, int trim_end, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if ((int64_t)n_threads > n_reads) n_threads = n_reads > 0 ? (int)n_reads : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    hist_job *jobs = (hist_job *)malloc(sizeof(hist_job) * (size_t)n_threads);
    for (int i = 0; i < n_threads; ++i) {
        hist_job j = {orc_hist_new(), concat, off, n_reads * i / n_threads, n_reads * (i + 1) / n_threads, k, s, l, open, t, trim_start, trim_end};
        jobs[i] = j;
        pthread_create(&th[i], NULL, hist_worker, &jobs[i]);
    }
    for (int i = 0; i < n_threads; ++i) pthread_join(th[i], NULL);
    orc_hist *out = jobs[0].h;
    for (int i = 1; i < n_threads; ++i) { orc_hist_merge(out, jobs[i].h); orc_hist_free(jobs[i].h); }
    free(th); free(jobs);
    return out;
}

int64_t orc_hist_size(const orc_hist *h) { return (int64_t)h->n; }

typedef struct { uint64_t k; int64_t v; } kv_t;
static int kv_cmp(const void *a, const void *b)
{
    uint64_t x = ((const kv_t *)a)->k, y = ((const kv_t *)b)->k;
    return x < y ? -1 : x > y;
}

void orc_hist_export_sorted(const orc_hist *h, uint64_t *hash, int64_t *count)
{
    kv_t *a = (kv_t *)malloc(sizeof(kv_t) * (size_t)(h->n ? h->n : 1));
    uint64_t n = 0;
    for (uint64_t i = 0; i < h->cap; ++i)
        if (h->used[i]) { a[n].k = h->key[i]; a[n].v = h->val[i]; ++n; }
    qsort(a, n, sizeof(kv_t), kv_cmp);
    for (uint64_t i = 0; i < n; ++i) { hash[i] = a[i].k; count[i] = a[i].v; }
    free(a);
}

/* ----------------------------------------------------- read-side finalise */

typedef struct { uint64_t k; int64_t v; } cnt_t;
static int cnt_desc_cmp(const void *a, const void *b)
{   /* count descending; equal counts ordered by ascending hash: the reference sorts by count
       only (src/placement.cpp:1757-1759, order of equal counts unspecified) -- canonicalised. */
    const cnt_t *x = (const cnt_t *)a, *y = (const cnt_t *)b;
    if (x->v != y->v) return x->v > y->v ? -1 : 1;
    return x->k < y->k ? -1 : x->k > y->k;
}

int64_t orc_finalize_reads(const uint64_t *hash, const int64_t *count, int64_t n, int k,
                           double mask_fraction, int min_support_cfg,
                           uint64_t *kept_hash, double *kept_log, orc_read_state *st)
{
    uint8_t *dead = (uint8_t *)calloc((size_t)(n ? n : 1), 1);
    int64_t alive = n;
    /* homopolymer seeds: src/placement.cpp:1708-1722 */
    const char bases[4] = {'A', 'C', 'G', 'T'};
    for (int b = 0; b < 4; ++b) {
        uint64_t hh = orc_homopolymer_hash(bases[b], k);
        int64_t lo = 0, hi = n - 1;
        while (lo <= hi) {
            int64_t mid = (lo + hi) >> 1;
            if (hash[mid] < hh) lo = mid + 1; else if (hash[mid] > hh) hi = mid - 1;
            else { if (!dead[mid]) { dead[mid] = 1; --alive; } break; }
        }
    }
    /* top-fraction mask: src/placement.cpp:1774-1799 ; numToMask = (size_t)(frac * uniqueSeeds) */
    if (mask_fraction > 0.0 && alive > 0) {
        int64_t num = (int64_t)(mask_fraction * (double)alive);
        if (num > 0) {
            cnt_t *a = (cnt_t *)malloc(sizeof(cnt_t) * (size_t)alive);
            int64_t m = 0;
            for (int64_t i = 0; i < n; ++i) if (!dead[i]) { a[m].k = hash[i]; a[m].v = count[i]; ++m; }
            qsort(a, (size_t)m, sizeof(cnt_t), cnt_desc_cmp);
            for (int64_t i = 0; i < num && i < m; ++i) {
                int64_t lo = 0, hi = n - 1;
                while (lo <= hi) {
                    int64_t mid = (lo + hi) >> 1;
                    if (hash[mid] < a[i].k) lo = mid + 1; else if (hash[mid] > a[i].k) hi = mid - 1;
                    else { dead[mid] = 1; --alive; break; }
                }
            }
            free(a);
        }
    }
    /* resolveMinReadSupport: src/placement.cpp:931-955 */
    int64_t min_support = min_support_cfg;
    double est_cov = 0.0;
    if (min_support < 0) {
        uint64_t sum = 0, cnt = 0;
        for (int64_t i = 0; i < n; ++i)
            if (!dead[i] && count[i] >= 2) { sum += (uint64_t)count[i]; ++cnt; }
        est_cov = cnt > 0 ? (double)sum / (double)cnt : 0.0;
        min_support = est_cov > 3.0 ? 2 : 1;
    }
    /* computeReadSeedMagnitudes: src/placement.cpp:957-984.  The reference sums in hash-map iteration
       order (unspecified); canonical order here and on the GPU: kept seeds in ascending-hash order, summed
       sequentially inside consecutive blocks of ORC_SUM_BLOCK, then the block sums sequentially. */
    int64_t kept = 0, total = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (dead[i]) continue;
        total += count[i];
        if (count[i] < min_support) continue;
        kept_hash[kept] = hash[i];
        kept_log[kept] = log1p((double)count[i]);
        ++kept;
    }
    double mag2 = 0.0, lsum = 0.0;
    for (int64_t b = 0; b < kept; b += ORC_SUM_BLOCK) {
        double m2 = 0.0, ls = 0.0;
        const int64_t e = b + ORC_SUM_BLOCK < kept ? b + ORC_SUM_BLOCK : kept;
        for (int64_t i = b; i < e; ++i) { const double L = kept_log[i]; m2 += L * L; ls += L; }
        mag2 += m2;
        lsum += ls;
    }
    free(dead);
    st->min_support = min_support;
    st->n_unique_in = alive;
    st->n_kept = kept;
    st->total_freq = total;
    st->log_magnitude = sqrt(mag2);
    st->log_cont_den = lsum;
    st->est_coverage = est_cov;
    return kept;
}

/* --------------------------------------------------------- node scoring */

static inline int64_t find_kept(const uint64_t *kh, int64_t n, uint64_t x)
{
    int64_t lo = 0, hi = n - 1;
    while (lo <= hi) {
        int64_t mid = (lo + hi) >> 1;
        if (kh[mid] < x) lo = mid + 1; else if (kh[mid] > x) hi = mid - 1; else return mid;
    }
    return -1;
}

void orc_score_nodes(int64_t n_nodes, const uint32_t *parent, const uint64_t *offsets,
                     const uint64_t *ch_hash, const int16_t *ch_par, const int16_t *ch_child,
                     int64_t n_kept, const uint64_t *kept_hash, const double *kept_log,
                     double log_magnitude, double log_cont_den,
                     double *wc_den_out, double *metrics5, int64_t *counts2, double *scores5)
{
    /* weighted-containment denominator from the root's changes in stored order:
       src/placement.cpp:1863-1876 */
    double wc_den = 0.0;
    if (n_nodes > 0) {
        for (uint64_t i = offsets[0]; i < offsets[1]; ++i)
            if (ch_child[i] > 0 && find_kept(kept_hash, n_kept, ch_hash[i]) >= 0)
                wc_den += 1.0 / (double)ch_child[i];
    }
    if (wc_den_out) *wc_den_out = wc_den;

    for (int64_t nd = 0; nd < n_nodes; ++nd) {
        double m[5]; int64_t c[2];
        if (nd == 0) { memset(m, 0, sizeof m); c[0] = c[1] = 0; }   /* rootMetrics = 0, :1878 */
        else { memcpy(m, metrics5 + 5 * (int64_t)parent[nd], sizeof m); memcpy(c, counts2 + 2 * (int64_t)parent[nd], sizeof c); }
        /* computeChildMetrics: src/placement.cpp:242-345, changes in stored order */
        for (uint64_t i = offsets[nd]; i < offsets[nd + 1]; ++i) {
            const int64_t pc = ch_par[i], cc = ch_child[i];
            const double logC = cc > 0 ? log1p((double)cc) : 0.0;
            const double logP = pc > 0 ? log1p((double)pc) : 0.0;
            m[4] += logC * logC - logP * logP;
            c[1] += (cc > 0) - (pc > 0);
            if (cc == pc) continue;
            int64_t ki = find_kept(kept_hash, n_kept, ch_hash[i]);
            if (ki < 0) continue;
            const double L = kept_log[ki];
            const int64_t pd = (int64_t)((pc == 0) & (cc != 0)) - (int64_t)((cc == 0) & (pc != 0));
            c[0] += pd;
            {
                const double o = pc > 0 ? L / (double)pc : 0.0;
                const double nw = cc > 0 ? L / (double)cc : 0.0;
                m[0] += nw - o;
            }
            m[1] += L * (logC - logP);
            {
                const double o = pc > 0 ? 1.0 / (double)pc : 0.0;
                const double nw = cc > 0 ? 1.0 / (double)cc : 0.0;
                m[2] += nw - o;
            }
            m[3] += (double)pd * L;
        }
        memcpy(metrics5 + 5 * nd, m, sizeof m);
        memcpy(counts2 + 2 * nd, c, sizeof c);
        /* getters: src/placement.hpp:120-149 */
        double *sc = scores5 + 5 * nd;
        sc[0] = log_magnitude <= 0.0 ? 0.0 : m[0] / log_magnitude;
        {
            double gm = sqrt(m[4]);
            if (log_magnitude <= 0.0 || gm <= 0.0) sc[1] = 0.0;
            else {
                double v = m[1] / (log_magnitude * gm);
                sc[1] = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
            }
        }
        /* presenceIntersectionCount is size_t in the reference: convert through uint64 */
        sc[2] = n_kept > 0 ? (double)(uint64_t)c[0] / (double)(uint64_t)n_kept : 0.0;
        sc[3] = wc_den > 0.0 ? m[2] / wc_den : 0.0;
        sc[4] = log_cont_den > 0.0 ? m[3] / log_cont_den : 0.0;
    }
}

/* ------------------------------------------------------------- best / ties */

typedef struct { double best; uint32_t idx; uint32_t *tie; int64_t n, cap; } best_t;

/* src/placement.cpp:355-371 */
static void best_update(best_t *b, uint32_t node, double score)
{
    double tol = b->best * 0.0001;
    if (tol < 1e-9) tol = 1e-9;
    if (score > b->best + tol) {
        b->best = score; b->idx = node; b->n = 0;
        if (b->n < b->cap) b->tie[b->n] = node;
        b->n = 1;
    } else if (score >= b->best - tol && score > 0) {
        if (b->n == 0 || b->tie[(b->n - 1 < b->cap ? b->n - 1 : b->cap - 1)] != b->idx) {
            if (b->n < b->cap) b->tie[b->n] = b->idx;
            ++b->n;
        }
        if (node != b->idx) {
            if (b->n < b->cap) b->tie[b->n] = node;
            ++b->n;
        }
    }
}

static int u32_cmp(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

void orc_best_ties(int64_t n_nodes, const uint32_t *parent, const double *scores5, int force_leaf,
                   double *best_score, uint32_t *best_idx, uint32_t *tie_out, int64_t tie_cap,
                   int64_t *n_tie_out)
{
    /* BFS visit order of the single-threaded traversal = ascending (depth, DFS index):
       children are appended in DFS-index order (src/panmap_utils.cpp:283) and a pre-order
       numbering keeps subtrees contiguous. */
    int32_t *depth = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_nodes ? n_nodes : 1));
    uint8_t *has_child = (uint8_t *)calloc((size_t)(n_nodes ? n_nodes : 1), 1);
    int32_t maxd = 0;
    for (int64_t i = 0; i < n_nodes; ++i) {
        depth[i] = i == 0 ? 0 : depth[parent[i]] + 1;
        if (i > 0) has_child[parent[i]] = 1;
        if (depth[i] > maxd) maxd = depth[i];
    }
    int64_t *start = (int64_t *)calloc((size_t)maxd + 2, sizeof(int64_t));
    for (int64_t i = 0; i < n_nodes; ++i) ++start[depth[i] + 1];
    for (int32_t d = 0; d <= maxd; ++d) start[d + 1] += start[d];
    uint32_t *order = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n_nodes ? n_nodes : 1));
    {
        int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * ((size_t)maxd + 2));
        memcpy(fill, start, sizeof(int64_t) * ((size_t)maxd + 2));
        for (int64_t i = 0; i < n_nodes; ++i) order[fill[depth[i]]++] = (uint32_t)i;
        free(fill);
    }
    best_t b[5];
    for (int m = 0; m < 5; ++m) { b[m].best = 0.0; b[m].idx = UINT32_MAX; b[m].tie = tie_out + (int64_t)m * tie_cap; b[m].n = 0; b[m].cap = tie_cap; }
    for (int64_t j = 0; j < n_nodes; ++j) {
        uint32_t nd = order[j];
        if (force_leaf && has_child[nd]) continue;   /* src/placement.cpp:794-795 */
        for (int m = 0; m < 5; ++m) best_update(&b[m], nd, scores5[5 * (int64_t)nd + m]);
    }
    for (int m = 0; m < 5; ++m) {
        int64_t n = b[m].n < tie_cap ? b[m].n : tie_cap;
        /* finalizeTiedIndices: src/placement.cpp:395-401 */
        if (n > 0) {
            qsort(b[m].tie, (size_t)n, sizeof(uint32_t), u32_cmp);
            int64_t u = 1;
            for (int64_t i = 1; i < n; ++i) if (b[m].tie[i] != b[m].tie[u - 1]) b[m].tie[u++] = b[m].tie[i];
            n = u;
            b[m].idx = b[m].tie[0];
        }
        best_score[m] = b[m].best;
        best_idx[m] = b[m].idx;
        n_tie_out[m] = n;
    }
    free(depth); free(has_child); free(start); free(order);
}
