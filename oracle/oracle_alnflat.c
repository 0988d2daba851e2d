/* ORACLE / TEST INFRASTRUCTURE ONLY -- never linked into the product.
 * Flattens what align_reads_direct (the compiled reference, oracle/_ref) left in its caller-owned result array
 * (layout: src/mm_align.h:20-37) into plain arrays a numpy comparison can take at 10^7 reads:
 * per read the eight scalars, per pair / single read the `mapped` flag, and the malloc'd CIGARs in record order in one
 * arena (each freed once copied, as the caller of align_reads_direct has to, src/conversion.cpp:531-536). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {            /* read_align_t, src/mm_align.h:20-31 */
    int32_t pos, rs, re, qs, qe;
    uint8_t mapq, rev, proper_frag;
    int32_t n_cigar;
    uint32_t *cigar;
    char *md;
} orc_read_align;
typedef struct {            /* align_pair_result_t, src/mm_align.h:34-38 */
    orc_read_align r1, r2;
    int mapped;
} orc_pair_result;

/* total CIGAR operations of all reads (size of the arena the second call needs) */
int64_t orc_align_cigar_total(const orc_pair_result *res, int64_t n_res, int paired) {
    int64_t tot = 0;
    for (int64_t i = 0; i < n_res; ++i) {
        if (res[i].r1.cigar) tot += res[i].r1.n_cigar;
        if (paired && res[i].r2.cigar) tot += res[i].r2.n_cigar;
    }
    return tot;
}

/* fields: [n_reads][8] = pos rs re qs qe mapq rev proper_frag; n_cig: [n_reads]; mapped: [n_res]; arena: cap words.
 * Frees every CIGAR (and md) and clears the pointers.  Returns the words written, -1 when the arena is too small. */
int64_t orc_align_flatten(orc_pair_result *res, int64_t n_res, int paired, int32_t *fields, int32_t *n_cig, uint8_t *mapped, uint32_t *arena,
                          int64_t cap) {
    int64_t used = 0, r = 0;
    for (int64_t i = 0; i < n_res; ++i) {
        mapped[i] = (uint8_t)(res[i].mapped != 0);
        for (int m = 0; m < (paired ? 2 : 1); ++m, ++r) {
            orc_read_align *a = m ? &res[i].r2 : &res[i].r1;
            int32_t *f = fields + 8 * r;
            f[0] = a->pos; f[1] = a->rs; f[2] = a->re; f[3] = a->qs; f[4] = a->qe; f[5] = a->mapq; f[6] = a->rev; f[7] = a->proper_frag;
            const int32_t n = a->cigar ? a->n_cigar : 0;
            n_cig[r] = n;
            if (used + n > cap) return -1;
            if (n > 0) memcpy(arena + used, a->cigar, (size_t)n * sizeof(uint32_t));
            used += n;
            free(a->cigar); a->cigar = NULL;
            free(a->md); a->md = NULL;
        }
    }
    return used;
}

/* The argument arrays of align_reads_direct from one flat read buffer (concat + n+1 offsets): NUL-terminated copies in
 * `buf` (sum of lengths + n bytes), mate 2 of every pair reverse-complemented as seeding::readFastqPaired hands it over
 * (src/seeding.cpp:231-284: only upper-case A/C/G/T are complemented, anything else is kept), pointers and lengths filled. */
void orc_align_prepare_reads(const char *concat, const int64_t *off, int64_t n, int revcomp_mate2, char *buf, char **ptrs, int *lens) {
    int64_t w = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t len = off[i + 1] - off[i];
        const char *s = concat + off[i];
        char *d = buf + w;
        if (revcomp_mate2 && (i & 1)) {
            for (int64_t j = 0; j < len; ++j) {
                char c = s[len - 1 - j];
                d[j] = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : c;
            }
        } else {
            memcpy(d, s, (size_t)len);
        }
        d[len] = 0;
        ptrs[i] = d;
        lens[i] = (int)len;
        w += len + 1;
    }
}
