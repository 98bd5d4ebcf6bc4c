/*
 * bgt_synth.h -- seeded synthetic-cohort generator (bench and test tooling of libbgt_hip.so).
 *
 * The reference has no generator; its cohorts come from `bgt import` (import.c:8-120), whose PBWT
 * encoder (pbwt.c:57-66, 288-311) is strictly sequential in the row.  For benchmark-size cohorts
 * (BASELINE.json: 10k x 1M, 100k x 1M, 100k x 10M) rows are therefore drawn directly in the PBWT
 * domain -- every row is an independent run-length string, written with the reference's canonical
 * byte code (one byte per non-zero hex digit of a run, high digit first, maximal runs; pbwt.c:24-50) --
 * and the 'S' checkpoints are then derived by bgth_pbf_from_rle().  Because decode(encode(x)) = x and
 * the encoding is canonical, the resulting .pbf is byte-identical to what the reference encoder
 * would write for the decoded genotype matrix.
 *
 * Model per site (restated for the PBWT domain): ALT frequency f from a skewed spectrum
 * (half the sites rare, f = 1/(2..201)/2; half U(0,0.5)); plane 0 (low bit: ALT or <M>) carries
 * round(f*m) ones in ~sqrt(ones) clusters; plane 1 (high bit: missing or <M>) carries Binomial(m,1e-3)
 * scattered ones plus, for 5 % of the sites, a further Binomial(m,2e-2).
 * Every row depends only on (seed, row), so generation is reproducible and parallel.
 */
#ifndef BGT_SYNTH_H
#define BGT_SYNTH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bgth_synth_s bgth_synth_t;

/* Draw rows [row0, row0+n_rows) of the cohort (seed, m). n_threads<=0: all host cores. */
bgth_synth_t  *bgth_synth_rows(int m, int64_t row0, int64_t n_rows, uint64_t seed, int n_threads);
const uint8_t *bgth_synth_rle(const bgth_synth_t *s);     /* concatenated strings, row-major plane-minor */
const uint32_t *bgth_synth_len(const bgth_synth_t *s);    /* 2*n_rows lengths                            */
int64_t        bgth_synth_bytes(const bgth_synth_t *s);
void           bgth_synth_free(bgth_synth_t *s);
/* site line of row `row`: 1-based position on contig 11, REF and ALT nucleotide, 2 or 3 alleles */
void           bgth_synth_site(uint64_t seed, int64_t row, int32_t *pos1, char *ref, char *alt, int32_t *n_allele);


/* Exported by libbgt.so (host shell): write a complete synthetic BGT database prefix.{pbf,bcf,bcf.csi,spl}
 * of n_samples x n_sites (needs the device for the checkpoints). 0 on success. */
int bgt_synth_trio(const char *prefix, int n_samples, int64_t n_sites, uint64_t seed, int device);

#ifdef __cplusplus
}
#endif
#endif
