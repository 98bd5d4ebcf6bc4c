/*
 * orc.h -- CPU ORACLE for the BGT genotype-matrix read path.           TEST INFRASTRUCTURE ONLY.
 *
 * This is a from-scratch scalar C restatement of the arithmetic the reference (lh3/bgt) performs on
 * the path  pbf_seek/pbf_read -> pbc_dec | pbs_dec -> bgtm_cal_info.  It exists to CHECK the HIP
 * implementation; it is never linked into, imported by, or called from the product library
 * (bgt_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against fixtures
 * produced by the compiled reference (oracle/_ref, built from /root/reference by oracle/Makefile;
 * generating script tests/golden/make_golden.py) and, when oracle/_ref is present, against the
 * reference binaries run live on fresh random inputs.
 *
 * Every function cites the reference file:line whose behaviour it restates.
 */
#ifndef ORC_H
#define ORC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------- run-length byte code (reference pbwt.c:12-36) ---------- */
uint32_t orc_rle_len(uint8_t byte);                       /* length carried by one code byte       */
int      orc_rle_put_run(uint8_t *dst, uint32_t len, int bit); /* bytes written (>=1)              */
int      orc_rle_encode(int m, const uint8_t *bits, uint8_t *dst); /* whole row; dst needs m+1 B   */
int64_t  orc_rle_count_ones(const uint8_t *rle, int n);   /* sum of lengths of 1-runs              */

/* ---------- full-row codec (reference pbwt.c:57-119) ---------- */
typedef struct {
    int32_t  m;
    int32_t *perm;      /* S_k   : perm[j] = column at PBWT rank j after the last row */
    int32_t *prev;      /* S_k-1 : scratch, swapped with perm on every row            */
    uint8_t *bits;      /* m+1 bytes: last decoded row A_k (decode) / B_k then RLE (encode) */
} orc_codec_t;

orc_codec_t *orc_codec_new(int m);
void         orc_codec_free(orc_codec_t *c);
void         orc_codec_decode(orc_codec_t *c, const uint8_t *rle, int n); /* -> c->bits (column order) */
int          orc_codec_encode(orc_codec_t *c, const uint8_t *a, uint8_t *dst); /* -> RLE length     */

/* ---------- subset decoder by rank tracking (reference pbwt.c:129-170, 340-347) ---------- */
typedef struct { uint32_t rank, slot; } orc_track_t;
void orc_track_init(int m, const int32_t *perm, int n_sub, const int32_t *cols, orc_track_t *t);
void orc_track_decode(int m, int n_sub, orc_track_t *t, const uint8_t *rle, int n, uint8_t *a);

/* ---------- PBF container, in memory (reference pbwt.c:176-388) ---------- */
typedef struct orc_pbf_s orc_pbf_t;
orc_pbf_t *orc_pbf_open(const uint8_t *buf, size_t len);  /* borrows buf; NULL on bad magic        */
void       orc_pbf_close(orc_pbf_t *p);
int        orc_pbf_m(const orc_pbf_t *p);
int        orc_pbf_g(const orc_pbf_t *p);
int        orc_pbf_shift(const orc_pbf_t *p);
int64_t    orc_pbf_n(const orc_pbf_t *p);
int        orc_pbf_subset(orc_pbf_t *p, int n_sub, const int32_t *cols);
int        orc_pbf_seek(orc_pbf_t *p, int64_t row);
const uint8_t **orc_pbf_read(orc_pbf_t *p);               /* g plane pointers or NULL at the end   */
int64_t    orc_pbf_tell(const orc_pbf_t *p);              /* index of the next row to be read      */
const int32_t *orc_pbf_perm(const orc_pbf_t *p, int plane);
int        orc_pbf_subset_width(const orc_pbf_t *p);        /* columns per returned plane            */

/* PBF writer to a growable memory buffer (reference pbwt.c:199-219, 264-311) */
typedef struct orc_pbw_s orc_pbw_t;
orc_pbw_t *orc_pbw_new(int m, int g, int shift);
int        orc_pbw_row(orc_pbw_t *w, uint8_t *const *planes);
size_t     orc_pbw_finish(orc_pbw_t *w, uint8_t **out);  /* caller frees *out; w is consumed       */

/* ---------- allele-count reduction (reference bgt.c:735-757) ---------- */
/* a0/a1: byte per haplotype (0/1), n_hap = 2*n_out. group: 1-based group id per SAMPLE or NULL.
 * out layout: [0]=AN [1]=AC [2]=AC<M>, then per group g: [3+3g]=AN_g [4+3g]=AC_g [5+3g]=AC<M>_g   */
void orc_allele_counts(int n_hap, const uint8_t *a0, const uint8_t *a1,
                       const uint32_t *group, int n_groups, int32_t *out);

/* Whole scan = the loop view.c:151 drives with -G: for rows [row0,row1) decode both planes of the
 * (sub)set and reduce. counts: int32[(row1-row0)][3*(1+G)] (G = n_groups if n_groups>1 else 0).
 * gt (optional): 2-bit codes a1<<1|a0, 4 per byte, low bits first, (n_hap+3)/4 bytes per row.
 * Returns rows processed or <0. */
int64_t orc_scan(orc_pbf_t *p, int64_t row0, int64_t row1, const uint32_t *group, int n_groups,
                 int32_t *counts, uint8_t *gt);

#ifdef __cplusplus
}
#endif
#endif
