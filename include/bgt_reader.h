/*
 * bgt_reader.h -- the reader API of BGT (drop-in for the reference's bgt.h:83-123), served by the
 * MI355X path.  `#include "bgt.h"` of this repository resolves here.
 *
 * Same entry points, argument meaning, return conventions and -- because bgt-server.go reads fields
 * directly (bgt-server.go:326-355) -- the same struct layouts as the reference on x86-64
 * (sizeof: bgt_t 104, bgtm_t 184 with n_gt_read@16 h_out@80 a@88 n_fields@104 tbl_line@120 n_aal@144,
 * bgt_info_t 400, bcf1_t 152, bcf_hdr_t 104, fmf_t 48; checked by tests/test_host_shell.py).
 * Differences are confined to what the opaque pointers point at:
 *   bgt_file_t::idx   the in-memory site table of prefix.bcf (rid,pos,rlen,alleles,_row per site)
 *                     instead of a CSI index; prefix.bcf.csi must still exist, as in the reference
 *   bgt_t::pb         the device reader (bgth_reader_t over the HBM-resident prefix.pbf)
 *   bgt_t::bcf / itr  cursor / region state over the site table
 * Not provided in this build (genotype-independent or marked "next" in SURVEY.md 8f): BED filters,
 * -a/-S/-H allele queries, -t tables; their entry points exist and fail loudly.
 */
#ifndef BGT_READER_H
#define BGT_READER_H

#include <stdint.h>
#include "../bgt_amd/host/bcf2.h"
#include "../bgt_amd/host/metadata.h"

#define BGT_F_SET_AC    0x0001      /* -C: write AC/AN                       (ref bgt.h:8-11) */
#define BGT_F_NO_GT     0x0002      /* -G: no per-sample genotypes */
#define BGT_F_CNT_AL    0x0004
#define BGT_F_CNT_HAP   0x0008
#define BGT_MAX_GROUPS  32
#define BGT_MAX_ALLELES 64
#define BGT_SET_ALL_SAMPLES (-1)

typedef struct {
    char *prefix;
    fmf_t *f;                       /* prefix.spl */
    bcf_hdr_t *h0;                  /* header of the site-only prefix.bcf */
    void *idx;                      /* site table (see above) */
    int32_t *mgs;                   /* minimal group size per sample, -1 = unset */
    void *gpu;                      /* appended: shared HBM image of prefix.pbf (bgth_pbf_t*) */
} bgt_file_t;

typedef struct {
    const bgt_file_t *f;
    void *pb;                       /* device reader */
    void *bcf;                      /* cursor */
    bcf1_t *b0;                     /* the current site as a site-only record */
    void *itr;                      /* region, or NULL */
    const void *bed;
    int bed_excl, n_out, n_groups, mgs_def, *out;
    uint32_t *group, *gtag;
    bcf_hdr_t *h_out;
    const void *h_al;
} bgt_t;

typedef struct { const bcf1_t *b0; const uint8_t *a[2]; } bgt_rec_t;

typedef struct {
    int32_t ac[2], an, n_groups;
    int32_t gan[BGT_MAX_GROUPS], gac[BGT_MAX_GROUPS][2];
} bgt_info_t;

typedef struct { kstring_t chr; char *al; int rid, pos, rlen; } bgt_allele_t;
typedef struct { uint64_t hap; int tot, *cnt; } bgt_hapcnt_t;

typedef struct {
    int n_bgt, n_out, n_groups, flag;
    uint64_t n_gt_read;
    uint64_t *sample_idx;           /* db<<32 | sample */
    uint32_t *group;                /* 1-based group of every output sample */
    int32_t *mgs, mgs_def;
    bgt_t **bgt;
    bgt_rec_t *r;                   /* one look-ahead site per database */
    kexpr_t *site_flt;
    bcf_hdr_t *h_out;
    uint8_t *a[2];
    int n_fields;
    kexpr_t **fields;
    kstring_t tbl_line;
    int n_aal;
    bgt_allele_t *aal;
    void *h_al;
    int *alcnt;
    uint64_t *hap;
} bgtm_t;

extern int bgt_no_file;

#ifdef __cplusplus
extern "C" {
#endif
bgt_file_t *bgt_open(const char *prefix);
void bgt_close(bgt_file_t *bgt);

bgt_t *bgt_reader_init(const bgt_file_t *bf);
void bgt_reader_destroy(bgt_t *bgt);
/* BED interval sets (reference bedidx.c): keep / drop sites overlapping an interval */
void *bed_read(const char *fn);
int   bed_overlap(const void *bed, const char *chr, int beg, int end);
void  bed_destroy(void *bed);
void bgt_set_bed(bgt_t *bgt, const void *bed, int excl);
int bgt_set_region(bgt_t *bgt, const char *reg);
int bgt_set_start(bgt_t *bgt, int64_t n);
int bgt_read(bgt_t *bgt, bcf1_t *b);

bgtm_t *bgtm_reader_init(int n_files, bgt_file_t *const *fns);
void bgtm_reader_destroy(bgtm_t *bm);
void bgtm_set_flag(bgtm_t *bm, int flag);
int bgtm_set_flt_site(bgtm_t *bm, const char *expr);
void bgtm_set_bed(bgtm_t *bm, const void *bed, int excl);
int bgtm_set_region(bgtm_t *bm, const char *reg);
int bgtm_set_start(bgtm_t *bm, int64_t n);
int bgtm_set_table(bgtm_t *bm, const char *fmt);
int bgtm_set_alleles(bgtm_t *bm, const char *expr, const fmf_t *f, const char *fn);
int bgtm_set_mgs(bgtm_t *bm, int mgs_def);
int bgtm_add_group(bgtm_t *bm, const char *expr);
int bgtm_prepare(bgtm_t *bm);
int bgtm_test_mgs(const bgtm_t *bm);
int bgtm_read(bgtm_t *bm, bcf1_t *b);
/* extension (not in the reference): the next site as one VCF text line (no newline), identical to
 * vcf_format1(bm->h_out, b, s) after bgtm_read(bm, b); genotype columns come formatted from the device */
int bgtm_read_vcf(bgtm_t *bm, bcf1_t *b, kstring_t *s);
void bgtm_want_vcf_text(bgtm_t *bm);   /* before bgtm_prepare, if bgtm_read_vcf will be used */

bgt_hapcnt_t *bgtm_hapcnt(const bgtm_t *bm, int *n_hap);
char *bgtm_hapcnt_print_destroy(const bgtm_t *bm, int n_hap, bgt_hapcnt_t *hc);
char *bgtm_alcnt_print(const bgtm_t *bm);
int bgt_al_parse(const char *al, bgt_allele_t *a);
void bgt_al_format(const bgt_allele_t *a, kstring_t *s);
void bgt_al_from_bcf(const bcf_hdr_t *h, const bcf1_t *b, bgt_allele_t *a, bgt_allele_t *r);

/* command line front end (`bgt view`, reference view.c:14-183) */
int main_view(int argc, char *argv[]);
#ifdef __cplusplus
}
#endif
#endif
