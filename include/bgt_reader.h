/*
 * bgt_reader.h -- the reader API of BGT served by the MI355X path.  `#include "bgt.h"` of this repository
 * resolves here.
 *
 * Drop-in for the interface the reference declares in bgt.h:18-123: the same entry points, argument meaning and
 * return conventions, and -- because bgt-server.go reads struct fields directly (bgt-server.go:326-355) -- the
 * same member names, types and order, hence the same layouts on x86-64 (sizeof bgt_t 104, bgtm_t 184 with
 * n_gt_read@16 h_out@80 a@88 n_fields@104 tbl_line@120 n_aal@144, bgt_info_t 400, bcf1_t 152, bcf_hdr_t 104,
 * fmf_t 48; tests/test_host_shell.py compiles a probe against this header and checks every number).
 *
 * What differs is what the opaque pointers point at:
 *   bgt_file_t::idx    the site table of prefix.bcf held column-wise in memory, read on first use; a reader with a
 *                      region reads its sites through prefix.bcf.csi instead
 *   bgt_file_t::gpu    (appended member) the HBM image of prefix.pbf shared by the readers of the file
 *   bgt_t::pb          the reader's device side: a bgth_reader_t, possibly over a partial image of its own
 *   bgt_t::bcf, ::itr  cursor / region state over the site table
 *   bgtm_t::a          after bgtm_read(): the two byte planes of the merged site, as in the reference (bgt.c:829-842;
 *                      tests/test_api_harness.py reads them through both libraries).  Two departures: with
 *                      BGT_F_NO_GT no genotype is decoded at all (counts only), so bm->a is not filled; and the text
 *                      extension bgtm_read_vcf() leaves the device's GT vector / VCF text there instead
 * Everything `bgt view` of the reference does is served; import / atomize and the server are not part of this
 * library.
 */
#ifndef BGT_READER_H
#define BGT_READER_H

#include <stdint.h>
#include "../bgt_amd/host/bcf2.h"
#include "../bgt_amd/host/metadata.h"

/* bgtm_set_flag() bits (reference bgt.h:8-11) */
#define BGT_F_SET_AC    0x0001      /* -C: write AN / AC INFO                                    */
#define BGT_F_NO_GT     0x0002      /* -G: no per-sample genotype columns                        */
#define BGT_F_CNT_AL    0x0004      /* -S: count, per sample, the alleles of the set it carries  */
#define BGT_F_CNT_HAP   0x0008      /* -H: collect the haplotypes over the alleles of the set    */

#define BGT_MAX_GROUPS      32      /* -s may be given this many times                           */
#define BGT_MAX_ALLELES     64      /* a haplotype over the allele set is one 64-bit word        */
#define BGT_SET_ALL_SAMPLES (-1)

/* one database: prefix.bcf (sites) + prefix.pbf (genotypes) + prefix.spl (samples) */
typedef struct {
    char      *prefix;
    fmf_t     *f;                   /* sample metadata, prefix.spl                               */
    bcf_hdr_t *h0;                  /* header of the site-only prefix.bcf                        */
    void      *idx;                 /* site table (lazy)                                         */
    int32_t   *mgs;                 /* per sample: smallest group it may appear in, -1 = unset   */
    void      *gpu;                 /* appended: bgth_pbf_t*, the whole-file image in HBM        */
    int        gpu_opening;         /* appended: a reader is loading that image right now        */
    int        sites_pending;       /* appended: background loads of the site table in flight    */
    int        sites_loading;       /* appended: a thread is reading the site table right now    */
} bgt_file_t;

/* a reader over one database */
typedef struct {
    const bgt_file_t *f;
    void       *pb;                 /* device side of the reader                                 */
    void       *bcf;                /* cursor over the sites                                     */
    bcf1_t     *b0;                 /* the current site as a site-only record                    */
    void       *itr;                /* region being walked, or NULL                              */
    const void *bed;                /* BED interval set of -B, or NULL                           */
    int         bed_excl,           /* -e: drop the overlapping sites instead                    */
                n_out,              /* selected samples                                          */
                n_groups,
                mgs_def,
               *out;                /* [n_out] sample indices, ascending                         */
    uint32_t   *group,              /* [n_out] 1-based group of each                             */
               *gtag;               /* [samples] group tag while groups are being added          */
    bcf_hdr_t  *h_out;
    const void *h_al;               /* allele set of -a, or NULL                                 */
} bgt_t;

/* what a reader hands to the merge: the site and (if configured) its two byte planes */
typedef struct {
    const bcf1_t  *b0;
    const uint8_t *a[2];
} bgt_rec_t;

/* allele numbers of a site: total and per group (reference bgt.h:44-47) */
typedef struct {
    int32_t ac[2],                  /* ALT, <M>                                                  */
            an,
            n_groups;
    int32_t gan[BGT_MAX_GROUPS],
            gac[BGT_MAX_GROUPS][2];
} bgt_info_t;

/* an allele in normal form: chr (NUL) allele packed in `chr`, `al` points at the allele */
typedef struct {
    kstring_t chr;
    char     *al;
    int       rid, pos, rlen;
} bgt_allele_t;

/* one distinct haplotype over the allele set and how many carry it, in total and per group */
typedef struct {
    uint64_t hap;
    int      tot, *cnt;
} bgt_hapcnt_t;

/* a reader over several databases: samples side by side, sites merged */
typedef struct {
    int          n_bgt, n_out, n_groups, flag;
    uint64_t     n_gt_read;         /* genotypes decoded so far (statistics)                     */
    uint64_t    *sample_idx;        /* [n_out] database << 32 | sample                           */
    uint32_t    *group;             /* [n_out] 1-based group                                     */
    int32_t     *mgs, mgs_def;
    bgt_t      **bgt;               /* [n_bgt]                                                   */
    bgt_rec_t   *r;                 /* [n_bgt] one look-ahead site per database                  */
    kexpr_t     *site_flt;          /* -f                                                        */
    bcf_hdr_t   *h_out;
    uint8_t     *a[2];              /* merged planes, or GT vector / text (see above)            */
    int          n_fields;          /* -t                                                        */
    kexpr_t    **fields;
    kstring_t    tbl_line;
    int          n_aal;             /* alleles of the set met so far                             */
    bgt_allele_t *aal;
    void        *h_al;
    int         *alcnt;             /* [n_out] -S                                                */
    uint64_t    *hap;               /* [2 n_out] -H                                              */
} bgtm_t;

/* extension (resident processes): build the whole-file device image and the site table now; 0 or -1 */
int     bgt_file_preload(const bgt_file_t *bf);
extern int bgt_no_file;             /* 1: never interpret an argument as a file name (server)    */

#ifdef __cplusplus
extern "C" {
#endif

/* ---- databases ---- */
bgt_file_t *bgt_open(const char *prefix);                 /* NULL if any of .bcf .bcf.csi .spl is unusable */
void        bgt_close(bgt_file_t *bgt);

/* ---- one database ---- */
bgt_t *bgt_reader_init(const bgt_file_t *bf);
void   bgt_reader_destroy(bgt_t *bgt);
int    bgt_set_region(bgt_t *bgt, const char *reg);       /* "chr", "chr:beg-end" (1-based, commas allowed); <0 on error */
int    bgt_set_start(bgt_t *bgt, int64_t n);              /* start at the n-th site (0-based)              */
void   bgt_set_bed(bgt_t *bgt, const void *bed, int excl);
int    bgt_read(bgt_t *bgt, bcf1_t *b);                   /* row number >= 0, <0 at the end                */

/* ---- several databases ---- */
bgtm_t *bgtm_reader_init(int n_files, bgt_file_t *const *fns);
void    bgtm_reader_destroy(bgtm_t *bm);
void    bgtm_set_flag(bgtm_t *bm, int flag);
int     bgtm_set_mgs(bgtm_t *bm, int mgs_def);
int     bgtm_add_group(bgtm_t *bm, const char *expr);     /* -s: ",a,b" | ":a,b" | file | expression; size or <0 */
int     bgtm_set_flt_site(bgtm_t *bm, const char *expr);  /* -f; non-zero = parse error bits               */
int     bgtm_set_region(bgtm_t *bm, const char *reg);
int     bgtm_set_start(bgtm_t *bm, int64_t n);
void    bgtm_set_bed(bgtm_t *bm, const void *bed, int excl);
int     bgtm_set_table(bgtm_t *bm, const char *fmt);      /* -t                                            */
int     bgtm_set_alleles(bgtm_t *bm, const char *expr, const fmf_t *f, const char *fn);   /* -a [-d [-M]]: alleles, or <0 */
int     bgtm_prepare(bgtm_t *bm);                         /* merged samples, groups, output header         */
int     bgtm_test_mgs(const bgtm_t *bm);
int     bgtm_read(bgtm_t *bm, bcf1_t *b);                 /* >= 0 per emitted site, <0 at the end          */

/* extensions (not in the reference): the next site as one VCF text line without the newline -- what
 * vcf_format1(bm->h_out, b, s) gives after bgtm_read(bm, b) -- with the genotype columns formatted on the device;
 * announce it before bgtm_prepare so that the device is asked for the text */
void    bgtm_want_vcf_text(bgtm_t *bm);
int     bgtm_read_vcf(bgtm_t *bm, bcf1_t *b, kstring_t *s);
/* every remaining site as VCF text in one call -- one device scan for all counts, the lines formatted on several host
 * threads (BGT_THREADS, default min(cores, 16)), written in order; the bytes of the bgtm_read_vcf loop.  Only for one
 * database without genotype columns, region, BED, allele set or table; returns the records written, -1 if the query
 * needs the site-by-site path (nothing was written), -2 on a device error.  Blocks are streamed out as they become ready, so
 * a -2 may FOLLOW part of the body (whole lines, in order, nothing after the failure): a caller must not take the output
 * for complete -- `bgt view` prints "[E::main_view] reading stopped on an error ...: the output is incomplete." and exits 1. */
long    bgtm_write_vcf_bulk(bgtm_t *bm, FILE *fp, long n_rec);   /* one database or a merge of up to 64 */

/* ---- allele sets: samples carrying all of them (-S), haplotype counts (-H) ---- */
char         *bgtm_alcnt_print(const bgtm_t *bm);                                  /* malloc'd text, caller frees */
bgt_hapcnt_t *bgtm_hapcnt(const bgtm_t *bm, int *n_hap);
char         *bgtm_hapcnt_print_destroy(const bgtm_t *bm, int n_hap, bgt_hapcnt_t *hc);
int           bgt_al_parse(const char *al, bgt_allele_t *a);                       /* "chr:pos:rlen|REF:ALT"      */
void          bgt_al_format(const bgt_allele_t *a, kstring_t *s);
void          bgt_al_from_bcf(const bcf_hdr_t *h, const bcf1_t *b, bgt_allele_t *a, bgt_allele_t *r);

/* ---- BED interval sets (reference bedidx.c) ---- */
void *bed_read(const char *fn);
int   bed_overlap(const void *bed, const char *chr, int beg, int end);
void  bed_destroy(void *bed);

/* ---- command line front ends: `bgt view` (reference view.c:14-183), `bgt import` (import.c:8-120) ---- */
int main_view(int argc, char *argv[]);
/* `bgt view` for a RESIDENT host (extension; `bgt-server -u SOCKET` is the one in this repo): the same query on the
 * caller's thread, written to out / err, with the databases taken from and given back to the host (open / close are
 * called with ctx) instead of being opened and closed per query; nothing process-wide is touched.  host == NULL is what
 * main_view runs in a command-line process.  With BGT_SERVER=<socket> in its environment main_view hands the query --
 * arguments, working directory and its own stdout / stderr descriptors -- to such a host and leaves with its status. */
typedef struct {
    bgt_file_t *(*open)(const char *prefix, void *ctx);
    void        (*close)(bgt_file_t *bgt, void *ctx);
    void         *ctx;
} bgt_view_host_t;
int view_run(int argc, char *argv[], FILE *out, FILE *err, const bgt_view_host_t *host);
int main_pbfview(int argc, char **argv);   /* `bgt pbfview` (reference pbfview.c) */
int main_import(int argc, char *argv[]);

#ifdef __cplusplus
}
#endif
#endif
