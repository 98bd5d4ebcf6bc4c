/* reader.c -- the BGT reader API (include/bgt_reader.h) on top of the MI355X codec (include/bgt_hip.h).
 *
 * Host logic only: which site comes next, which samples are selected, how the per-database results are
 * merged, which INFO fields are written, whether the site passes the filter.  Every genotype and every
 * allele count comes from the device through bgth_reader_read(); there is no CPU decoder here.
 * Reference behaviour restated: bgt.c:40-81 (open), :89-246 (single reader), :272-356 (site pull),
 * :364-676 (multi reader set-up), :692-757 (INFO and filter), :797-888 (merge loop).
 */
#define _GNU_SOURCE                                            /* F_SETPIPE_SZ */
#include <assert.h>
#include <ctype.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <pthread.h>
#include <zlib.h>
#include "../../include/bgt_reader.h"
#include "../../include/bgt_hip.h"

int bgt_no_file = 0;

/* ------------------------------------------------------------------------------------------------
 * site table: prefix.bcf held column-wise in memory
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int64_t n;
    int32_t *rid, *pos, *rlen, *row, *n_allele;
    uint64_t *raw_off;                                           /* the record's `shared` block as in the file (bgt_read); 64-bit: */
    uint64_t *ref_off, *alt_off;                                 /* WGS-scale tables pass 4 GiB.  REF / ALT point INTO that block */
    uint32_t *ref_len, *alt_len, *raw_len;
    uint16_t *n_info; float *qual;
    char *pool; size_t pool_len, pool_cap;
    int32_t max_rlen;
} sitetab_t;

static void st_free(sitetab_t *t)
{
    if (!t) return;
    free(t->rid); free(t->pos); free(t->rlen); free(t->row); free(t->n_allele);
    free(t->ref_off); free(t->alt_off); free(t->ref_len); free(t->alt_len); free(t->raw_off); free(t->raw_len);
    free(t->n_info); free(t->qual); free(t->pool); free(t);
}

static uint64_t st_intern(sitetab_t *t, const uint8_t *s, int n)
{
    const uint64_t at = t->pool_len;
    if (t->pool_len + (size_t)n + 1 > t->pool_cap) {
        t->pool_cap = t->pool_cap ? t->pool_cap * 2 : 1 << 16;
        while (t->pool_len + (size_t)n + 1 > t->pool_cap) t->pool_cap *= 2;
        t->pool = (char*)realloc(t->pool, t->pool_cap);
    }
    memcpy(t->pool + at, s, (size_t)n);
    t->pool[at + (uint64_t)n] = 0;
    t->pool_len += (size_t)n + 1;
    return at;
}

static int tv_bytes(int type) { return type == 1 || type == 7 ? 1 : type == 2 ? 2 : (type == 3 || type == 5) ? 4 : 0; }
static int32_t tv_int(const uint8_t *p, int type)
{
    if (type == 1) return *(const int8_t*)p;
    if (type == 2) { int16_t v; memcpy(&v, p, 2); return v; }
    { int32_t v; memcpy(&v, p, 4); return v; }
}
static int tv_size(const uint8_t *p, const uint8_t **q, int *type)
{
    *type = *p & 15;
    if (*p >> 4 != 15) { *q = p + 1; return *p >> 4; }
    { int t = p[1] & 15; *q = p + 2 + tv_bytes(t); return tv_int(p + 2, t); }
}

/* one record of the site-only BCF appended to the table: rid/pos/rlen, REF, first ALT, number of alleles and
 * INFO/_row (the row of the genotype matrix; ref bgt.c:272-288 asserts it is present and that there are no
 * samples).  Returns 0, or <0 for a record this format does not allow. */
static int st_append(sitetab_t *t, int64_t *cap, const bcf1_t *b, int row_key)
{
    const uint8_t *p = (const uint8_t*)b->shared.s, *q, *ref, *alt;
    int n, type, i, row = -1;
    uint64_t at;
    if (t->n == *cap) {
        const int64_t c = *cap = *cap ? *cap * 2 : 1 << 12;
        t->rid = (int32_t*)realloc(t->rid, (size_t)c * 4); t->pos = (int32_t*)realloc(t->pos, (size_t)c * 4);
        t->rlen = (int32_t*)realloc(t->rlen, (size_t)c * 4); t->row = (int32_t*)realloc(t->row, (size_t)c * 4);
        t->n_allele = (int32_t*)realloc(t->n_allele, (size_t)c * 4);
        t->ref_off = (uint64_t*)realloc(t->ref_off, (size_t)c * 8); t->alt_off = (uint64_t*)realloc(t->alt_off, (size_t)c * 8);
        t->ref_len = (uint32_t*)realloc(t->ref_len, (size_t)c * 4); t->alt_len = (uint32_t*)realloc(t->alt_len, (size_t)c * 4);
        t->raw_off = (uint64_t*)realloc(t->raw_off, (size_t)c * 8); t->raw_len = (uint32_t*)realloc(t->raw_len, (size_t)c * 4);
        t->n_info = (uint16_t*)realloc(t->n_info, (size_t)c * 2); t->qual = (float*)realloc(t->qual, (size_t)c * 4);
    }
    if (b->n_sample != 0 || b->n_allele < 2) return -3;
    n = tv_size(p, &q, &type); p = q + n;                                   /* ID */
    n = tv_size(p, &q, &type);                                              /* REF */
    ref = q; t->ref_len[t->n] = (uint32_t)n; p = q + n;
    n = tv_size(p, &q, &type);                                              /* first ALT */
    alt = q; t->alt_len[t->n] = (uint32_t)n; p = q + n;
    for (i = 2; i < (int)b->n_allele; ++i) { n = tv_size(p, &q, &type); p = q + n; }
    n = tv_size(p, &q, &type); p = q + (size_t)n * tv_bytes(type);          /* FILTER */
    for (i = 0; i < (int)b->n_info; ++i) {
        int kt, key;
        tv_size(p, &q, &kt); key = tv_int(q, kt); p = q + tv_bytes(kt);
        n = tv_size(p, &q, &type);
        if (key == row_key && n >= 1) row = tv_int(q, type);
        p = q + (size_t)n * tv_bytes(type);
    }
    if (row < 0) return -3;
    t->rid[t->n] = b->rid; t->pos[t->n] = b->pos; t->rlen[t->n] = b->rlen;
    t->row[t->n] = row; t->n_allele[t->n] = b->n_allele;
    /* one copy of the record: every user of REF / ALT passes their lengths, none needs a terminator */
    at = st_intern(t, (const uint8_t*)b->shared.s, (int)b->shared.l);
    t->raw_off[t->n] = at; t->raw_len[t->n] = (uint32_t)b->shared.l;
    t->ref_off[t->n] = at + (uint64_t)(ref - (const uint8_t*)b->shared.s);
    t->alt_off[t->n] = at + (uint64_t)(alt - (const uint8_t*)b->shared.s);
    t->n_info[t->n] = (uint16_t)b->n_info; t->qual[t->n] = b->qual;
    if (b->rlen > t->max_rlen) t->max_rlen = b->rlen;
    ++t->n;
    return 0;
}

/* every record of the file */
static sitetab_t *st_load(bgzr_t *fp, const bcf_hdr_t *h)
{
    const int row_key = bcf_id2int(h, BCF_DT_ID, "_row");
    sitetab_t *t = (sitetab_t*)calloc(1, sizeof(*t));
    bcf1_t *b = bcf_init1();
    int64_t cap = 0;
    int ret;
    if (row_key < 0) { bcf_destroy1(b); st_free(t); return NULL; }
    while ((ret = bcf_read1_stream(fp, b)) == 0)
        if ((ret = st_append(t, &cap, b, row_key)) < 0) break;
    bcf_destroy1(b);
    if (ret < -1) { st_free(t); return NULL; }
    return t;
}

/* The records that overlap [beg,end) of contig tid, found through prefix.bcf.csi (coordinate-sorted index: bins
 * of 2^(min_shift+3k) bases, per bin the chunks of the file that hold its records; reference hts.c:725-907) so
 * that a region query of a large database reads a few BGZF blocks instead of every site.  NULL if the index
 * cannot be used (then the caller loads the whole table). */
typedef struct { uint64_t beg, end; } chunk_t;
static int cmp_chunk(const void *a, const void *b)
{
    const uint64_t x = ((const chunk_t*)a)->beg, y = ((const chunk_t*)b)->beg;
    return x < y ? -1 : x > y;
}

static sitetab_t *st_load_region(const char *prefix, const bcf_hdr_t *h, int tid, int beg, int end)
{
    const int row_key = bcf_id2int(h, BCF_DT_ID, "_row");
    char *fn = (char*)malloc(strlen(prefix) + 16);
    bgzr_t *ix, *fp = NULL;
    sitetab_t *t = NULL;
    chunk_t *ch = NULL;
    int32_t hdr4[4], n_ref, r, n_ch = 0, m_ch = 0, i;
    uint8_t magic[4];
    int ok = 0;
    sprintf(fn, "%s.bcf.csi", prefix);
    ix = bgzr_open(fn);
    if (ix == NULL || row_key < 0) goto done;
    if (bgzr_read(ix, magic, 4) != 4 || memcmp(magic, "CSI\1", 4) != 0 || bgzr_read(ix, hdr4, 12) != 12) goto done;
    {   /* min_shift, depth, l_aux */
        const int min_shift = hdr4[0], depth = hdr4[1], l_aux = hdr4[2];
        if (min_shift < 0 || depth < 0 || depth > 10 || l_aux < 0) goto done;
        for (i = 0; i < l_aux; ++i) { uint8_t c; if (bgzr_read(ix, &c, 1) != 1) goto done; }
        if (bgzr_read(ix, &n_ref, 4) != 4 || tid < 0 || tid >= n_ref) goto done;
        if (end > beg) {
            for (r = 0; r <= tid; ++r) {
                int32_t n_bin, k;
                if (bgzr_read(ix, &n_bin, 4) != 4 || n_bin < 0) goto done;
                for (k = 0; k < n_bin; ++k) {
                    uint32_t bin; uint64_t loff; int32_t n_chunk, c, want = 0;
                    if (bgzr_read(ix, &bin, 4) != 4 || bgzr_read(ix, &loff, 8) != 8 || bgzr_read(ix, &n_chunk, 4) != 4 || n_chunk < 0) goto done;
                    if (r == tid) {                           /* is `bin` one of the bins that overlap [beg,end)? */
                        int l, s = min_shift + depth * 3;
                        uint32_t first = 0;
                        for (l = 0; l <= depth; s -= 3, first += 1u << (l * 3), ++l)
                            if (bin >= first + (uint32_t)(beg >> s) && bin <= first + (uint32_t)((end - 1) >> s) &&
                                bin < first + (1u << (l * 3))) want = 1;
                    }
                    for (c = 0; c < n_chunk; ++c) {
                        chunk_t x;
                        if (bgzr_read(ix, &x, 16) != 16) goto done;
                        if (want) {
                            if (n_ch == m_ch) { m_ch = m_ch ? m_ch << 1 : 16; ch = (chunk_t*)realloc(ch, (size_t)m_ch * sizeof(chunk_t)); }
                            ch[n_ch++] = x;
                        }
                    }
                }
            }
        }
    }
    sprintf(fn, "%s.bcf", prefix);
    if ((fp = bgzr_open(fn)) == NULL) goto done;
    t = (sitetab_t*)calloc(1, sizeof(*t));
    if (n_ch > 0) {
        bcf1_t *b = bcf_init1();
        int64_t cap = 0;
        int bad = 0;
        qsort(ch, (size_t)n_ch, sizeof(chunk_t), cmp_chunk);
        for (i = 0; i < n_ch && !bad;) {                      /* overlapping / touching chunks are read once */
            uint64_t cb = ch[i].beg, ce = ch[i].end;
            for (++i; i < n_ch && ch[i].beg <= ce; ++i) if (ch[i].end > ce) ce = ch[i].end;
            if (bgzr_seek(fp, cb) < 0) { bad = 1; break; }
            while (bgzr_tell(fp) < ce) {
                const int ret = bcf_read1_stream(fp, b);
                if (ret != 0) { bad = ret < -1; break; }
                if (b->rid != tid || b->pos >= end) { if (b->rid > tid || b->pos >= end) break; else continue; }
                if (b->pos + b->rlen <= beg) continue;
                if (st_append(t, &cap, b, row_key) < 0) { bad = 1; break; }
            }
        }
        bcf_destroy1(b);
        if (bad) { st_free(t); t = NULL; goto done; }
    }
    ok = 1;
done:
    (void)ok;
    free(ch); free(fn);
    if (ix) bgzr_close(ix);
    if (fp) bgzr_close(fp);
    return t;
}

/* order of sites across databases: contig, position, reference length, then the first ALT as bytes with
 * the shorter one first on a tie (ref vcf.c:1152-1164) */
static int st_cmp(const sitetab_t *a, int64_t i, const sitetab_t *b, int64_t j)
{
    int la, lb, r;
    if (a->rid[i] != b->rid[j]) return a->rid[i] - b->rid[j];
    if (a->pos[i] != b->pos[j]) return a->pos[i] - b->pos[j];
    if (a->rlen[i] != b->rlen[j]) return a->rlen[i] - b->rlen[j];
    la = a->alt_len[i]; lb = b->alt_len[j];
    r = strncmp(a->pool + a->alt_off[i], b->pool + b->alt_off[j], (size_t)(la < lb ? la : lb));
    return r ? r : la - lb;
}

/* ------------------------------------------------------------------------------------------------
 * private state behind the opaque pointers of bgt_t
 * ------------------------------------------------------------------------------------------------ */
typedef struct { int n, m; char **key; } alset_t;             /* bgtm_t::h_al / bgt_t::h_al: formatted alleles of -a */
static int al_present(const alset_t *h, const char *chr, int rid, int pos, int rlen, const char *ref, int l_ref,
                      const char *alt, int l_alt);
static const sitetab_t *file_sites(const bgt_file_t *bf);
static const sitetab_t *sites_of(const bgt_t *bgt);
typedef struct { int64_t next; } cursor_t;                   /* bgt_t::bcf */
typedef struct { int tid, beg, end; int64_t at; int done; } region_t;   /* bgt_t::itr */
typedef struct {                                              /* bgt_t::pb */
    bgth_reader_t *rd; int64_t site; const int32_t *counts; int skip_device;
    int want;                  /* BGTH_WANT_* bits the device delivers per site for this reader */
    int text_mode;             /* the caller formats VCF text (bgtm_read_vcf): also ask for the genotype text */
    const int8_t *gt8; const char *gttext;   /* genotype vector / text of the current site (device formatted) */
    bgth_pbf_t *own_img;       /* a partial image of the .pbf that only this reader uses (region / start queries) */
    void *own_sites;           /* sitetab_t: the sites of this reader's region, loaded through the CSI index */
    int n_groups_total;        /* as passed to the last selection, to re-apply it on another image */
    /* -S / -H: the device folds matched rows into per-reader accumulators (bgth_reader_fold_last); drained here */
    int folded;                /* the device holds folds not yet drained */
    int32_t *f_car; uint64_t *f_hap;   /* running totals: carriers[n_out], signatures[2 n_out] */
} devrd_t;

/* The whole-file image of a database.  BGT_GPUS spreads it over several devices (site-range shards behind the codec
 * seam, SURVEY.md 8e: `bgt view` then uses every listed GPU, two-database merges included, with no change above this
 * line): "N" = devices 0..N-1, "all" = every visible device, or an explicit list "0,1,2,3" (a device may repeat:
 * several shards on one GPU).  Unset: device 0. */
static bgth_pbf_t *open_whole_image(const char *fn)
{
    const char *e = getenv("BGT_GPUS");
    int dev[64], n = 0;
    if (e && *e) {
        if (strcmp(e, "all") == 0) { n = bgth_device_count(); if (n > 64) n = 64; for (int i = 0; i < n; ++i) dev[i] = i; }
        else if (strchr(e, ',')) {
            const char *q = e;
            while (*q && n < 64) { dev[n++] = atoi(q); q = strchr(q, ','); if (!q) break; ++q; }
        } else { n = atoi(e); if (n > 64) n = 64; for (int i = 0; i < n; ++i) dev[i] = i; }
    }
    if (n > 1) return bgth_pbf_open_sharded(fn, n, dev);
    return bgth_pbf_open(fn, n == 1 ? dev[0] : 0);
}

/* ------------------------------------------------------------------------------------------------
 * files
 * ------------------------------------------------------------------------------------------------ */
static void set_mgs(bgt_file_t *bf)                           /* ref bgt.c:24-38: optional _mgs:i: tag */
{
    int i, j, key = -1;
    for (i = 0; i < bf->f->n_rows; ++i) bf->mgs[i] = -1;
    for (i = 0; i < bf->f->n_keys; ++i) if (strcmp(bf->f->keys[i], "_mgs") == 0) key = i;
    if (key < 0) return;
    for (i = 0; i < bf->f->n_rows; ++i) {
        const fmf1_t *r = &bf->f->rows[i];
        for (j = 0; j < r->n_meta; ++j)
            if ((int)r->meta[j].key == key && r->meta[j].type == FMF_INT && r->meta[j].v.i >= 0) bf->mgs[i] = r->meta[j].v.i;
    }
}

bgt_file_t *bgt_open(const char *prefix)
{
    char *fn = (char*)malloc(strlen(prefix) + 16);
    bgt_file_t *bf = NULL;
    bgzr_t *fp;
    FILE *t;
    sprintf(fn, "%s.bcf", prefix);
    if ((fp = bgzr_open(fn)) == NULL) goto fail;
    bf = (bgt_file_t*)calloc(1, sizeof(*bf));
    if ((bf->h0 = bcf_hdr_read_stream(fp)) == NULL) goto fail;
    sprintf(fn, "%s.bcf.csi", prefix);                       /* the reference refuses a BGT without its index */
    if ((t = fopen(fn, "rb")) == NULL) goto fail;
    fclose(t);
    sprintf(fn, "%s.spl", prefix);                            /* (the site table is read on first use: file_sites) */
    if ((bf->f = fmf_read(fn)) == NULL) goto fail;
    bf->prefix = strdup(prefix);
    bf->mgs = (int32_t*)calloc((size_t)(bf->f->n_rows ? bf->f->n_rows : 1), 4);
    set_mgs(bf);
    bgzr_close(fp);
    free(fn);
    return bf;
fail:
    free(fn);
    if (fp) bgzr_close(fp);
    if (bf) bgt_close(bf);
    return NULL;
}

static void wait_for_site_loads(bgt_file_t *bf);
void bgt_close(bgt_file_t *bf)
{
    if (!bf) return;
    wait_for_site_loads(bf);
    if (bf->gpu) bgth_pbf_close((bgth_pbf_t*)bf->gpu);
    free(bf->mgs);
    st_free((sitetab_t*)bf->idx);
    if (bf->h0) bcf_hdr_destroy(bf->h0);
    if (bf->f) fmf_destroy(bf->f);
    free(bf->prefix); free(bf);
}

/* ------------------------------------------------------------------------------------------------
 * single-database reader
 * ------------------------------------------------------------------------------------------------ */
bgt_t *bgt_reader_init(const bgt_file_t *bf)
{
    bgt_t *bgt = (bgt_t*)calloc(1, sizeof(*bgt));
    devrd_t *dv = (devrd_t*)calloc(1, sizeof(*dv));
    bgt->f = bf;
    bgt->pb = dv;
    bgt->bcf = calloc(1, sizeof(cursor_t));
    bgt->b0 = bcf_init1();
    bgt->gtag = (uint32_t*)calloc((size_t)(bf->f->n_rows ? bf->f->n_rows : 1), 4);
    return bgt;
}

void bgt_reader_destroy(bgt_t *bgt)
{
    devrd_t *dv;
    if (!bgt) return;
    dv = (devrd_t*)bgt->pb;
    if (dv) { if (dv->rd) bgth_reader_destroy(dv->rd); if (dv->own_img) bgth_pbf_close(dv->own_img); st_free((sitetab_t*)dv->own_sites); free(dv->f_car); free(dv->f_hap); free(dv); }
    bcf_destroy1(bgt->b0);
    free(bgt->gtag); free(bgt->group); free(bgt->out); free(bgt->bcf); free(bgt->itr);
    if (bgt->h_out) bcf_hdr_destroy(bgt->h_out);
    free(bgt);
}

/* names after ':' or ',' separated by commas, or the first column of a file (ref hts.c hts_readlines) */
static char **read_names(const char *arg, int *n_out)
{
    char **s = NULL;
    int n = 0, m = 0;
    gzFile fp = gzopen(arg, "r");
    *n_out = 0;
    if (fp) {
        char line[65536];
        while (gzgets(fp, line, sizeof(line))) {
            size_t l = strcspn(line, "\t\n");
            if (line[0] == '\n' || line[0] == 0) continue;
            line[l] = 0;
            if (n == m) { m = m ? m * 2 : 16; s = (char**)realloc(s, (size_t)m * sizeof(char*)); }
            s[n++] = strdup(line);
        }
        gzclose(fp);
    } else if (*arg == ':' || *arg == ',') {
        const char *p, *q;
        for (q = p = arg + 1;; ++p)
            if (*p == ',' || *p == 0) {
                if (n == m) { m = m ? m * 2 : 16; s = (char**)realloc(s, (size_t)m * sizeof(char*)); }
                s[n] = (char*)calloc((size_t)(p - q) + 1, 1);
                memcpy(s[n++], q, (size_t)(p - q));
                q = p + 1;
                if (*p == 0) break;
            }
    } else return NULL;
    *n_out = n;
    return s;
}

static int cmp_str(const void *a, const void *b) { return strcmp(*(char* const*)a, *(char* const*)b); }

/* tag the samples of one more group; returns its size or <0 (ref bgt.c:121-161) */
static int add_group_core(bgt_t *bgt, int n, char **samples, const char *expr)
{
    const fmf_t *f = bgt->f->f;
    int i, size = 0;
    if (n == BGT_SET_ALL_SAMPLES) {
        for (i = 0; i < f->n_rows; ++i) bgt->gtag[i] = 1;
        bgt->n_groups = 1;
        return f->n_rows;
    }
    if (n > 0 || expr != NULL) {
        kexpr_t *ke = NULL;
        int err;
        if (expr && (ke = ke_parse(expr, &err)) == NULL) return -1;
        if (n > 0) qsort(samples, (size_t)n, sizeof(char*), cmp_str);
        for (i = 0; i < f->n_rows; ++i) {
            int add = 0;
            if (ke && fmf_test(f, i, ke)) add = 1;
            if (n > 0 && bsearch(&f->rows[i].name, samples, (size_t)n, sizeof(char*), cmp_str)) {
                const int mgs = bgt->f->mgs[i] >= 0 ? bgt->f->mgs[i] : bgt->mgs_def;
                if (mgs == 1 || mgs == 0) add = 1;           /* a sample may be named only if its mgs allows it */
            }
            if (add) { ++size; bgt->gtag[i] = (uint32_t)bgt->n_groups + 1; }   /* a later group overrides */
        }
        ke_destroy(ke);
        ++bgt->n_groups;
        return size;
    }
    return -1;
}

static int is_file(const char *fn)
{
    FILE *fp;
    if (bgt_no_file) return 0;
    if ((fp = fopen(fn, "r")) == NULL) return 0;
    fclose(fp);
    return 1;
}

static int add_group(bgt_t *bgt, const char *expr)            /* ref bgt.c:174-188 */
{
    if (*expr == ':' || *expr == ',' || (*expr != '?' && is_file(expr))) {
        int n, i, ret;
        char **s = read_names(expr, &n);
        ret = add_group_core(bgt, n, s, NULL);
        for (i = 0; i < n; ++i) free(s[i]);
        free(s);
        return ret;
    }
    return add_group_core(bgt, 0, NULL, expr);
}

/* "chr", "chr:beg-end", "chr:beg" with 1-based inclusive coordinates and optional thousands commas
 * (ref hts.c:821-850); returns the length of the contig name */
static int parse_region(const char *s, int *beg, int *end)
{
    int l = (int)strlen(s), name_end = l, i, k;
    *beg = 0; *end = 1 << 29;
    for (i = l - 1; i >= 0; --i) if (s[i] == ':') break;
    if (i >= 0) name_end = i;
    if (name_end < l) {
        int hyphens = 0;
        for (i = name_end + 1; i < l; ++i) {
            if (s[i] == '-') ++hyphens;
            else if (!(s[i] >= '0' && s[i] <= '9') && s[i] != ',') break;
        }
        if (i < l || hyphens > 1) name_end = l;
    }
    if (name_end < l) {
        char *tmp = (char*)malloc((size_t)(l - name_end) + 1), *p;
        for (i = name_end + 1, k = 0; i < l; ++i) if (s[i] != ',') tmp[k++] = s[i];
        tmp[k] = 0;
        if ((*beg = (int)strtol(tmp, &p, 10) - 1) < 0) *beg = 0;
        *end = *p ? (int)strtol(p + 1, &p, 10) : 1 << 29;
        free(tmp);
        if (*beg > *end) { name_end = l; *beg = 0; *end = 1 << 29; }
    }
    return name_end;
}

int bgt_set_region(bgt_t *bgt, const char *reg)               /* ref bgt.c:190-196, hts.c:852-866 */
{
    const sitetab_t *t;
    devrd_t *dv = (devrd_t*)bgt->pb;
    region_t *r;
    int beg, end, tid, nl;
    char *name;
    int64_t lo, hi;
    free(bgt->itr); bgt->itr = NULL;
    nl = parse_region(reg, &beg, &end);
    name = (char*)calloc((size_t)nl + 1, 1);
    memcpy(name, reg, (size_t)nl);
    if ((tid = bcf_id2int(bgt->f->h0, BCF_DT_CTG, name)) < 0) tid = bcf_id2int(bgt->f->h0, BCF_DT_CTG, reg);
    free(name);
    if (tid < 0) return -1;
    /* while nobody has needed the whole site table yet, read only this region's sites through the CSI index */
    st_free((sitetab_t*)dv->own_sites); dv->own_sites = NULL;
    if (bgt->f->idx == NULL) dv->own_sites = st_load_region(bgt->f->prefix, bgt->f->h0, tid, beg, end);
    t = sites_of(bgt);
    r = (region_t*)calloc(1, sizeof(*r));
    r->tid = tid; r->beg = beg; r->end = end;
    /* first site that can overlap: sites are sorted by (contig, position); a site starting up to
     * max_rlen before `beg` may still reach into the region */
    lo = 0; hi = t->n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (t->rid[mid] < tid || (t->rid[mid] == tid && t->pos[mid] + t->max_rlen <= beg)) lo = mid + 1; else hi = mid;
    }
    r->at = lo;
    bgt->itr = r;
    bgt->b0->shared.l = 0;
    return 0;
}

int bgt_set_start(bgt_t *bgt, int64_t i)                      /* ref bgt.c:198-201, vcf.c:1195-1209 */
{
    const sitetab_t *t = sites_of(bgt);
    if (i < 0 || i >= t->n) return -1;                        /* the reference does not move the file then */
    ((cursor_t*)bgt->bcf)->next = i;
    return 0;
}

void bgt_set_bed(bgt_t *bgt, const void *bed, int excl) { bgt->bed = bed; bgt->bed_excl = excl; }

/* selected samples in .spl order and their groups; install the column selection on the device
 * (ref bgt.c:207-246; the subset list is {2s, 2s+1} for every selected sample s) */
/* the HBM image of prefix.pbf is opened on first need and cached on the file handle, shared by every
 * reader of that file; each reader owns its own device reader (stream, selection, result buffers) */
static pthread_mutex_t g_open_lock = PTHREAD_MUTEX_INITIALIZER;   /* readers of one file may start on different threads */
static pthread_cond_t g_open_cond = PTHREAD_COND_INITIALIZER;     /* ... and wait here while another one loads the file's image */
static pthread_mutex_t g_sites_lock = PTHREAD_MUTEX_INITIALIZER;  /* the site table has a lock of its own: it loads beside the image */

/* the site table of the whole file, read on first use */
static const sitetab_t *file_sites(const bgt_file_t *cbf);
static pthread_cond_t g_sites_cond = PTHREAD_COND_INITIALIZER;     /* bgt_close waits here for background loads of its file */
static void *sites_loader(void *arg)
{
    bgt_file_t *bf = (bgt_file_t*)arg;
    file_sites(bf);
    pthread_mutex_lock(&g_sites_lock);
    --bf->sites_pending;
    pthread_cond_broadcast(&g_sites_cond);
    pthread_mutex_unlock(&g_sites_lock);
    return NULL;
}
/* start reading the site table on a thread of its own while the caller opens the .pbf image (the two are the long
 * steps before the first site of a whole-file walk); whoever needs the table first waits on the same lock.  The file
 * counts the loads in flight: bgt_close must not free it under a loader that has not finished (or not even started). */
static int sites_prefetch(const bgt_file_t *cbf, pthread_t *th)
{
    bgt_file_t *bf = (bgt_file_t*)cbf;
    int ok;
    pthread_mutex_lock(&g_sites_lock);
    if (bf->idx) { pthread_mutex_unlock(&g_sites_lock); return 0; }
    ++bf->sites_pending;
    pthread_mutex_unlock(&g_sites_lock);
    ok = pthread_create(th, NULL, sites_loader, (void*)bf) == 0;
    if (!ok) {
        pthread_mutex_lock(&g_sites_lock);
        --bf->sites_pending;
        pthread_cond_broadcast(&g_sites_cond);
        pthread_mutex_unlock(&g_sites_lock);
    }
    return ok;
}

static void wait_for_site_loads(bgt_file_t *bf)
{
    pthread_mutex_lock(&g_sites_lock);
    while (bf->sites_pending > 0) pthread_cond_wait(&g_sites_cond, &g_sites_lock);
    pthread_mutex_unlock(&g_sites_lock);
}

static const sitetab_t *file_sites(const bgt_file_t *cbf)
{
    bgt_file_t *bf = (bgt_file_t*)cbf;
    static const sitetab_t empty;
    /* the global lock only guards the test and the publication: the tables of two databases of a merge load side by side
     * (a per-file loading flag keeps a second reader of the SAME file waiting instead of loading it twice) */
    pthread_mutex_lock(&g_sites_lock);
    while (bf->sites_loading) pthread_cond_wait(&g_sites_cond, &g_sites_lock);
    if (bf->idx == NULL) {
        char *fn = (char*)malloc(strlen(bf->prefix) + 8);
        bgzr_t *fp;
        sitetab_t *t = NULL;
        bf->sites_loading = 1;
        pthread_mutex_unlock(&g_sites_lock);
        sprintf(fn, "%s.bcf", bf->prefix);
        if ((fp = bgzr_open(fn)) != NULL) {
            bcf_hdr_t *h = bcf_hdr_read_stream(fp);              /* skip the header */
            if (h) { t = st_load(fp, bf->h0); bcf_hdr_destroy(h); }
            bgzr_close(fp);
        }
        if (t == NULL) fprintf(stderr, "[E::%s] cannot read the sites of '%s'\n", __func__, fn);
        free(fn);
        pthread_mutex_lock(&g_sites_lock);
        bf->idx = t;
        bf->sites_loading = 0;
        pthread_cond_broadcast(&g_sites_cond);
    }
    pthread_mutex_unlock(&g_sites_lock);
    return bf->idx ? (const sitetab_t*)bf->idx : &empty;
}

/* the table a reader walks: its own (region) table if it has one, else the file's */
static const sitetab_t *sites_of(const bgt_t *bgt)
{
    const devrd_t *dv = (const devrd_t*)bgt->pb;
    return dv && dv->own_sites ? (const sitetab_t*)dv->own_sites : file_sites(bgt->f);
}

/* File rows [*r0, *r1) this reader can visit: the sites of its region, or from its start site on; 0 if that is
 * (nearly) the whole file.  A region query of a large database then loads a few 8192-row blocks of the .pbf
 * instead of all of it (the reference seeks to the nearest checkpoint, pbwt.c:349-372). */
static int needed_rows(const bgt_t *bgt, int64_t *r0, int64_t *r1)
{
    const region_t *r = (const region_t*)bgt->itr;
    const sitetab_t *t;
    int64_t i, lo, hi, mn = INT64_MAX, mx = -1;
    if (r == NULL && ((const cursor_t*)bgt->bcf)->next <= 0) return 0;     /* a whole-file walk: no need to wait for the table */
    t = sites_of(bgt);
    if (t->n == 0) return 0;
    if (r) {
        lo = r->at;
        for (hi = lo; hi < t->n && t->rid[hi] == r->tid && t->pos[hi] < r->end; ++hi) {}
    } else {
        lo = ((const cursor_t*)bgt->bcf)->next; hi = t->n;
        if (lo <= 0) return 0;
    }
    if (hi <= lo) { *r0 = *r1 = 0; return 1; }                  /* nothing to visit: an empty range */
    if (((const devrd_t*)bgt->pb)->own_sites == NULL && (hi - lo) * 2 > t->n) return 0;   /* most of the file anyway */
    for (i = lo; i < hi; ++i) { if (t->row[i] < mn) mn = t->row[i]; if (t->row[i] > mx) mx = t->row[i]; }
    *r0 = mn; *r1 = mx + 1;
    return 1;
}

/* BGT_TRACE=1: wall-clock of the stages of getting a database ready, on stderr (tuning aid) */
static double rd_now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static void rd_lap(double *t0, const char *what)
{
    const double t1 = rd_now_ms();
    if (getenv("BGT_TRACE")) fprintf(stderr, "[bgt trace]   %-32s %8.2f ms\n", what, t1 - *t0);
    *t0 = t1;
}

static int ensure_device(bgt_t *bgt)
{
    bgt_file_t *wf = (bgt_file_t*)bgt->f;
    devrd_t *dv = (devrd_t*)bgt->pb;
    char *fn;
    int64_t r0 = 0, r1 = 0;
    int partial;
    double t_lap = rd_now_ms();
    if (dv->rd) return 0;
    fn = (char*)malloc(strlen(wf->prefix) + 8);
    sprintf(fn, "%s.pbf", wf->prefix);
    partial = needed_rows(bgt, &r0, &r1) && r1 > r0;           /* (reads the site table: before taking the lock) */
    /* The whole-file image is shared by every reader of the file and loaded by the first that needs it -- outside the
     * lock, so that the databases of a merge load side by side (bgtm_prepare starts one thread per database). */
    pthread_mutex_lock(&g_open_lock);
    while (wf->gpu_opening) pthread_cond_wait(&g_open_cond, &g_open_lock);
    if (wf->gpu == NULL && partial) {
        pthread_mutex_unlock(&g_open_lock);
        dv->own_img = bgth_pbf_open_rows(fn, r0, r1, 0);         /* private to this reader */
        if (dv->own_img) dv->rd = bgth_reader_create(dv->own_img);
    } else {
        if (wf->gpu == NULL) {
            void *img;
            wf->gpu_opening = 1;
            pthread_mutex_unlock(&g_open_lock);
            img = open_whole_image(fn);                          /* the whole file */
            rd_lap(&t_lap, "image open");
            pthread_mutex_lock(&g_open_lock);
            wf->gpu = img; wf->gpu_opening = 0;
            pthread_cond_broadcast(&g_open_cond);
        }
        pthread_mutex_unlock(&g_open_lock);
        if (wf->gpu) dv->rd = bgth_reader_create((bgth_pbf_t*)wf->gpu);
        rd_lap(&t_lap, "reader create");
    }
    free(fn);
    if (dv->rd == NULL) { fprintf(stderr, "[E::%s] %s\n", __func__, bgth_last_error()); return -1; }
    return 0;
}

/* Extension for resident processes (the server): load the file's whole .pbf image into HBM and its site table now, so
 * that no query pays for either.  0, or -1 if the image could not be built (a later query tries again). */
int bgt_file_preload(const bgt_file_t *bf)
{
    bgt_file_t *wf = (bgt_file_t*)bf;
    if (wf == NULL) return -1;
    pthread_mutex_lock(&g_open_lock);
    while (wf->gpu_opening) pthread_cond_wait(&g_open_cond, &g_open_lock);
    if (wf->gpu == NULL) {
        char *fn = (char*)malloc(strlen(wf->prefix) + 8);
        void *img;
        sprintf(fn, "%s.pbf", wf->prefix);
        wf->gpu_opening = 1;
        pthread_mutex_unlock(&g_open_lock);
        img = open_whole_image(fn);
        if (img == NULL) fprintf(stderr, "[E::%s] %s\n", __func__, bgth_last_error());
        pthread_mutex_lock(&g_open_lock);
        wf->gpu = img; wf->gpu_opening = 0;
        pthread_cond_broadcast(&g_open_cond);
        free(fn);
    }
    pthread_mutex_unlock(&g_open_lock);
    (void)file_sites(bf);
    return wf->gpu ? 0 : -1;
}

/* selection + output configuration of the reader's device side (also after a change of image) */
static int apply_selection(bgt_t *bgt)
{
    devrd_t *dv = (devrd_t*)bgt->pb;
    int32_t *cols;
    int i, rc;
    if (bgt->n_out <= 0 || dv->rd == NULL) return 0;
    cols = (int32_t*)malloc((size_t)bgt->n_out * 2 * 4);
    for (i = 0; i < bgt->n_out; ++i) { cols[2 * i] = bgt->out[i] * 2; cols[2 * i + 1] = bgt->out[i] * 2 + 1; }
    rc = bgth_reader_select(dv->rd, bgt->n_out * 2, cols, dv->n_groups_total > 1 ? bgt->group : NULL,
                            dv->n_groups_total > 1 ? dv->n_groups_total : 1);
    if (rc < 0) fprintf(stderr, "[E::%s] %s\n", __func__, bgth_last_error());
    free(cols);
    if (rc >= 0) bgth_reader_config(dv->rd, dv->want, 0);
    return rc;
}

/* bring the device's allele-set folds of this database into the host totals (before the device reader goes away, and
 * before anybody looks at bm->alcnt / bm->hap) */
static int drain_folds(bgt_t *bgt)
{
    devrd_t *dv = (devrd_t*)bgt->pb;
    int32_t *c;
    uint64_t *h;
    int i, rc;
    if (!dv->folded || dv->rd == NULL || bgt->n_out <= 0) return 0;
    c = (int32_t*)malloc((size_t)bgt->n_out * 4);
    h = (uint64_t*)malloc((size_t)bgt->n_out * 16);
    rc = bgth_reader_take_folds(dv->rd, c, h);
    if (rc < 0) fprintf(stderr, "[E::%s] %s\n", __func__, bgth_last_error());
    else {
        if (dv->f_car == NULL) dv->f_car = (int32_t*)calloc((size_t)bgt->n_out, 4);
        if (dv->f_hap == NULL) dv->f_hap = (uint64_t*)calloc((size_t)bgt->n_out * 2, 8);
        for (i = 0; i < bgt->n_out; ++i) dv->f_car[i] += c[i];
        for (i = 0; i < bgt->n_out * 2; ++i) dv->f_hap[i] |= h[i];
        dv->folded = 0;
    }
    free(c); free(h);
    return rc;
}

/* a row outside the reader's partial image was asked for (the region or start changed after the image was
 * opened): switch to the shared image of the whole file */
static int promote_to_full(bgt_t *bgt)
{
    devrd_t *dv = (devrd_t*)bgt->pb;
    bgt_file_t *wf = (bgt_file_t*)bgt->f;
    if (dv->own_img == NULL) return -1;
    if (drain_folds(bgt) < 0) return -1;
    bgth_reader_destroy(dv->rd); dv->rd = NULL;
    bgth_pbf_close(dv->own_img); dv->own_img = NULL;
    pthread_mutex_lock(&g_open_lock);
    while (wf->gpu_opening) pthread_cond_wait(&g_open_cond, &g_open_lock);
    if (wf->gpu == NULL) {
        char *fn = (char*)malloc(strlen(wf->prefix) + 8);
        void *img;
        sprintf(fn, "%s.pbf", wf->prefix);
        wf->gpu_opening = 1;
        pthread_mutex_unlock(&g_open_lock);
        img = open_whole_image(fn);
        pthread_mutex_lock(&g_open_lock);
        wf->gpu = img; wf->gpu_opening = 0;
        pthread_cond_broadcast(&g_open_cond);
        free(fn);
    }
    pthread_mutex_unlock(&g_open_lock);
    if (wf->gpu) dv->rd = bgth_reader_create((bgth_pbf_t*)wf->gpu);
    if (dv->rd == NULL) return -1;
    return apply_selection(bgt);
}

static int prepare_one(bgt_t *bgt, int n_groups_total, int need_device)
{
    const fmf_t *f = bgt->f->f;
    devrd_t *dv = (devrd_t*)bgt->pb;
    int i, rc = 0;
    dv->skip_device = !need_device;
    if (need_device && ensure_device(bgt) < 0) rc = -1;
    if (bgt->n_groups == 0) add_group_core(bgt, BGT_SET_ALL_SAMPLES, NULL, NULL);
    for (i = 0, bgt->n_out = 0; i < f->n_rows; ++i) if (bgt->gtag[i] > 0) ++bgt->n_out;
    bgt->out = (int*)realloc(bgt->out, (size_t)(bgt->n_out ? bgt->n_out : 1) * sizeof(int));
    bgt->group = (uint32_t*)realloc(bgt->group, (size_t)(bgt->n_out ? bgt->n_out : 1) * 4);
    for (i = 0, bgt->n_out = 0; i < f->n_rows; ++i)
        if (bgt->gtag[i] > 0) { bgt->group[bgt->n_out] = bgt->gtag[i]; bgt->out[bgt->n_out++] = i; }
    dv->n_groups_total = n_groups_total;
    {
        double t_lap = rd_now_ms();
        if (bgt->n_out > 0 && dv->rd && apply_selection(bgt) < 0) rc = -1;
        rd_lap(&t_lap, "selection");
    }
    dv->site = -1;
    bgt->b0->shared.l = 0;
    return rc;
}

/* next site of this database: a region walks the overlapping sites in file order, otherwise the cursor
 * advances (ref bgt.c:272-288, :315-331; hts.c:868-900) */
static int64_t next_site(bgt_t *bgt)
{
    const sitetab_t *t = sites_of(bgt);
    region_t *r = (region_t*)bgt->itr;
    if (r) {
        while (!r->done && r->at < t->n) {
            const int64_t i = r->at++;
            if (t->rid[i] != r->tid || t->pos[i] >= r->end) { r->done = 1; break; }
            if (t->pos[i] + t->rlen[i] > r->beg) return i;
        }
        r->done = 1;
        return -1;
    } else {
        cursor_t *c = (cursor_t*)bgt->bcf;
        return c->next < t->n ? c->next++ : -1;
    }
}

static void fill_b0(bgt_t *bgt, int64_t i)
{
    const sitetab_t *t = sites_of(bgt);
    bcf_set_site(bgt->b0, t->rid[i], t->pos[i], t->rlen[i], t->pool + t->ref_off[i], t->ref_len[i],
                 t->pool + t->alt_off[i], t->alt_len[i], NULL);
    bgt->b0->n_allele = (uint32_t)t->n_allele[i];
}

/* pull one site and its genotype row (ref bgt.c:333-345) */
static int read_rec(bgt_t *bgt, bgt_rec_t *r)
{
    const sitetab_t *t = sites_of(bgt);
    devrd_t *dv = (devrd_t*)bgt->pb;
    const uint8_t **a;
    int64_t i;
    r->b0 = NULL; r->a[0] = r->a[1] = NULL;
    if (bgt->n_out == 0) return -1;
    for (;;) {                                                /* -B / -e: sites by BED overlap (ref bgt.c:315-331) */
        if ((i = next_site(bgt)) < 0) return -1;
        if (bgt->bed) {
            const int hit = bed_overlap(bgt->bed, bgt->f->h0->id[BCF_DT_CTG][t->rid[i]].key, t->pos[i], t->pos[i] + t->rlen[i]);
            if (bgt->bed_excl ? hit : !hit) continue;
        }
        if (bgt->h_al && !al_present((const alset_t*)bgt->h_al, bgt->f->h0->id[BCF_DT_CTG][t->rid[i]].key, t->rid[i], t->pos[i],
                                     t->rlen[i], t->pool + t->ref_off[i], t->ref_len[i], t->pool + t->alt_off[i], t->alt_len[i]))
            continue;                                         /* -a: sites of the allele set only (ref bgt.c:326) */
        break;
    }
    fill_b0(bgt, i);
    dv->site = i;
    if (dv->skip_device) {                                    /* `view -G` without -C/-f/-s groups: the output does */
        static const int32_t zero[3 * 33];                    /* not depend on a single genotype */
        dv->counts = zero; r->b0 = bgt->b0;
        return t->row[i];
    }
    if (dv->rd == NULL) return -2;
    if (bgth_reader_seek(dv->rd, t->row[i]) < 0 || (a = bgth_reader_read(dv->rd)) == NULL) {
        if (dv->own_img == NULL || promote_to_full(bgt) < 0 ||   /* outside a partial image: take the whole file */
            bgth_reader_seek(dv->rd, t->row[i]) < 0 || (a = bgth_reader_read(dv->rd)) == NULL) {
            fprintf(stderr, "[E::%s] %s\n", __func__, bgth_last_error());
            return -2;
        }
    }
    dv->counts = bgth_reader_last_counts(dv->rd);
    dv->gt8 = bgth_reader_last_gt8(dv->rd); dv->gttext = bgth_reader_last_gt_text(dv->rd);
    r->b0 = bgt->b0; r->a[0] = a[0]; r->a[1] = a[1];
    return t->row[i];
}

static const int8_t bits2gt[4] = {2, 4, 0, 6};              /* (allele+1)<<1 for REF, ALT, missing, <M> (ref bgt.c:250) */

static void gen_gt(const bcf_hdr_t *h, bcf1_t *b, int m, const uint8_t *const *a, const int32_t *mgs)
{                                                             /* ref bgt.c:290-313 */
    int i, m2 = m;
    b->indiv.l = 0;
    if (mgs) { for (i = m2 = 0; i < m; ++i) m2 += mgs[i] <= 1; if (m2 == 0) return; }
    b->n_fmt = 1; b->n_sample = (uint32_t)m2;
    bcf_enc_int1(&b->indiv, bcf_id2int(h, BCF_DT_ID, "GT"));
    bcf_enc_size(&b->indiv, 2, BCF_BT_INT8);
    ks_need(&b->indiv, (size_t)m2 * 2 + 1);
    for (i = 0; i < m << 1; ++i)
        if (!mgs || mgs[i >> 1] <= 1) b->indiv.s[b->indiv.l++] = (char)bits2gt[a[1][i] << 1 | a[0][i]];
    b->indiv.s[b->indiv.l] = 0;
}

/* the same from the vector the device already built (BGTH_WANT_GT8): 2m bytes, every sample kept */
static void gen_gt8(const bcf_hdr_t *h, bcf1_t *b, int m, const uint8_t *gt8)
{
    b->indiv.l = 0;
    if (m == 0) return;
    b->n_fmt = 1; b->n_sample = (uint32_t)m;
    bcf_enc_int1(&b->indiv, bcf_id2int(h, BCF_DT_ID, "GT"));
    bcf_enc_size(&b->indiv, 2, BCF_BT_INT8);
    ks_need(&b->indiv, (size_t)m * 2 + 1);
    memcpy(b->indiv.s + b->indiv.l, gt8, (size_t)m * 2);
    b->indiv.l += (size_t)m * 2;
    b->indiv.s[b->indiv.l] = 0;
}

int bgt_read(bgt_t *bgt, bcf1_t *b)                           /* ref bgt.c:347-356 */
{
    bgt_rec_t r;
    int ret;
    if (bgt->h_out == NULL) {
        devrd_t *dv = (devrd_t*)bgt->pb;
        kstring_t s = {0, 0, 0};
        int i;
        prepare_one(bgt, 1, 1);
        dv->want = BGTH_WANT_PLANES;
        if (dv->rd) bgth_reader_config(dv->rd, dv->want, 0);
        bgt->h_out = bcf_hdr_init();                          /* input header + FORMAT + the selected samples */
        ks_putn(&s, bgt->f->h0->text, bgt->f->h0->l_text > 0 ? (size_t)bgt->f->h0->l_text : 0);
        while (s.l && s.s[s.l - 1] == 0) --s.l;
        if (bgt->n_out > 0) {
            ks_puts(&s, "\tFORMAT");
            for (i = 0; i < bgt->n_out; ++i) { ks_putc(&s, '\t'); ks_puts(&s, bgt->f->f->rows[bgt->out[i]].name); }
        }
        bgt->h_out->text = s.s; bgt->h_out->l_text = (int32_t)s.l + 1;
        bcf_hdr_parse(bgt->h_out);
    }
    if ((ret = read_rec(bgt, &r)) < 0) return ret;
    {   /* bcfcpy(b, r.b0) of the reference: the site record exactly as the file holds it (ID, every allele, FILTER,
         * INFO with _row), then the genotypes */
        const sitetab_t *t = sites_of(bgt);
        const int64_t i = ((devrd_t*)bgt->pb)->site;
        b->rid = t->rid[i]; b->pos = t->pos[i]; b->rlen = t->rlen[i]; b->qual = t->qual[i];
        b->n_info = t->n_info[i]; b->n_allele = (uint32_t)t->n_allele[i]; b->n_fmt = 0; b->n_sample = 0;
        b->shared.l = 0; ks_putn(&b->shared, t->pool + t->raw_off[i], t->raw_len[i]);
        b->unpacked = 0;
    }
    gen_gt(bgt->h_out, b, bgt->n_out, r.a, NULL);
    return ret;
}

/* ------------------------------------------------------------------------------------------------
 * multi-database reader
 * ------------------------------------------------------------------------------------------------ */
bgtm_t *bgtm_reader_init(int n_files, bgt_file_t *const *bf)
{
    bgtm_t *bm = (bgtm_t*)calloc(1, sizeof(*bm));
    int i;
    bm->n_bgt = n_files;
    bm->bgt = (bgt_t**)calloc((size_t)n_files, sizeof(void*));
    for (i = 0; i < n_files; ++i) bm->bgt[i] = bgt_reader_init(bf[i]);
    bm->r = (bgt_rec_t*)calloc((size_t)n_files, sizeof(bgt_rec_t));
    return bm;
}

void bgtm_reader_destroy(bgtm_t *bm)
{
    int i;
    if (!bm) return;
    free(bm->hap); free(bm->alcnt);
    for (i = 0; i < bm->n_aal; ++i) free(bm->aal[i].chr.s);
    free(bm->aal);
    if (bm->h_al) {
        alset_t *h = (alset_t*)bm->h_al;
        for (i = 0; i < h->n; ++i) free(h->key[i]);
        free(h->key); free(h);
    }
    if (bm->site_flt) ke_destroy(bm->site_flt);
    free(bm->mgs); free(bm->group); free(bm->sample_idx);
    if (bm->h_out) bcf_hdr_destroy(bm->h_out);
    free(bm->a[0]); free(bm->a[1]);
    for (i = 0; i < bm->n_fields; ++i) ke_destroy(bm->fields[i]);
    free(bm->fields); free(bm->tbl_line.s);
    for (i = 0; i < bm->n_bgt; ++i) bgt_reader_destroy(bm->bgt[i]);
    free(bm->r); free(bm->bgt); free(bm);
}

int bgtm_add_group(bgtm_t *bm, const char *expr)              /* ref bgt.c:408-416 */
{
    int i, ret = 0, size = 0;
    for (i = 0; i < bm->n_bgt; ++i) {
        if ((ret = add_group(bm->bgt[i], expr)) < 0) break;
        size += ret;
    }
    if (i == bm->n_bgt) ++bm->n_groups;
    return i == bm->n_bgt ? size : ret;
}

int bgtm_set_region(bgtm_t *bm, const char *reg)
{
    int i, ret = 0;
    for (i = 0; i < bm->n_bgt; ++i) if ((ret = bgt_set_region(bm->bgt[i], reg)) < 0) break;
    return ret;
}

int bgtm_set_start(bgtm_t *bm, int64_t n) { int i; for (i = 0; i < bm->n_bgt; ++i) bgt_set_start(bm->bgt[i], n); return 0; }
void bgtm_set_bed(bgtm_t *bm, const void *bed, int excl) { int i; for (i = 0; i < bm->n_bgt; ++i) bgt_set_bed(bm->bgt[i], bed, excl); }
void bgtm_set_flag(bgtm_t *bm, int flag) { bm->flag = flag; }

int bgtm_set_flt_site(bgtm_t *bm, const char *expr)           /* ref bgt.c:444-455: non-zero = parse error bits */
{
    int err;
    if (bm->site_flt) ke_destroy(bm->site_flt);
    bm->site_flt = ke_parse(expr, &err);
    if (err != 0) { bm->site_flt = NULL; return err; }
    return 0;
}

int bgtm_set_mgs(bgtm_t *bm, int mgs_def)
{
    int i;
    for (i = 0; i < bm->n_bgt; ++i) bm->bgt[i]->mgs_def = mgs_def;
    bm->mgs_def = mgs_def;
    return 0;
}

/* -t: comma-separated expressions, commas inside parentheses do not split (ref bgt.c:547-593) */
int bgtm_set_table(bgtm_t *bm, const char *fmt)
{
    int n = 0, m = 0, depth = 0, i, ok = 1;
    char **piece = NULL;
    const char *p, *q;
    for (i = 0; i < bm->n_fields; ++i) ke_destroy(bm->fields[i]);
    free(bm->fields); bm->fields = NULL; bm->n_fields = 0;
    for (q = p = fmt;; ++p) {
        if (*p == '(') ++depth;
        else if (*p == ')') --depth;
        else if (*p == 0 || (*p == ',' && depth == 0)) {
            if (n == m) { m = m ? m << 1 : 16; piece = (char**)realloc(piece, (size_t)m * sizeof(char*)); }
            piece[n] = (char*)calloc((size_t)(p - q) + 1, 1);
            memcpy(piece[n++], q, (size_t)(p - q));
            q = p + 1;
            if (*p == 0) break;
        }
    }
    if (depth != 0) ok = 0;
    if (ok) {
        bm->fields = (kexpr_t**)calloc((size_t)n, sizeof(kexpr_t*));
        for (i = 0; i < n; ++i) {
            int err;
            bm->fields[i] = ke_parse(piece[i], &err);
            if (err) { ok = 0; break; }
        }
        if (!ok) {
            int j;
            for (j = 0; j <= i && j < n; ++j) if (bm->fields[j]) ke_destroy(bm->fields[j]);
            free(bm->fields); bm->fields = NULL;
        } else bm->n_fields = n;
    }
    for (i = 0; i < n; ++i) free(piece[i]);
    free(piece);
    return ok ? 0 : -1;
}
/* ------------------------------------------------------------------------------------------------
 * allele sets (-a): "chr:pos:rlen:alt" / "chr:pos:REF:ALT" / "chr:pos::REF" strings, one per comma or per
 * line of a file.  An allele is kept as {chr, 0-based pos, rlen, alt} with the bases common to both ends of
 * REF and ALT removed; a record matches the set through its first ALT (or, for a "reference allele" query,
 * through its REF).  Restated from reference bgt.c:976-1060 (parsing, normal form), :252-270 (matching),
 * :477-544 (the set, and the region it implies).
 * ------------------------------------------------------------------------------------------------ */

static int alset_has(const alset_t *h, const char *k)
{
    int i;
    for (i = 0; i < h->n; ++i) if (strcmp(h->key[i], k) == 0) return 1;
    return 0;
}

int bgt_al_parse(const char *al, bgt_allele_t *a)
{
    const char *p = al, *ref = NULL, *alt;
    int off, i, tmp;
    a->chr.l = 0; a->al = NULL; a->pos = -1; a->rlen = -1; a->rid = -1;
    while (*p && *p != ':') ++p;
    if (*p == 0) return -1;
    ks_putn(&a->chr, al, (size_t)(p - al)); ks_putc(&a->chr, 0);
    ++p;
    if (!isdigit((unsigned char)*p)) return -1;
    a->pos = (int)strtol(p, (char**)&p, 10) - 1;
    if (*p != ':') return -1;
    ++p;
    if (isdigit((unsigned char)*p)) a->rlen = (int)strtol(p, (char**)&p, 10);       /* reference length ... */
    else if (isalpha((unsigned char)*p)) {                                            /* ... or the reference bases */
        ref = p;
        while (isalpha((unsigned char)*p)) ++p;
        a->rlen = (int)(p - ref);
    }                                                                                 /* or empty: as long as the allele */
    if (*p != ':') return -1;
    alt = ++p;
    if (a->rlen < 0) { for (i = 0; isalpha((unsigned char)alt[i]); ++i) {} a->rlen = i; }
    for (off = 0; *p && isalpha((unsigned char)*p); ++p) {                            /* bases shared at the left end */
        if (ref && toupper((unsigned char)*p) == toupper((unsigned char)ref[off])) ++off;
        else break;
    }
    a->pos += off; a->rlen -= off;
    tmp = (int)a->chr.l;
    ks_puts(&a->chr, alt + off);
    a->al = a->chr.s + tmp;
    if (ref) {                                                                        /* and at the right end */
        const int l_alt = (int)(a->chr.s + a->chr.l - a->al);
        const int min_l = l_alt < a->rlen ? l_alt : a->rlen;
        ref += off;
        for (off = 0; off < min_l && isalpha((unsigned char)ref[a->rlen - 1 - off]) &&
             toupper((unsigned char)ref[a->rlen - 1 - off]) == toupper((unsigned char)a->al[l_alt - 1 - off]); ++off) {}
        a->rlen -= off;
        a->al[l_alt - off] = 0;
        a->chr.l -= (size_t)off;
    }
    return 0;
}

void bgt_al_format(const bgt_allele_t *a, kstring_t *s)
{
    s->l = 0;
    ks_putn(s, a->chr.s, (size_t)(a->al - a->chr.s - 1)); ks_putc(s, ':');
    ks_puti(s, a->pos); ks_putc(s, ':'); ks_puti(s, a->rlen); ks_putc(s, ':');
    ks_putn(s, a->al, (size_t)(a->chr.s + a->chr.l - a->al));
}

/* the first ALT of a record (and optionally its REF) in the same normal form */
static void al_from_site(const char *chr, int rid, int pos, int rlen, const char *ref, int l_ref, const char *alt, int l_alt,
                         bgt_allele_t *a, bgt_allele_t *r)
{
    const int min_l = l_ref < l_alt ? l_ref : l_alt, l_chr = (int)strlen(chr);
    int shift;
    for (shift = 0; shift < min_l && ref[shift] == alt[shift]; ++shift) {}
    a->rid = rid; a->pos = pos + shift; a->rlen = rlen - shift;
    a->chr.l = 0;
    ks_putn(&a->chr, chr, (size_t)l_chr); ks_putc(&a->chr, 0);
    ks_putn(&a->chr, alt + shift, (size_t)(l_alt - shift));
    a->al = a->chr.s + l_chr + 1;
    if (r) {
        r->rid = rid; r->pos = pos + shift; r->rlen = rlen - shift;
        r->chr.l = 0;
        ks_putn(&r->chr, chr, (size_t)l_chr); ks_putc(&r->chr, 0);
        ks_putn(&r->chr, ref + shift, (size_t)(l_ref - shift));
        r->al = r->chr.s + l_chr + 1;
    }
}

void bgt_al_from_bcf(const bcf_hdr_t *h, const bcf1_t *b, bgt_allele_t *a, bgt_allele_t *r)
{
    const uint8_t *p = (const uint8_t*)b->shared.s, *q, *ref, *alt;
    int type, n, l_ref, l_alt;
    n = bcf_dec_size(p, &q, &type); p = q + n;                                        /* ID */
    l_ref = bcf_dec_size(p, &q, &type); ref = q; p = q + l_ref;
    l_alt = b->n_allele > 1 ? bcf_dec_size(p, &q, &type) : 0; alt = q;
    al_from_site(h->id[BCF_DT_CTG][b->rid].key, b->rid, b->pos, b->rlen, (const char*)ref, l_ref, (const char*)alt, l_alt, a, r);
}

/* 0 = the site is not in the set, 1 = its first ALT is, 2 = its REF is (ref bgt.c:252-270) */
static int al_present(const alset_t *h, const char *chr, int rid, int pos, int rlen, const char *ref, int l_ref,
                      const char *alt, int l_alt)
{
    bgt_allele_t a, r;
    kstring_t s = {0, 0, 0};
    int ret = 0;
    memset(&a, 0, sizeof(a)); memset(&r, 0, sizeof(r));
    al_from_site(chr, rid, pos, rlen, ref, l_ref, alt, l_alt, &a, &r);
    bgt_al_format(&a, &s);
    if (alset_has(h, s.s)) ret = 1;
    else { bgt_al_format(&r, &s); if (alset_has(h, s.s)) ret = 2; }
    free(s.s); free(a.chr.s); free(r.chr.s);
    return ret;
}

static int is_readable_file(const char *fn)                   /* ref bgt.c:163-173 */
{
    FILE *fp;
    if (bgt_no_file) return 0;
    if ((fp = fopen(fn, "r")) == NULL) return 0;
    fclose(fp);
    return 1;
}

/* Row names of an FMF file whose metadata satisfy `ke`, reading line by line.  Unlike the in-memory test
 * (fmf_test) a value typed `f` is bound as a real and `_ROW_` is bound for every row (ref fmf.c:185-218). */
static char **fmf_stream_select(const char *fn, kexpr_t *ke, int *n_out)
{
    gzFile fp = fn && strcmp(fn, "-") ? gzopen(fn, "r") : gzdopen(0, "r");
    char **names = NULL, *line = NULL;
    int n = 0, m = 0, c;
    size_t l = 0, cap = 0;
    *n_out = 0;
    if (fp == NULL) return NULL;
    names = (char**)calloc(1, sizeof(char*));
    for (;;) {
        c = gzgetc(fp);
        if (c != -1 && c != '\n') {
            if (l + 2 > cap) { cap = cap ? cap << 1 : 256; line = (char*)realloc(line, cap); }
            line[l++] = (char)c;
            continue;
        }
        if (l > 0) {
            char *p, *q;
            int field = 0, err = 0, yes;
            line[l] = 0;
            ke_unset(ke);
            for (p = q = line;; ++p) {
                if (*p == 0 || *p == '\t') {
                    const int last = *p == 0;
                    *p = 0;
                    if (field == 0) ke_set_str(ke, "_ROW_", q);
                    else {
                        char *r = q;
                        while (*r && *r != ':') ++r;
                        if (*r == ':' && p - r >= 3) {
                            *r = 0;
                            if (r[1] == 'i') ke_set_int(ke, q, strtol(r + 3, NULL, 0));
                            else if (r[1] == 'f') ke_set_real(ke, q, strtod(r + 3, NULL));
                            else ke_set_str(ke, q, r + 3);
                            *r = ':';
                        }
                    }
                    q = p + 1; ++field;
                    if (last) break;
                }
            }
            yes = !!ke_eval_int(ke, &err);
            if (!err && yes) {
                if (n == m) { m = m ? m << 1 : 16; names = (char**)realloc(names, (size_t)m * sizeof(char*)); }
                names[n++] = strdup(line);                    /* the first field: the tabs were turned into terminators */
            }
        }
        l = 0;
        if (c == -1) break;
    }
    free(line);
    gzclose(fp);
    *n_out = n;
    return names;
}

int bgtm_set_alleles(bgtm_t *bm, const char *expr, const fmf_t *f, const char *fn)
{
    int i, n = 0, n_al = 0, diff_chr = 0, min_pos = INT_MAX, max_pos = INT_MIN;
    char **lines;
    bgt_allele_t *al;
    alset_t *h;
    kstring_t s = {0, 0, 0};
    if (!(*expr == ':' || *expr == ',' || (*expr != '?' && is_readable_file(expr)) || (f == NULL && fn == NULL && is_readable_file(expr)))) {
        /* an expression on the rows of a variant annotation file (-d): the names of the rows it accepts */
        int err;
        kexpr_t *ke;
        if (f == NULL && fn == NULL) return -1;
        ke = ke_parse(expr, &err);
        if (err) { if (ke) ke_destroy(ke); return -1; }
        if (f) {                                              /* -M: the file is in memory (ref bgt.c:499-502) */
            lines = (char**)calloc((size_t)(f->n_rows ? f->n_rows : 1), sizeof(char*));
            for (i = 0; i < f->n_rows; ++i) if (fmf_test(f, i, ke)) lines[n++] = strdup(f->rows[i].name);
        } else lines = fmf_stream_select(fn, ke, &n);          /* streamed (ref bgt.c:504-509, fmf.c:185-218) */
        ke_destroy(ke);
        if (lines == NULL) return -1;
    } else if ((lines = read_names(expr, &n)) == NULL) return -1;   /* ",a,b" / ":a,b" / a file, one allele per line */
    al = (bgt_allele_t*)calloc((size_t)(n ? n : 1), sizeof(*al));
    for (i = 0; i < n; ++i) {
        if (bgt_al_parse(lines[i], &al[n_al]) == 0) ++n_al;
        free(lines[i]);
    }
    free(lines);
    if (n_al == 0) { for (i = 0; i < n; ++i) free(al[i].chr.s); free(al); return 0; }
    h = (alset_t*)calloc(1, sizeof(*h));
    for (i = 0; i < n_al; ++i) {
        bgt_al_format(&al[i], &s);
        if (!alset_has(h, s.s)) {
            if (h->n == h->m) { h->m = h->m ? h->m << 1 : 16; h->key = (char**)realloc(h->key, (size_t)h->m * sizeof(char*)); }
            h->key[h->n++] = strdup(s.s);
            if (al[i].pos < min_pos) min_pos = al[i].pos;
            if (al[i].pos > max_pos) max_pos = al[i].pos;
            if (strcmp(al[i].chr.s, al[0].chr.s) != 0) diff_chr = 1;
        }
    }
    free(s.s);
    if (!diff_chr && bm->n_bgt > 0 && bm->bgt[0]->itr == NULL) {  /* one chromosome, no -r: the span of the alleles */
        char *reg = (char*)malloc(strlen(al[0].chr.s) + 32);
        sprintf(reg, "%s:%d-%d", al[0].chr.s, min_pos + 1, max_pos + 1);
        bgtm_set_region(bm, reg);
        free(reg);
    }
    for (i = 0; i < n; ++i) free(al[i].chr.s);
    free(al);
    bm->h_al = h;
    for (i = 0; i < bm->n_bgt; ++i) bm->bgt[i]->h_al = h;
    return h->n;
}

/* ------------------------------------------------------------------------------------------------
 * -H: count the distinct haplotypes over the alleles of the set (ref bgt.c:896-955).  Haplotypes are numbered
 * in order of first appearance and then sorted by decreasing total; the order among EQUAL totals is whatever
 * the reference's sort leaves (klib introsort: median-of-three quicksort that leaves stretches of <= 16 to
 * one final insertion sort, comb sort once the depth budget is spent), so the same steps are followed here.
 * ------------------------------------------------------------------------------------------------ */
#define HC_BEFORE(x, y) ((x).tot > (y).tot)
#define HC_SWAP(x, y) do { bgt_hapcnt_t t_ = (x); (x) = (y); (y) = t_; } while (0)

static void hc_insertion(bgt_hapcnt_t *a, int n)
{
    int i, j;
    for (i = 1; i < n; ++i)
        for (j = i; j > 0 && HC_BEFORE(a[j], a[j - 1]); --j) HC_SWAP(a[j], a[j - 1]);
}

static void hc_comb(bgt_hapcnt_t *a, int n)
{
    const double shrink = 1.2473309501039786540366528676643;
    int gap = n, swapped, i;
    do {
        if (gap > 2) { gap = (int)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
        swapped = 0;
        for (i = 0; i + gap < n; ++i)
            if (HC_BEFORE(a[i + gap], a[i])) { HC_SWAP(a[i], a[i + gap]); swapped = 1; }
    } while (swapped || gap > 2);
    if (gap != 1) hc_insertion(a, n);
}

static void hc_sort(bgt_hapcnt_t *a, int n)
{
    struct { int lo, hi, depth; } stack[160];
    int top = 0, lo, hi, depth;
    if (n < 1) return;
    if (n == 2) { if (HC_BEFORE(a[1], a[0])) HC_SWAP(a[0], a[1]); return; }
    for (depth = 2; (1ul << depth) < (unsigned long)n; ++depth) {}
    depth <<= 1;
    lo = 0; hi = n - 1;
    for (;;) {
        if (lo < hi) {
            int i = lo, j = hi, k = lo + ((hi - lo) >> 1) + 1;
            bgt_hapcnt_t pivot;
            if (--depth == 0) { hc_comb(a + lo, hi - lo + 1); hi = lo; continue; }
            if (HC_BEFORE(a[k], a[i])) { if (HC_BEFORE(a[k], a[j])) k = j; }      /* median of first, middle, last */
            else k = HC_BEFORE(a[j], a[i]) ? i : j;
            pivot = a[k];
            if (k != hi) HC_SWAP(a[k], a[hi]);
            for (;;) {
                do ++i; while (HC_BEFORE(a[i], pivot));
                do --j; while (i <= j && HC_BEFORE(pivot, a[j]));
                if (j <= i) break;
                HC_SWAP(a[i], a[j]);
            }
            HC_SWAP(a[i], a[hi]);
            if (i - lo > hi - i) {                            /* the larger side waits on the stack if it is > 16 */
                if (i - lo > 16) { stack[top].lo = lo; stack[top].hi = i - 1; stack[top].depth = depth; ++top; }
                lo = hi - i > 16 ? i + 1 : hi;
            } else {
                if (hi - i > 16) { stack[top].lo = i + 1; stack[top].hi = hi; stack[top].depth = depth; ++top; }
                hi = i - lo > 16 ? i - 1 : lo;
            }
        } else if (top == 0) { hc_insertion(a, n); return; }
        else { --top; lo = stack[top].lo; hi = stack[top].hi; depth = stack[top].depth; }
    }
}

/* bm->alcnt / bm->hap = the per-database totals, after draining what the devices still hold */
static int sync_folds(bgtm_t *bm)
{
    int i, off = 0, rc = 0;
    for (i = 0; i < bm->n_bgt; ++i) {
        bgt_t *bgt = bm->bgt[i];
        const devrd_t *dv = (const devrd_t*)bgt->pb;
        if (drain_folds(bgt) < 0) rc = -1;
        if (bm->alcnt && dv->f_car) memcpy(bm->alcnt + off, dv->f_car, (size_t)bgt->n_out * 4);
        if (bm->hap && dv->f_hap) memcpy(bm->hap + 2 * (size_t)off, dv->f_hap, (size_t)bgt->n_out * 16);
        off += bgt->n_out;
    }
    return rc;
}

bgt_hapcnt_t *bgtm_hapcnt(const bgtm_t *bm, int *n_hap)
{
    bgt_hapcnt_t *hc = NULL;
    int i, j, n = 0, m = 0, n_slot = 64, *slot;
    *n_hap = 0;
    if (bm->hap == NULL || bm->n_out == 0 || sync_folds((bgtm_t*)bm) < 0) return NULL;
    while (n_slot < bm->n_out * 4) n_slot <<= 1;              /* open addressing: haplotype -> its number */
    slot = (int*)malloc((size_t)n_slot * sizeof(int));
    for (i = 0; i < n_slot; ++i) slot[i] = -1;
    for (i = 0; i < bm->n_out << 1; ++i) {
        const uint64_t h = bm->hap[i];
        uint64_t k = (h * 0x9E3779B97F4A7C15ull) >> 20 & (uint64_t)(n_slot - 1);
        while (slot[k] >= 0 && hc[slot[k]].hap != h) k = (k + 1) & (uint64_t)(n_slot - 1);
        if (slot[k] < 0) {                                    /* first appearance */
            if (n == m) { m = m ? m << 1 : 16; hc = (bgt_hapcnt_t*)realloc(hc, (size_t)m * sizeof(*hc)); }
            hc[n].hap = h; hc[n].tot = 0; hc[n].cnt = (int*)calloc((size_t)bm->n_groups, sizeof(int));
            slot[k] = n++;
        }
        ++hc[slot[k]].tot;
        for (j = 0; j < bm->n_groups; ++j)                    /* the group id read as a bit mask, as the reference does */
            if (bm->group[i >> 1] & 1U << j) ++hc[slot[k]].cnt[j];
    }
    free(slot);
    hc_sort(hc, n);
    *n_hap = n;
    return hc;
}

char *bgtm_hapcnt_print_destroy(const bgtm_t *bm, int n_hap, bgt_hapcnt_t *hc)
{
    kstring_t s = {0, 0, 0};
    int i, j;
    ks_printf(&s, "NA\t%d\n", bm->n_aal);
    for (i = 0; i < bm->n_aal; ++i) {
        const bgt_allele_t *a = &bm->aal[i];
        ks_printf(&s, "AA\t%s:%d:%d:%s\n", a->chr.s, a->pos + 1, a->rlen, a->al);
    }
    ks_printf(&s, "NH\t%d\t%d\n", n_hap, bm->n_groups);
    for (i = 0; i < n_hap; ++i) {
        ks_puts(&s, "HC\t");
        for (j = 0; j < bm->n_aal; ++j) ks_putc(&s, (char)('0' + (hc[i].hap >> j & 1)));
        for (j = 0; j < bm->n_groups; ++j) ks_printf(&s, "\t%d", hc[i].cnt[j]);
        ks_putc(&s, '\n');
        free(hc[i].cnt);
    }
    free(hc);
    return s.s;
}

/* -S: the samples that carry every allele of the set (ref bgt.c:957-970) */
char *bgtm_alcnt_print(const bgtm_t *bm)
{
    kstring_t s = {0, 0, 0};
    int i;
    if (bm->alcnt == NULL || sync_folds((bgtm_t*)bm) < 0) return NULL;
    for (i = 0; i < bm->n_out; ++i) {
        if (bm->alcnt[i] == bm->n_aal) {
            const bgt_t *bgt = bm->bgt[bm->sample_idx[i] >> 32];
            if (bm->mgs[i] > 1) continue;
            ks_printf(&s, "SP\t%s\t%d\n", bgt->f->f->rows[(uint32_t)bm->sample_idx[i]].name, (int)(bm->sample_idx[i] >> 32) + 1);
        }
    }
    return s.s;
}

typedef struct { bgt_t *bgt; int n_groups, need_device, rc, started; pthread_t th; } prep_job_t;
static void *prep_worker(void *p)
{
    prep_job_t *j = (prep_job_t*)p;
    j->rc = prepare_one(j->bgt, j->n_groups, j->need_device);
    return NULL;
}

/* merged sample list, groups, output header, device selections (ref bgt.c:597-676) */
/* ---- the merged output: who is in it, and the header that announces it -------------------------------------------------
 * Output sample t of a merge is sample out[j] of database i, databases in argv order, samples in .spl order
 * (reference bgt.c:611-620).  One pass over the databases' selections fills the three per-sample tables; a merge that
 * shows no genotype column at all (every sample masked by its minimal group size) becomes a -G run. */
static void merged_sample_table(bgtm_t *bm)
{
    const size_t cap = (size_t)(bm->n_out ? bm->n_out : 1);
    int db, t = 0, shown = 0;
    bm->mgs = (int32_t*)realloc(bm->mgs, cap * sizeof(int32_t));
    bm->group = (uint32_t*)realloc(bm->group, cap * sizeof(uint32_t));
    bm->sample_idx = (uint64_t*)realloc(bm->sample_idx, cap * sizeof(uint64_t));
    for (db = 0; db < bm->n_bgt; ++db) {
        const bgt_t *one = bm->bgt[db];
        const int32_t *file_mgs = one->f->mgs;
        int j;
        for (j = 0; j < one->n_out; ++j, ++t) {
            const int smp = one->out[j];
            bm->sample_idx[t] = (uint64_t)db << 32 | (uint32_t)smp;
            bm->group[t] = bm->n_groups ? one->group[j] : 1;                /* no -s at all: everybody is group 1 */
            bm->mgs[t] = file_mgs[smp] >= 0 ? file_mgs[smp] : bm->mgs_def;
            shown += bm->mgs[t] <= 1;
        }
    }
    if (bm->n_groups == 0) bm->n_groups = 1;
    if (shown == 0) bm->flag |= BGT_F_NO_GT;
}

/* The header's meta lines as data: the goldens pin every byte (the reference prints them at bgt.c:628-658).  A count field
 * comes in two flavours: the cohort's and, with suffix 1..n_groups, one per sample group. */
typedef struct { const char *id, *what; } hdr_item_t;
static const hdr_item_t k_hdr_counts[] = { {"AC", "Count of alternate alleles"}, {"AN", "Count of total alleles"} };
static const hdr_item_t k_hdr_alts[] = {
    {"M", "Multi-allele"}, {"DEL", "Deletion"}, {"DUP", "Duplication"}, {"INS", "Insertion"}, {"INV", "Inversion"},
    {"DUP:TANDEM", "Tandem duplication"}, {"DEL:ME", "Deletion of mobile element"}, {"INS:ME", "Insertion of mobile element"} };
#define N_ITEMS(a) ((int)(sizeof(a) / sizeof((a)[0])))

static void header_text(const bgtm_t *bm, kstring_t *h)
{
    const bcf_hdr_t *first = bm->bgt[0]->f->h0;                             /* contigs: the first database speaks for all */
    int g, k, db, t = 0;
    ks_puts(h, "##fileformat=VCFv4.1\n");
    for (k = 0; k < N_ITEMS(k_hdr_counts); ++k)
        ks_printf(h, "##INFO=<ID=%s,Number=A,Type=String,Description=\"%s\">\n", k_hdr_counts[k].id, k_hdr_counts[k].what);
    for (g = 1; g <= bm->n_groups; ++g)
        for (k = 0; k < N_ITEMS(k_hdr_counts); ++k)
            ks_printf(h, "##INFO=<ID=%s%d,Number=A,Type=String,Description=\"%s for sample group %d\">\n", k_hdr_counts[k].id, g,
                      k_hdr_counts[k].what, g);
    ks_puts(h, "##INFO=<ID=END,Number=1,Type=Integer,Description=\"Ending position\">\n"
               "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n");
    for (k = 0; k < N_ITEMS(k_hdr_alts); ++k)
        ks_printf(h, "##ALT=<ID=%s,Description=\"%s\">\n", k_hdr_alts[k].id, k_hdr_alts[k].what);
    for (k = 0; k < first->n[BCF_DT_CTG]; ++k)
        ks_printf(h, "##contig=<ID=%s,length=%d>\n", first->id[BCF_DT_CTG][k].key, first->id[BCF_DT_CTG][k].val->info[0]);
    ks_puts(h, "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO");
    if (bm->flag & BGT_F_NO_GT) return;
    ks_puts(h, "\tFORMAT");
    for (db = 0; db < bm->n_bgt; ++db) {                                    /* one column per sample that may be shown */
        const bgt_t *one = bm->bgt[db];
        int j;
        for (j = 0; j < one->n_out; ++j)
            if (bm->mgs[t++] <= 1) { ks_putc(h, '\t'); ks_puts(h, one->f->f->rows[one->out[j]].name); }
    }
}

int bgtm_prepare(bgtm_t *bm)
{
    kstring_t h = {0, 0, 0};
    int i, need_counts, rc = 0;
    if (bm->n_bgt == 0) return 0;
    /* does any output depend on a genotype?  not for `-G` without -C / -f / several groups (ref bgt.c:850) */
    need_counts = (bm->flag & BGT_F_SET_AC) || bm->site_flt || bm->n_fields > 0 || bm->n_groups > 1;
    for (i = 0; i < bm->n_bgt; ++i) {                          /* whole-file walks: the site tables load beside the images */
        pthread_t th;
        const devrd_t *dv = (const devrd_t*)bm->bgt[i]->pb;
        if (bm->bgt[i]->itr == NULL && dv && dv->own_sites == NULL && sites_prefetch(bm->bgt[i]->f, &th)) pthread_detach(th);
    }
    {   /* the databases of a merge get ready side by side: each one's .pbf image loads on its own host thread */
        const int need_device = !(bm->flag & BGT_F_NO_GT) || need_counts || (bm->h_al && (bm->flag & (BGT_F_CNT_AL | BGT_F_CNT_HAP)));
        prep_job_t *job = (prep_job_t*)calloc((size_t)bm->n_bgt, sizeof(prep_job_t));
        for (i = 0; i < bm->n_bgt; ++i) {
            job[i].bgt = bm->bgt[i]; job[i].n_groups = bm->n_groups; job[i].need_device = need_device;
            if (i > 0 && need_device && pthread_create(&job[i].th, NULL, prep_worker, &job[i]) == 0) job[i].started = 1;
        }
        for (i = 0; i < bm->n_bgt; ++i) if (!job[i].started) prep_worker(&job[i]);
        for (i = bm->n_out = 0; i < bm->n_bgt; ++i) {
            if (job[i].started) pthread_join(job[i].th, NULL);
            if (job[i].rc < 0) rc = -1;
            bm->n_out += bm->bgt[i]->n_out;
        }
        free(job);
    }
    merged_sample_table(bm);
    header_text(bm, &h);
    if (bm->h_out) bcf_hdr_destroy(bm->h_out);
    bm->h_out = bcf_hdr_init();
    bm->h_out->l_text = (int32_t)h.l + 1; bm->h_out->m_text = (int32_t)h.m; bm->h_out->text = h.s;
    bcf_hdr_parse(bm->h_out);

    bm->a[0] = (uint8_t*)realloc(bm->a[0], (size_t)(bm->n_out ? bm->n_out : 1) << 1);
    bm->a[1] = (uint8_t*)realloc(bm->a[1], (size_t)(bm->n_out ? bm->n_out : 1) << 2);   /* planes: 2 B, text: 4 B per sample */

    if (bm->h_al) {                                           /* ref bgt.c:668-674 */
        for (i = 0; i < bm->n_bgt; ++i) {
            devrd_t *dv = (devrd_t*)bm->bgt[i]->pb;
            free(dv->f_car); free(dv->f_hap); dv->f_car = NULL; dv->f_hap = NULL; dv->folded = 0;
        }
        free(bm->alcnt); bm->alcnt = NULL;
        if (bm->flag & BGT_F_CNT_AL) bm->alcnt = (int*)calloc((size_t)(bm->n_out ? bm->n_out : 1), sizeof(int));
        free(bm->hap); bm->hap = NULL;
        if (bm->flag & BGT_F_CNT_HAP) bm->hap = (uint64_t*)calloc((size_t)(bm->n_out ? bm->n_out : 1) << 1, 8);
        for (i = 0; i < bm->n_aal; ++i) free(bm->aal[i].chr.s);
        free(bm->aal);
        bm->n_aal = 0;
        bm->aal = (bgt_allele_t*)calloc((size_t)((const alset_t*)bm->h_al)->n * 2 + 1, sizeof(bgt_allele_t));
    }

    /* What the device has to deliver per site.  Nothing but counts with -G.  Otherwise the finished genotype
     * vector of bgt_gen_gt (and its VCF text when the caller writes VCF): then bm->a[0] holds the merged vector
     * and bm->a[1] the merged text instead of the two byte planes.  The byte planes themselves are only needed
     * when a per-sample mask drops samples from the output (mgs > 1, ref bgt.c:300-311). */
    {
        int want = 0, shown = 0;
        for (i = 0; i < bm->n_out; ++i) shown += bm->mgs[i] <= 1;
        if (!(bm->flag & BGT_F_NO_GT)) want = shown == bm->n_out ? BGTH_WANT_GT8 : BGTH_WANT_PLANES;
        if (bm->h_al && (bm->flag & (BGT_F_CNT_AL | BGT_F_CNT_HAP))) want = BGTH_WANT_BITS;   /* -S / -H: the rows stay on the device */
        for (i = 0; i < bm->n_bgt; ++i) {
            devrd_t *dv = (devrd_t*)bm->bgt[i]->pb;
            dv->want = want | ((want & BGTH_WANT_GT8) && dv->text_mode ? BGTH_WANT_GTTEXT : 0);
            if (dv->rd) bgth_reader_config(dv->rd, dv->want, 0);
        }
    }
    return rc;
}

int bgtm_test_mgs(const bgtm_t *bm)                           /* ref bgt.c:678-688 */
{
    int i, cnt[BGT_MAX_GROUPS];
    memset(cnt, 0, sizeof(cnt));
    for (i = 0; i < bm->n_out; ++i) ++cnt[bm->group[i] - 1];
    for (i = 0; i < bm->n_out; ++i) if (bm->mgs[i] > cnt[bm->group[i] - 1]) return 0;
    return 1;
}

static char *group_key(char key[5], char nc, int g)           /* AN1..AN32 / AC1..AC32 (ref bgt.c:692-698) */
{
    key[0] = 'A'; key[1] = nc;
    if (g < 9) { key[2] = (char)('0' + g + 1); key[3] = 0; }
    else { key[2] = (char)('0' + (g + 1) / 10); key[3] = (char)('0' + (g + 1) % 10); key[4] = 0; }
    return key;
}

static void assign_counts(kexpr_t *e, const bgt_info_t *ss)   /* ref bgt.c:700-710 */
{
    int i;
    char key[5];
    ke_set_int(e, "AN", ss->an);
    ke_set_int(e, "AC", ss->ac[0]);
    for (i = 0; i < ss->n_groups; ++i) {
        ke_set_int(e, group_key(key, 'N', i), ss->gan[i]);
        ke_set_int(e, group_key(key, 'C', i), ss->gac[i][0]);
    }
}

static int pass_site_flt(const bgt_info_t *ss, kexpr_t *flt)  /* ref bgt.c:712-719 */
{
    int err, yes;
    if (flt == NULL) return 1;
    assign_counts(flt, ss);
    yes = !!ke_eval_int(flt, &err);
    return err ? 0 : yes;
}

/* one line of `-t` output: every field expression evaluated on the counts and on CHROM / POS / END / REF / ALT of
 * the site; an evaluation error prints `*` (ref bgt.c:759-795) */
static void gen_tbl_line(bgtm_t *bm, const bgt_info_t *ss, const bcf1_t *b, const char *ref, int l_ref,
                         const char *alt, int l_alt)
{
    int i;
    kstring_t *s = &bm->tbl_line;
    char *r = (char*)malloc((size_t)l_ref + 1), *a = (char*)malloc((size_t)l_alt + 1);
    memcpy(r, ref, (size_t)l_ref); r[l_ref] = 0;
    memcpy(a, alt, (size_t)l_alt); a[l_alt] = 0;
    s->l = 0;
    for (i = 0; i < bm->n_fields; ++i) {
        kexpr_t *e = bm->fields[i];
        int64_t vi; double vr; const char *vs; int type, err;
        if (i) ks_putc(s, '\t');
        assign_counts(e, ss);
        ke_set_str(e, "CHROM", bm->h_out->id[BCF_DT_CTG][b->rid].key);
        ke_set_int(e, "POS", b->pos + 1);
        ke_set_int(e, "END", b->pos + b->rlen);
        ke_set_str(e, "REF", r);
        ke_set_str(e, "ALT", a);
        err = ke_eval(e, &vi, &vr, &vs, &type);
        if (err) ks_putc(s, '*');
        else if (type == KEV_INT) ks_printf(s, "%ld", (long)vi);
        else if (type == KEV_REAL) ks_printf(s, "%lg", vr);
        else if (type == KEV_STR) ks_puts(s, vs);
    }
    ks_need(s, 1); s->s[s->l] = 0;
    free(r); free(a);
}

static void fill_info(const bcf_hdr_t *h, const bgt_info_t *ss, bcf1_t *b)   /* ref bgt.c:721-733 */
{
    bcf_append_info_ints(h, b, "AN", 1, &ss->an);
    bcf_append_info_ints(h, b, "AC", (int)b->n_allele - 1, ss->ac);
    if (ss->n_groups > 1) {
        int i;
        char key[5];
        for (i = 0; i < ss->n_groups; ++i) {
            bcf_append_info_ints(h, b, group_key(key, 'N', i), 1, &ss->gan[i]);
            bcf_append_info_ints(h, b, group_key(key, 'C', i), (int)b->n_allele - 1, ss->gac[i]);
        }
    }
}

/* one merged site.  Returns 0 = emitted, 1 = filtered out, -1 = no more sites (ref bgt.c:797-878), -2 = a database
 * could not deliver its row (device failure, row outside the .pbf): the caller must not take that for the end.
 * AC/AN of the merged site = sum over the databases that carry the site of the counts the device
 * reduced for that database's row (ref bgt.c:735-757 over the concatenated planes; a database
 * without the site contributes code 2 = missing, which adds to no count, ref :837-840,755-756). */
static int read_core(bgtm_t *bm, bcf1_t *b)
{
    int i, off = 0, n_rest = 0, max_allele = 0, best = -1, l_ref, al_ret = 0;
    uint8_t had[bm->n_bgt > 0 ? bm->n_bgt : 1];                /* which databases carry this site */
    const sitetab_t *bt = NULL;
    int64_t bs = -1;
    bgt_info_t ss;
    for (i = 0; i < bm->n_bgt; ++i) {
        if (bm->r[i].b0 == NULL && read_rec(bm->bgt[i], &bm->r[i]) < -1) return -2;   /* device / seek failure: an
                                                                * error, not the end of this database's sites */
        n_rest += bm->r[i].b0 != NULL;
        if (bm->r[i].b0) bm->n_gt_read += (uint64_t)bm->bgt[i]->n_out;
    }
    if (n_rest == 0) return -1;
    for (i = 0; i < bm->n_bgt; ++i) {                         /* the smallest look-ahead site */
        const sitetab_t *t = sites_of(bm->bgt[i]);
        const int64_t s = ((devrd_t*)bm->bgt[i]->pb)->site;
        if (bm->r[i].b0 == NULL) continue;
        if (best >= 0) {
            const int c = st_cmp(bt, bs, t, s);
            if (c > 0) { best = i; bt = t; bs = s; max_allele = t->n_allele[s]; }
            else if (c == 0 && t->n_allele[s] > max_allele) max_allele = t->n_allele[s];
        } else { best = i; bt = t; bs = s; max_allele = t->n_allele[s]; }
    }
    assert(best >= 0 && max_allele >= 2);
    bcf_set_site(b, bt->rid[bs], bt->pos[bs], bt->rlen[bs], bt->pool + bt->ref_off[bs], bt->ref_len[bs],
                 bt->pool + bt->alt_off[bs], bt->alt_len[bs], max_allele > 2 ? "<M>" : NULL);
    l_ref = bt->ref_len[bs];
    if (l_ref != b->rlen) { int32_t val = b->pos + b->rlen; bcf_append_info_ints(bm->h_out, b, "END", 1, &val); }

    if (bm->h_al) {                                           /* ref bgt.c:843-848 */
        al_ret = al_present((const alset_t*)bm->h_al, bm->h_out->id[BCF_DT_CTG][b->rid].key, b->rid, b->pos, b->rlen,
                            bt->pool + bt->ref_off[bs], bt->ref_len[bs], bt->pool + bt->alt_off[bs], bt->alt_len[bs]);
    }
    memset(&ss, 0, sizeof(ss));
    ss.n_groups = bm->n_groups;
    for (i = 0; i < bm->n_bgt; ++i) {                         /* consume the databases that have this site */
        bgt_t *bgt = bm->bgt[i];
        const sitetab_t *t = sites_of(bgt);
        devrd_t *dv = (devrd_t*)bgt->pb;
        had[i] = 0;
        if (bgt->n_out == 0) continue;
        if (bm->r[i].b0 && st_cmp(bt, bs, t, dv->site) == 0) {
            const int32_t *c = dv->counts;
            int g;
            had[i] = 1;
            bm->r[i].b0 = NULL;
            if (bm->r[i].a[0]) {
                memcpy(bm->a[0] + off, bm->r[i].a[0], (size_t)bgt->n_out << 1);
                memcpy(bm->a[1] + off, bm->r[i].a[1], (size_t)bgt->n_out << 1);
            }
            if (dv->gt8) memcpy(bm->a[0] + off, dv->gt8, (size_t)bgt->n_out << 1);
            if (dv->gttext) memcpy(bm->a[1] + 2 * (size_t)off, dv->gttext, (size_t)bgt->n_out << 2);
            ss.an += c[0]; ss.ac[0] += c[1]; ss.ac[1] += c[2];
            if (bm->n_groups > 1)
                for (g = 0; g < bm->n_groups; ++g) {
                    ss.gan[g] += c[3 * (1 + g)]; ss.gac[g][0] += c[3 * (1 + g) + 1]; ss.gac[g][1] += c[3 * (1 + g) + 2];
                }
        } else if (dv->want & BGTH_WANT_PLANES) {             /* this database lacks the site: all missing */
            memset(bm->a[0] + off, 0, (size_t)bgt->n_out << 1);
            memset(bm->a[1] + off, 1, (size_t)bgt->n_out << 1);
        } else if (dv->want & BGTH_WANT_GT8) {                /* the same in vector / text form: 0 bytes, "./." */
            memset(bm->a[0] + off, 0, (size_t)bgt->n_out << 1);
            if (dv->want & BGTH_WANT_GTTEXT) {
                int k;
                for (k = 0; k < bgt->n_out; ++k) memcpy(bm->a[1] + 2 * (size_t)off + 4 * (size_t)k, "\t./.", 4);
            }
        }
        off += bgt->n_out << 1;
    }
    if (bm->h_al && al_ret == 0) return 1;                    /* not an allele of the set */
    if ((bm->flag & BGT_F_SET_AC) || bm->site_flt || bm->n_fields > 0 || bm->n_groups > 1) {
        fill_info(bm->h_out, &ss, b);
        if (bm->n_fields > 0)
            gen_tbl_line(bm, &ss, b, bt->pool + bt->ref_off[bs], bt->ref_len[bs], bt->pool + bt->alt_off[bs], bt->alt_len[bs]);
        if (!pass_site_flt(&ss, bm->site_flt)) return 1;
    }
    if (bm->h_al) {                                           /* ref bgt.c:859-876 */
        /* -S: +1 for every sample that carries the allele (a reference-allele query counts code 0); -H: bit n_aal of a
         * haplotype = it carries this allele.  Both reductions run on the device over the row it just decoded, one fold
         * per database that has the site (a database without it is all code 2 and adds nothing to either). */
        const int do_al = (bm->flag & BGT_F_CNT_AL) && bm->alcnt, do_hap = (bm->flag & BGT_F_CNT_HAP) && bm->hap;
        if (do_al || do_hap)
            for (i = 0; i < bm->n_bgt; ++i) {
                devrd_t *dv = (devrd_t*)bm->bgt[i]->pb;
                if (!had[i] || dv->rd == NULL) continue;
                if (bgth_reader_fold_last(dv->rd, do_al ? (al_ret == 2 ? 0 : 1) : -1, do_hap ? (bm->n_aal & 63) : -1) < 0) {
                    fprintf(stderr, "[E::%s] %s\n", __func__, bgth_last_error());
                    return -2;
                }
                dv->folded = 1;
            }
        al_from_site(bm->h_out->id[BCF_DT_CTG][b->rid].key, b->rid, b->pos, b->rlen, bt->pool + bt->ref_off[bs], bt->ref_len[bs],
                     bt->pool + bt->alt_off[bs], bt->alt_len[bs], &bm->aal[bm->n_aal++], NULL);
    }
    return 0;
}

/* After bgtm_read the reference leaves the merged site's two byte planes in bm->a (bgt.c:829-842: a[0][j] = low bit,
 * a[1][j] = high bit of haplotype j's 2-bit code; a database without the site contributes code 2).  The device hands this
 * reader the finished GT vector instead (bytes 2, 4, 0, 6 for codes 0..3, bgt.c:250), which read_core merged into
 * bm->a[0]: once the record has taken its copy, the vector is turned back into the planes IN PLACE, so a caller that
 * reads bm->a after bgtm_read finds what bgt.h:70 promises (two compares per byte; auto-vectorised). */
static void planes_from_gt8(bgtm_t *bm)
{
    uint8_t *restrict lo = bm->a[0], *restrict hi = bm->a[1];
    const size_t n = (size_t)bm->n_out << 1;
    size_t j;
    for (j = 0; j < n; ++j) {
        const uint8_t v = lo[j];
        hi[j] = (uint8_t)((v == 0) | (v == 6));
        lo[j] = (uint8_t)((v == 4) | (v == 6));
    }
}

int bgtm_read(bgtm_t *bm, bcf1_t *b)                          /* ref bgt.c:880-888 */
{
    int ret;
    if (bm->h_out == NULL && bgtm_prepare(bm) < 0) return -2;
    while ((ret = read_core(bm, b)) > 0) {}
    if (ret == -1 && bm->h_al && sync_folds(bm) < 0) ret = -2;  /* the end: bm->alcnt / bm->hap are complete for callers that read them */
    if (ret >= 0 && (bm->flag & BGT_F_NO_GT) == 0) {
        if (bm->n_bgt > 0 && (((devrd_t*)bm->bgt[0]->pb)->want & BGTH_WANT_GT8)) {
            gen_gt8(bm->h_out, b, bm->n_out, bm->a[0]);
            planes_from_gt8(bm);                              /* bm->a as bgt.h:70 promises it to the caller */
        }
        else gen_gt(bm->h_out, b, bm->n_out, (const uint8_t *const*)bm->a, bm->mgs);
    }
    return ret;
}

/* Extension (not in the reference): the next site as one VCF text line without the newline -- exactly what
 * vcf_format1(bm->h_out, b, s) gives after bgtm_read(bm, b) -- with the genotype columns taken from the text
 * the device formatted (b then carries the site and INFO only). */
void bgtm_want_vcf_text(bgtm_t *bm)      /* call before bgtm_prepare: the caller will use bgtm_read_vcf */
{
    int i;
    for (i = 0; i < bm->n_bgt; ++i) ((devrd_t*)bm->bgt[i]->pb)->text_mode = 1;
}

int bgtm_read_vcf(bgtm_t *bm, bcf1_t *b, kstring_t *s)
{
    int ret;
    if (bm->h_out == NULL) { bgtm_want_vcf_text(bm); if (bgtm_prepare(bm) < 0) return -2; }
    if (bm->n_bgt == 0 || !(((devrd_t*)bm->bgt[0]->pb)->want & BGTH_WANT_GTTEXT) || bm->n_out == 0) {
        if ((ret = bgtm_read(bm, b)) >= 0) vcf_format1(bm->h_out, b, s);
        return ret;
    }
    while ((ret = read_core(bm, b)) > 0) {}
    if (ret < 0) return ret;
    b->n_fmt = 0; b->n_sample = 0; b->indiv.l = 0;
    vcf_format1(bm->h_out, b, s);
    ks_puts(s, "\tGT");
    ks_putn(s, (const char*)bm->a[1], (size_t)bm->n_out << 2);
    return ret;
}

/* ------------------------------------------------------------------------------------------------
 * Extension: the whole walk of `bgt view -G [-C] [-f EXPR] [-s ..] prefix [prefix2 ...]` in bulk.
 * The site-by-site contract of bgtm_read (one call, one site) costs ~0.4 us of host work per site -- record
 * assembly, INFO, the filter expression, text formatting -- on ONE thread, 30 times the device's time for the same
 * sites.  When nothing in the query needs that contract (no genotype columns, VCF text, the whole files or a start
 * offset) the same per-site functions run here over the merged site walk on several threads: one device scan per
 * database (side by side) delivers the counts of every row, the sites are cut into blocks, every thread formats blocks into buffers of its
 * own with its own copy of the filter expression, and the blocks are written in order.  Byte-identical to the
 * bgtm_read_vcf loop.  Returns the number of records written, or -1 if the query needs the site-by-site path.
 * ------------------------------------------------------------------------------------------------ */
#define BULK_MAX_DB 64
#define BULK_MAX_THREADS 128
/* one database's device pass, in pieces: `ready` rows (from r0 on) have their counts on the host; formatters start on a
 * block of sites as soon as the rows it uses are there, while the device scans the next piece */
typedef struct bulk_scan_s {
    bgth_reader_t *rd; int64_t r0, r1, piece; int32_t *counts; int cstride; int rc; char err[256]; pthread_t th; int started;
    volatile int64_t ready;
    volatile int *failed;                                         /* set (under the lock) when a piece fails: releases every waiter */
    pthread_mutex_t *lock; pthread_cond_t *cond;
} bulk_scan_t;

typedef struct {
    bgtm_t *bm; int n_db; int need_counts, cstride;
    bulk_scan_t *scan; volatile int failed;
    const sitetab_t *t[BULK_MAX_DB]; int64_t row_min[BULK_MAX_DB]; const int32_t *counts[BULK_MAX_DB];
    /* the merged walk: site j of the output is site idx[d][j] of database d (-1: that database lacks it); lead[j] = the
     * database whose record describes the site (the smallest look-ahead, first of equals: read_core's `best`) */
    int64_t n_sites; int32_t *idx[BULK_MAX_DB]; uint8_t *lead;   /* one database: both NULL = its own order from site lo0 on */
    int64_t lo0;
    int64_t n_blocks, blk_sites;
    kstring_t *out; int64_t *n_lines; volatile int *done;
    int64_t next_block;
    double wall_ms, cpu_ms;                                       /* BGT_TRACE: formatting time over all blocks, wall and thread CPU */
    int via_record;
    pthread_mutex_t lock; pthread_cond_t cond;
} bulk_t;

#define BULK_IDX(k, d, j) ((k)->idx[d] ? (int64_t)(k)->idx[d][j] : (k)->lo0 + (j))
#define BULK_LEAD(k, j)   ((k)->lead ? (k)->lead[j] : 0)

/* Buffers that outlive a walk.  A formatted block is ~200 KB and the counts of a database are 12 bytes per row: fresh
 * allocations of that size are mmap'ed and page-faulted by 64 threads at once, all through one address-space lock (measured
 * at C2 scale in a resident host: a block of 4,096 sites took 8 ms to format instead of 1.8, and the device pass's copy
 * to the host stalled behind the faults for up to 50 ms).  So blocks are formatted into buffers taken from -- and returned
 * to -- a process-wide pool, and the counts buffer of the last walk is kept for the next one. */
#define BULK_POOL_MAX 256
static struct { pthread_mutex_t lock; char *s[BULK_POOL_MAX]; size_t m[BULK_POOL_MAX]; int n; int32_t *counts; size_t counts_bytes; } g_bulk_pool =
    { PTHREAD_MUTEX_INITIALIZER, {0}, {0}, 0, NULL, 0 };
static void bulk_buf_get(kstring_t *o, size_t want)
{
    o->l = 0; o->m = 0; o->s = NULL;
    pthread_mutex_lock(&g_bulk_pool.lock);
    if (g_bulk_pool.n > 0) { --g_bulk_pool.n; o->s = g_bulk_pool.s[g_bulk_pool.n]; o->m = g_bulk_pool.m[g_bulk_pool.n]; }
    pthread_mutex_unlock(&g_bulk_pool.lock);
    if (o->s == NULL) { o->s = (char*)malloc(want); o->m = o->s ? want : 0; }
}
static void bulk_buf_put(kstring_t *o)
{
    if (o->s == NULL) return;
    pthread_mutex_lock(&g_bulk_pool.lock);
    if (g_bulk_pool.n < BULK_POOL_MAX && o->m <= (1u << 20)) { g_bulk_pool.s[g_bulk_pool.n] = o->s; g_bulk_pool.m[g_bulk_pool.n] = o->m; ++g_bulk_pool.n; o->s = NULL; }
    pthread_mutex_unlock(&g_bulk_pool.lock);
    free(o->s);
    o->s = NULL; o->l = o->m = 0;
}
static int32_t *bulk_counts_get(size_t bytes)
{
    int32_t *p = NULL;
    pthread_mutex_lock(&g_bulk_pool.lock);
    if (g_bulk_pool.counts && g_bulk_pool.counts_bytes >= bytes) { p = g_bulk_pool.counts; g_bulk_pool.counts = NULL; }
    pthread_mutex_unlock(&g_bulk_pool.lock);
    return p ? p : (int32_t*)malloc(bytes);
}
static void bulk_counts_put(int32_t *p, size_t bytes)
{
    int32_t *old = NULL;
    if (p == NULL) return;
    pthread_mutex_lock(&g_bulk_pool.lock);
    if (bytes <= ((size_t)1 << 30) && (g_bulk_pool.counts == NULL || g_bulk_pool.counts_bytes < bytes)) {
        old = g_bulk_pool.counts; g_bulk_pool.counts = p; g_bulk_pool.counts_bytes = bytes; p = NULL;
    }
    pthread_mutex_unlock(&g_bulk_pool.lock);
    free(old); free(p);
}

static inline char *put_dec(char *p, int64_t v)                 /* decimal digits of v at p; returns the end */
{
    char tmp[24];
    int n = 0;
    uint64_t u = v < 0 ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
    if (v < 0) *p++ = '-';
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    while (n) *p++ = tmp[--n];
    return p;
}

static void *bulk_worker(void *arg)
{
    bulk_t *k = (bulk_t*)arg;
    bgtm_t *bm = k->bm;
    int last_rid = -1;
    const char *chrom = NULL;
    size_t l_chrom = 0;
    kexpr_t *flt = ke_clone(bm->site_flt);
    bcf1_t *b = bcf_init1();
    kstring_t line = {0, 0, 0};
    for (;;) {
        int64_t blk, j, j0, j1, n = 0;
        kstring_t *o;
        pthread_mutex_lock(&k->lock);
        blk = k->next_block++;
        pthread_mutex_unlock(&k->lock);
        if (blk >= k->n_blocks) break;
        o = &k->out[blk];
        bulk_buf_get(o, (size_t)k->blk_sites * 64);
        j0 = blk * k->blk_sites; j1 = j0 + k->blk_sites < k->n_sites ? j0 + k->blk_sites : k->n_sites;
        if (k->need_counts) {                                     /* wait for the pieces that hold this block's rows */
            int d;
            for (d = 0; d < k->n_db; ++d) {
                int64_t need = -1;
                if (k->scan[d].rd == NULL) continue;
                for (j = j0; j < j1; ++j) if (BULK_IDX(k, d, j) >= 0 && k->t[d]->row[BULK_IDX(k, d, j)] > need) need = k->t[d]->row[BULK_IDX(k, d, j)];
                if (need < 0) continue;
                pthread_mutex_lock(&k->lock);
                while (!k->failed && k->scan[d].ready <= need - k->scan[d].r0) pthread_cond_wait(&k->cond, &k->lock);
                pthread_mutex_unlock(&k->lock);
            }
        }
        if (k->failed) j1 = j0;                                   /* a device pass failed: nothing more is formatted */
        const double w0 = rd_now_ms();
        struct timespec c0; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &c0);
        for (j = j0; j < j1; ++j) {                               /* what read_core does for one merged site without genotypes */
            const sitetab_t *t = k->t[BULK_LEAD(k, j)];
            const int64_t i = BULK_IDX(k, BULK_LEAD(k, j), j);
            int d, max_allele = 0;
            for (d = 0; d < k->n_db; ++d)
                if (BULK_IDX(k, d, j) >= 0 && k->t[d]->n_allele[BULK_IDX(k, d, j)] > max_allele) max_allele = k->t[d]->n_allele[BULK_IDX(k, d, j)];
            bgt_info_t ss;
            if (k->need_counts) {
                int g;
                memset(&ss, 0, sizeof(ss));
                ss.n_groups = bm->n_groups;
                for (d = 0; d < k->n_db; ++d) {                   /* the databases that carry the site add their counts */
                    const int32_t *c;
                    if (BULK_IDX(k, d, j) < 0) continue;
                    c = k->counts[d] + (size_t)(k->t[d]->row[BULK_IDX(k, d, j)] - k->row_min[d]) * (size_t)k->cstride;
                    ss.an += c[0]; ss.ac[0] += c[1]; ss.ac[1] += c[2];
                    if (bm->n_groups > 1)
                        for (g = 0; g < bm->n_groups; ++g) { ss.gan[g] += c[3 * (1 + g)]; ss.gac[g][0] += c[3 * (1 + g) + 1]; ss.gac[g][1] += c[3 * (1 + g) + 2]; }
                }
                if (!pass_site_flt(&ss, flt)) continue;
            }
            if (k->via_record) {                                  /* BGT_BULK_VIA_RECORD=1: through a BCF record, as bgtm_read_vcf does */
                bcf_set_site(b, t->rid[i], t->pos[i], t->rlen[i], t->pool + t->ref_off[i], (int)t->ref_len[i],
                             t->pool + t->alt_off[i], (int)t->alt_len[i], max_allele > 2 ? "<M>" : NULL);
                if ((int)t->ref_len[i] != b->rlen) { int32_t val = b->pos + b->rlen; bcf_append_info_ints(bm->h_out, b, "END", 1, &val); }
                if (k->need_counts) fill_info(bm->h_out, &ss, b);
                vcf_format1(bm->h_out, b, &line);
                ks_putn(o, line.s, line.l); ks_putc(o, '\n');
            } else {
                /* The same bytes written directly (vcf_format1 over the record bcf_set_site / fill_info would build: ID and FILTER
                 * are '.', QUAL is 0, INFO = [END] AN AC [AN<g> AC<g> ...] with one AC value per ALT allele): encoding a record and
                 * decoding it again cost 0.5 us of CPU per site, the bound of a resident query (16 formatter cores: 32 of 44 ms). */
                const int three = max_allele > 2, l_ref = (int)t->ref_len[i], l_alt = (int)t->alt_len[i];
                char *q;
                if (t->rid[i] != last_rid) { last_rid = t->rid[i]; chrom = bm->h_out->id[BCF_DT_CTG][last_rid].key; l_chrom = strlen(chrom); }
                ks_need(o, l_chrom + (size_t)l_ref + (size_t)l_alt + 128 + (k->need_counts && bm->n_groups > 1 ? (size_t)bm->n_groups * 48 : 0));
                q = o->s + o->l;
                memcpy(q, chrom, l_chrom); q += l_chrom;
                *q++ = '\t'; q = put_dec(q, (int64_t)t->pos[i] + 1);
                *q++ = '\t'; *q++ = '.'; *q++ = '\t';
                memcpy(q, t->pool + t->ref_off[i], (size_t)l_ref); q += l_ref;
                *q++ = '\t';
                memcpy(q, t->pool + t->alt_off[i], (size_t)l_alt); q += l_alt;
                if (three) { memcpy(q, ",<M>", 4); q += 4; }
                memcpy(q, "\t0\t.\t", 5); q += 5;                    /* (QUAL of a record bcf_set_site makes is 0, not missing) */
                if ((int)t->ref_len[i] != t->rlen[i]) {
                    memcpy(q, "END=", 4); q = put_dec(q + 4, (int64_t)t->pos[i] + t->rlen[i]);
                    if (k->need_counts) *q++ = ';';
                } else if (!k->need_counts) *q++ = '.';
                if (k->need_counts) {
                    int g;
                    memcpy(q, "AN=", 3); q = put_dec(q + 3, ss.an);
                    memcpy(q, ";AC=", 4); q = put_dec(q + 4, ss.ac[0]);
                    if (three) { *q++ = ','; q = put_dec(q, ss.ac[1]); }
                    if (bm->n_groups > 1)
                        for (g = 0; g < bm->n_groups; ++g) {
                            char key[5];
                            size_t lk;
                            *q++ = ';'; group_key(key, 'N', g); lk = strlen(key); memcpy(q, key, lk); q += lk; *q++ = '='; q = put_dec(q, ss.gan[g]);
                            *q++ = ';'; group_key(key, 'C', g); memcpy(q, key, lk); q += lk; *q++ = '='; q = put_dec(q, ss.gac[g][0]);
                            if (three) { *q++ = ','; q = put_dec(q, ss.gac[g][1]); }
                        }
                }
                *q++ = '\n';
                o->l = (size_t)(q - o->s);
            }
            ++n;
        }
        struct timespec c1; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &c1);
        const double w1 = rd_now_ms();
        pthread_mutex_lock(&k->lock);
        k->wall_ms += w1 - w0; k->cpu_ms += (c1.tv_sec - c0.tv_sec) * 1e3 + (c1.tv_nsec - c0.tv_nsec) * 1e-6;
        k->n_lines[blk] = n; k->done[blk] = 1;
        pthread_cond_broadcast(&k->cond);
        pthread_mutex_unlock(&k->lock);
    }
    free(line.s);
    bcf_destroy1(b);
    ke_destroy(flt);
    return NULL;
}

/* BGT_TRACE=1: the timeline of a bulk walk, ms since its start (tuning aid) */
static double g_bulk_t0;
static void bulk_mark(const char *what, long n)
{
    if (getenv("BGT_TRACE")) fprintf(stderr, "[bgt trace]   bulk +%7.2f ms  %s %ld\n", rd_now_ms() - g_bulk_t0, what, n);
}

static void *bulk_scan_worker(void *arg)
{
    bulk_scan_t *q = (bulk_scan_t*)arg;
    int64_t a0;
    q->rc = 0;
    for (a0 = q->r0; a0 < q->r1 && q->rc == 0; ) {              /* every row's AN / AC, a piece of whole sub-blocks at a time */
        int64_t a1 = a0 + q->piece < q->r1 ? (a0 + q->piece) / q->piece * q->piece : q->r1;
        if (a1 <= a0 || a1 > q->r1) a1 = q->r1;
        if (bgth_reader_scan(q->rd, a0, a1, q->counts + (size_t)(a0 - q->r0) * (size_t)q->cstride, NULL) < 0) {
            q->rc = -1;
            strncpy(q->err, bgth_last_error(), sizeof(q->err) - 1); q->err[sizeof(q->err) - 1] = 0;
        }
        bulk_mark("device piece done, rows", (long)(a1 - q->r0));
        pthread_mutex_lock(q->lock);
        if (q->rc == 0) q->ready = a1 - q->r0; else *q->failed = 1;
        pthread_cond_broadcast(q->cond);
        pthread_mutex_unlock(q->lock);
        a0 = a1;
    }
    return NULL;
}

long bgtm_write_vcf_bulk(bgtm_t *bm, FILE *fp, long n_rec)
{
    bulk_t k;
    bulk_scan_t scan[BULK_MAX_DB];
    pthread_t th[BULK_MAX_THREADS];
    int n_started = 0;
    int64_t i, lo[BULK_MAX_DB], hi[BULK_MAX_DB], cur[BULK_MAX_DB], total = 0, cap = 0;
    long written = 0;
    int n_threads, j, d, failed = 0;
    g_bulk_t0 = rd_now_ms();
    if (bm->h_out == NULL && bgtm_prepare(bm) < 0) return -2;
    if (bm->n_bgt < 1 || bm->n_bgt > BULK_MAX_DB || !(bm->flag & BGT_F_NO_GT) || (bm->flag & (BGT_F_CNT_AL | BGT_F_CNT_HAP)) || bm->h_al || bm->n_fields > 0) return -1;
    memset(&k, 0, sizeof(k));
    k.bm = bm; k.n_db = bm->n_bgt;
    k.via_record = getenv("BGT_BULK_VIA_RECORD") != NULL;
    k.need_counts = (bm->flag & BGT_F_SET_AC) || bm->site_flt || bm->n_groups > 1;
    k.cstride = 3 * (1 + (bm->n_groups > 1 ? bm->n_groups : 0));
    for (d = 0; d < k.n_db; ++d) {                                /* every database: a plain walk from its cursor to its end */
        const bgt_t *bgt = bm->bgt[d];
        const devrd_t *dv = (const devrd_t*)bgt->pb;
        if (bgt->bed || bgt->h_al || bgt->itr || dv->own_sites || bgt->n_out == 0 || bm->r[d].b0) return -1;
        if (k.need_counts && dv->rd == NULL) return -1;
        k.t[d] = sites_of(bgt);
        lo[d] = ((cursor_t*)bgt->bcf)->next; hi[d] = k.t[d]->n;
        if (lo[d] < 0) lo[d] = 0;
        if (lo[d] > hi[d]) lo[d] = hi[d];
        total += hi[d] - lo[d];
    }
    if (total == 0) return 0;
    if (n_rec < total) return -1;                                /* -n counts EMITTED records: site by site */
    /* the merged order, exactly as read_core finds it: the smallest look-ahead site, every database at that site consumed.
     * (Built AFTER the device passes below have been started: it is ~5 ns x sites of one thread's time the device does not
     * have to wait for; one database = its own order.) */
    /* counts: one device pass per database over the rows its sites use, the databases side by side and each in pieces, so
     * that the formatters below work on the sites of piece k while the device scans piece k+1 */
    memset(scan, 0, sizeof(scan));
    k.scan = scan;
    pthread_mutex_init(&k.lock, NULL); pthread_cond_init(&k.cond, NULL);
    if (k.need_counts) {
        for (d = 0; d < k.n_db; ++d) {
            const sitetab_t *t = k.t[d];
            int64_t row_min = INT64_MAX, row_max = -1;
            for (i = lo[d]; i < hi[d]; ++i) { if (t->row[i] < row_min) row_min = t->row[i]; if (t->row[i] > row_max) row_max = t->row[i]; }
            if (row_max < 0) { k.row_min[d] = 0; continue; }
            k.row_min[d] = row_min;
            scan[d].rd = ((devrd_t*)bm->bgt[d]->pb)->rd; scan[d].r0 = row_min; scan[d].r1 = row_max + 1;
            /* pieces of 256 decoding units (sub-blocks of 2048 rows, or file blocks of an image without sub-checkpoints): a
             * piece must fill the chip by itself -- 64 units were a quarter of it, four times the device time.  A short image
             * has finer units (bgt_hip.cpp fit_sub_shift): at least 262,144 rows then, several rounds of workgroups */
            {
                const devrd_t *dvd = (const devrd_t*)bm->bgt[d]->pb;
                const bgth_pbf_t *img = dvd->own_img ? dvd->own_img : (const bgth_pbf_t*)bm->bgt[d]->f->gpu;
                scan[d].piece = img ? 256 * bgth_pbf_unit_rows(img) : 524288;
                if (scan[d].piece < 262144) scan[d].piece = 262144;
            }
            scan[d].cstride = k.cstride; scan[d].lock = &k.lock; scan[d].cond = &k.cond; scan[d].failed = &k.failed;
            scan[d].counts = k.n_db == 1 ? bulk_counts_get((size_t)(row_max - row_min + 1) * (size_t)k.cstride * 4)
                                         : (int32_t*)malloc((size_t)(row_max - row_min + 1) * (size_t)k.cstride * 4);
            if (scan[d].counts == NULL) { failed = 1; break; }
            k.counts[d] = scan[d].counts;
        }
        if (failed) {
            for (d = 0; d < k.n_db; ++d) { free(scan[d].counts); free(k.idx[d]); }
            free(k.lead);
            pthread_mutex_destroy(&k.lock); pthread_cond_destroy(&k.cond);
            return -1;
        }
        for (d = 0; d < k.n_db; ++d)
            if (scan[d].rd && pthread_create(&scan[d].th, NULL, bulk_scan_worker, &scan[d]) == 0) scan[d].started = 1;
    }
    bulk_mark("device passes started, databases", (long)k.n_db);
    if (k.n_db == 1) {                                            /* one database: site j of the output is its site lo + j */
        k.idx[0] = NULL; k.lead = NULL; k.lo0 = lo[0];
        k.n_sites = total;
    } else {
        for (d = 0; d < k.n_db; ++d) { cur[d] = lo[d]; k.idx[d] = NULL; }
        cap = k.n_db == 1 ? total : total / 2 + 1024;
        for (d = 0; d < k.n_db; ++d) k.idx[d] = (int32_t*)malloc((size_t)cap * 4);
        k.lead = (uint8_t*)malloc((size_t)cap);
        for (;;) {
            int best = -1;
            for (d = 0; d < k.n_db; ++d) {
                if (cur[d] >= hi[d]) continue;
                if (best < 0 || st_cmp(k.t[best], cur[best], k.t[d], cur[d]) > 0) best = d;
            }
            if (best < 0) break;
            if (k.n_sites == cap) {
                cap += cap / 2 + 1024;
                for (d = 0; d < k.n_db; ++d) k.idx[d] = (int32_t*)realloc(k.idx[d], (size_t)cap * 4);
                k.lead = (uint8_t*)realloc(k.lead, (size_t)cap);
            }
            k.lead[k.n_sites] = (uint8_t)best;
            {
                const int64_t at = cur[best];                         /* (compare the others with the chosen head before it moves) */
                for (d = 0; d < k.n_db; ++d) {
                    if (d != best && cur[d] < hi[d] && st_cmp(k.t[best], at, k.t[d], cur[d]) == 0) k.idx[d][k.n_sites] = (int32_t)cur[d]++;
                    else if (d != best) k.idx[d][k.n_sites] = -1;
                }
                k.idx[best][k.n_sites] = (int32_t)cur[best]++;
            }
            ++k.n_sites;
        }
    }
    bulk_mark("merged order built, sites", (long)k.n_sites);
    k.blk_sites = 4096;
    k.n_blocks = (k.n_sites + k.blk_sites - 1) / k.blk_sites;
    k.out = (kstring_t*)calloc((size_t)k.n_blocks, sizeof(kstring_t));
    k.n_lines = (int64_t*)calloc((size_t)k.n_blocks, 8);
    k.done = (volatile int*)calloc((size_t)k.n_blocks, sizeof(int));
    {
        long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
        const char *e = getenv("BGT_THREADS");
        n_threads = e ? atoi(e) : (int)(ncpu > 64 ? 64 : ncpu);   /* (formatting is memory-bound long before 64 threads) */
        if (n_threads < 1) n_threads = 1;
        if (n_threads > BULK_MAX_THREADS) n_threads = BULK_MAX_THREADS;
        if (n_threads > k.n_blocks) n_threads = (int)k.n_blocks;
    }
    for (j = 0; j < n_threads; ++j) if (pthread_create(&th[n_started], NULL, bulk_worker, &k) == 0) ++n_started;
    for (d = 0; d < k.n_db; ++d) if (scan[d].rd && !scan[d].started) {   /* no thread for it: the device pass runs here */
        bulk_scan_worker(&scan[d]);
    }
    bulk_mark("formatter threads started", (long)n_started);
    {   /* tens of MB through a pipe: 64 KB of buffer is a context switch per 64 KB; ask for the 1 MB an unprivileged process may have */
        struct stat st;
        const int fd = fileno(fp);
        if (fd >= 0 && total > 65536 && fstat(fd, &st) == 0 && S_ISFIFO(st.st_mode)) (void)fcntl(fd, F_SETPIPE_SZ, 1 << 20);
    }
    if (n_started == 0) bulk_worker(&k);                          /* no formatter thread could be started: format here */
    for (i = 0; i < k.n_blocks; ++i) {                            /* blocks leave in order, as soon as they are ready */
        pthread_mutex_lock(&k.lock);
        while (!k.done[i]) pthread_cond_wait(&k.cond, &k.lock);
        pthread_mutex_unlock(&k.lock);
        if (i == 0 || i == k.n_blocks / 2) bulk_mark("block ready to leave", (long)i);
        if (k.out[i].l && !k.failed) { fwrite(k.out[i].s, 1, k.out[i].l, fp); written += (long)k.n_lines[i]; }   /* (only what left) */
        bulk_buf_put(&k.out[i]);
    }
    bulk_mark("last block written, lines", written);
    bulk_mark("formatting, sum over blocks: wall us", (long)(k.wall_ms * 1e3));
    bulk_mark("formatting, sum over blocks: thread CPU us", (long)(k.cpu_ms * 1e3));
    for (j = 0; j < n_started; ++j) pthread_join(th[j], NULL);
    for (d = 0; d < k.n_db; ++d) {
        if (scan[d].started) pthread_join(scan[d].th, NULL);
        if (scan[d].rd && scan[d].rc < 0) { fprintf(stderr, "[E::%s] %s\n", __func__, scan[d].err); failed = 2; }
    }
    pthread_mutex_destroy(&k.lock); pthread_cond_destroy(&k.cond);
    for (d = 0; d < k.n_db; ++d) {
        bm->n_gt_read += (uint64_t)(hi[d] - lo[d]) * (uint64_t)bm->bgt[d]->n_out;
        ((cursor_t*)bm->bgt[d]->bcf)->next = hi[d];
        if (k.n_db == 1 && scan[d].counts) bulk_counts_put(scan[d].counts, (size_t)(scan[d].r1 - scan[d].r0) * (size_t)k.cstride * 4);
        else free(scan[d].counts);
        free(k.idx[d]);
    }
    free(k.lead);
    free(k.out); free(k.n_lines); free((void*)k.done);
    return failed ? -2 : written;
}
