/* import_cli.c -- `bgt import`: VCF / BCF with genotypes -> prefix.pbf + prefix.bcf (+ .csi) + prefix.spl
 * (what the reference does in import.c:8-120 over atomic.c:15-220).
 *
 * Three stages, restated from the reference's behaviour (not its code):
 *   1. records: CHROM POS REF ALT FILTER, INFO/END and INFO/CIGAR, FORMAT/GT of every (diploid) sample
 *      (reference vcf.c:539-797 for text, the BCF2 typed layout for binary input)
 *   2. atomizer (atomic.c): every ALT allele is cut along its CIGAR against REF (given in INFO/CIGAR, else "nM" for
 *      equal lengths and 1M + insertion / deletion + rest for indels) into atomic alleles -- SNPs, one-base-anchored
 *      insertions and deletions; symbolic ALTs stay whole.  Per atom the genotype of a haplotype is 0 REF, 1 this atom,
 *      2 missing, 3 another allele overlapping it (<M>).  Atoms are emitted in (contig, position, length, ALT) order;
 *      an atom found again in a later record while still buffered keeps its first genotypes (as the reference does).
 *   3. writers: the genotype rows go to the DEVICE encoder (bgth_encoder_write, pbf_encoder.hip: the image is byte for
 *      byte the reference writer's), the sites to a site-only BCF with INFO/_row and its CSI index, the names to .spl.
 * There is no CPU PBWT encoder in this build: without a HIP device import fails. */
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>
#include <pthread.h>
#include <stdarg.h>
#include "../../include/bgt_reader.h"
#include "../../include/bgt_hip.h"
#include "csi_writer.h"

/* ---------------------------------------------------------------------------------------------------------------
 * one input record, reduced to what the atomizer needs
 * --------------------------------------------------------------------------------------------------------------- */
typedef struct {
    int rid, pos, rlen, n_allele, n_sample, filtered;
    char **allele; int m_allele;             /* NUL-terminated copies */
    kstring_t cigar;                          /* INFO/CIGAR (all alleles, comma separated) or empty */
    int8_t *gt; size_t m_gt;                  /* [2 * n_sample] allele number, -1 = missing */
    kstring_t pool;
} rec_t;

typedef struct prefetch_s prefetch_t;
static void pf_destroy(prefetch_t *pf);

typedef struct {
    int is_bcf, keep_flt;
    gzFile tf; bgzr_t *bf;
    bcf_hdr_t *h;
    kstring_t line;
    bcf1_t *b;
    int id_gt, id_cigar, id_end;
    prefetch_t *pf;                           /* text input: lines parsed ahead on several threads */
} reader_t;

static int gz_getline(gzFile f, kstring_t *s)
{
    char buf[65536];
    s->l = 0;
    for (;;) {
        size_t n;
        if (gzgets(f, buf, sizeof(buf)) == NULL) return s->l ? 0 : -1;
        n = strlen(buf);
        ks_putn(s, buf, n);
        if (n && buf[n - 1] == '\n') break;
    }
    while (s->l && (s->s[s->l - 1] == '\n' || s->s[s->l - 1] == '\r')) s->s[--s->l] = 0;
    return 0;
}

/* header of a text VCF (reference vcf.c:366-411): meta lines, optional contig lines from `-t FILE`, the #CHROM line */
static bcf_hdr_t *read_text_header(gzFile f, const char *fn_ref, kstring_t *line)
{
    kstring_t txt = {0, 0, 0};
    bcf_hdr_t *h;
    while (gz_getline(f, line) >= 0) {
        if (line->l == 0) continue;
        if (line->s[0] != '#') { fprintf(stderr, "[E::%s] no sample line\n", __func__); free(txt.s); return NULL; }
        if (line->s[1] != '#' && fn_ref) {                         /* contigs "name length ..." go in front of #CHROM */
            gzFile g = gzopen(fn_ref, "r");
            kstring_t t = {0, 0, 0};
            while (g && gz_getline(g, &t) >= 0) {
                char *name = strtok(t.s, " \t"), *len = name ? strtok(NULL, " \t") : NULL;
                if (name && len) ks_printf(&txt, "##contig=<ID=%s,length=%ld>\n", name, atol(len));
            }
            if (g) gzclose(g);
            free(t.s);
        }
        ks_putn(&txt, line->s, line->l);
        if (line->s[1] != '#') break;
        ks_putc(&txt, '\n');
    }
    if (txt.l == 0) { free(txt.s); return NULL; }
    h = bcf_hdr_init();
    h->text = txt.s; h->l_text = (int32_t)txt.l + 1; h->m_text = (int32_t)txt.m;
    bcf_hdr_parse(h);
    return h;
}

static reader_t *reader_open(const char *fn, int is_bcf, const char *fn_ref, int keep_flt)
{
    reader_t *r = (reader_t*)calloc(1, sizeof(*r));
    r->is_bcf = is_bcf; r->keep_flt = keep_flt;
    if (is_bcf) {
        if ((r->bf = bgzr_open(fn)) != NULL) r->h = bcf_hdr_read_stream(r->bf);
        r->b = bcf_init1();
    } else {
        if ((r->tf = gzopen(strcmp(fn, "-") ? fn : "/dev/stdin", "r")) != NULL) { gzbuffer(r->tf, 1 << 20); r->h = read_text_header(r->tf, fn_ref, &r->line); }
    }
    if (r->h == NULL) { fprintf(stderr, "[E::%s] cannot read a %s header from '%s'\n", __func__, is_bcf ? "BCF" : "VCF", fn); return NULL; }
    r->id_gt = bcf_id2int(r->h, BCF_DT_ID, "GT");
    r->id_cigar = bcf_id2int(r->h, BCF_DT_ID, "CIGAR");
    r->id_end = bcf_id2int(r->h, BCF_DT_ID, "END");
    return r;
}

static void reader_close(reader_t *r)
{
    if (!r) return;
    pf_destroy(r->pf);
    if (r->tf) gzclose(r->tf);
    if (r->bf) bgzr_close(r->bf);
    if (r->b) bcf_destroy1(r->b);
    if (r->h) bcf_hdr_destroy(r->h);
    free(r->line.s); free(r);
}

static void rec_set_alleles(rec_t *c, int n)
{
    if (n > c->m_allele) { c->allele = (char**)realloc(c->allele, (size_t)n * sizeof(char*)); c->m_allele = n; }
    c->n_allele = n;
}

/* a record's complaint, kept with the record: lines are parsed ahead of their turn and out of order */
static void rec_err(char *err, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err, 256, fmt, ap);
    va_end(ap);
}

/* one text line (modified in place) -> rec_t.  Returns 0, or <-1 on a record this importer cannot take (message in err).
 * Reads the header only: lines are parsed side by side. */
static int parse_text_line(const reader_t *r, char *line, rec_t *c, char *err)
{
    char *f[10], *p, *q;
    int i, nf = 0, gt_idx = -1, ns = 0;
    for (p = line; nf < 9; ++nf) {                                 /* the eight fixed columns and FORMAT */
        f[nf] = p;
        if ((q = strchr(p, '\t')) == NULL) { ++nf; p = NULL; break; }
        *q = 0; p = q + 1;
    }
    if (nf < 8) { rec_err(err, "[E::%s] fewer than 8 columns\n", "read_text_record"); return -2; }
    if ((c->rid = bcf_id2int(r->h, BCF_DT_CTG, f[0])) < 0) {
        rec_err(err, "[E::%s] contig '%s' is not in the header (declare it, or give the contig list with -t)\n", "read_text_record", f[0]);
        return -2;
    }
    c->pos = atoi(f[1]) - 1;
    c->pool.l = 0;
    ks_puts(&c->pool, f[3]); ks_putc(&c->pool, 0);
    c->rlen = (int)strlen(f[3]);
    {   /* alleles: REF then the comma-separated ALTs ("." = none) */
        int n = 1;                                                  /* REF */
        const size_t off = c->pool.l;
        if (strcmp(f[4], ".")) { char *t; ++n; for (t = f[4]; *t; ++t) n += *t == ','; ks_puts(&c->pool, f[4]); ks_putc(&c->pool, 0); }
        rec_set_alleles(c, n);
        c->allele[0] = c->pool.s;
        if (c->n_allele > 1) {
            char *a = c->pool.s + off;
            for (i = 1; i < c->n_allele; ++i) { c->allele[i] = a; if ((q = strchr(a, ',')) != NULL) { *q = 0; a = q + 1; } }
        }
    }
    /* FILTER: "." and PASS (id 0) pass; anything else is filtered (reference vcf.c:1105-1113) */
    c->filtered = 0;
    if (strcmp(f[6], ".")) {
        int n_flt = 0, first = -1;
        char *save = NULL, *t;
        for (t = strtok_r(f[6], ";", &save); t; t = strtok_r(NULL, ";", &save)) {
            const int id = bcf_id2int(r->h, BCF_DT_ID, t);
            if (id >= 0) { if (n_flt++ == 0) first = id; }
        }
        c->filtered = !(n_flt == 0 || (n_flt == 1 && first == 0));
    }
    /* INFO: END (rlen = END - pos, reference vcf.c:648) and CIGAR */
    c->cigar.l = 0; if (c->cigar.s) c->cigar.s[0] = 0;
    if (strcmp(f[7], ".")) {
        char *save = NULL, *t;
        for (t = strtok_r(f[7], ";", &save); t; t = strtok_r(NULL, ";", &save)) {
            if (strncmp(t, "END=", 4) == 0 && r->id_end >= 0) c->rlen = atoi(t + 4) - c->pos;
            else if (strncmp(t, "CIGAR=", 6) == 0 && r->id_cigar >= 0) ks_puts(&c->cigar, t + 6);
        }
    }
    /* FORMAT / samples: only GT */
    c->n_sample = 0;
    if (nf >= 9 && p) {
        char *save = NULL, *t;
        for (t = strtok_r(f[8], ":", &save), i = 0; t; t = strtok_r(NULL, ":", &save), ++i) if (strcmp(t, "GT") == 0) gt_idx = i;
        if (gt_idx < 0) { rec_err(err, "[E::%s] no GT in FORMAT\n", "read_text_record"); return -2; }
        if (c->n_allele > 127) { rec_err(err, "[E::%s] %d alleles in one record: allele numbers are kept in 7 bits\n", "read_text_record", c->n_allele); return -2; }   /* (the BCF reader refuses the same) */
        ns = r->h->n[BCF_DT_SAMPLE];
        if ((size_t)ns * 2 > c->m_gt) { c->m_gt = (size_t)ns * 2; c->gt = (int8_t*)realloc(c->gt, c->m_gt); }
        for (i = 0; i < ns; ++i) {
            int k, g;
            if (p == NULL) { rec_err(err, "[E::%s] fewer sample columns than the header names\n", "read_text_record"); return -2; }
            if ((q = strchr(p, '\t')) != NULL) *q = 0;
            for (k = 0; k < gt_idx && p; ++k) { p = strchr(p, ':'); if (p) ++p; }      /* the gt_idx-th sub-field */
            for (g = 0; p && *p && *p != ':'; ++g) {
                int a;
                if (g >= 2) { rec_err(err, "[E::%s] only diploid genotypes can be imported\n", "read_text_record"); return -2; }
                if (*p == '.') { a = -1; ++p; }
                else {
                    char *e;
                    a = (int)strtol(p, &e, 10);
                    if (e == p || a < 0) { rec_err(err, "[E::%s] malformed genotype '%.8s'\n", "read_text_record", p); return -2; }   /* (as the BCF reader: -3 is no allele) */
                    p = e;
                }
                if (a >= c->n_allele) { rec_err(err, "[E::%s] genotype refers to allele %d of %d\n", "read_text_record", a, c->n_allele); return -2; }
                c->gt[2 * i + g] = (int8_t)a;
                if (*p == '/' || *p == '|') ++p;
            }
            if (g != 2) { rec_err(err, "[E::%s] only diploid genotypes can be imported\n", "read_text_record"); return -2; }
            p = q ? q + 1 : NULL;
        }
        c->n_sample = ns;
    }
    return 0;
}

/* ---- text lines parsed ahead: while the importer works through one batch of records, a filler thread reads the lines of
 * the next (the only serial part: gzgets) and parses them on several threads -- splitting a line into 2 x n_sample allele
 * numbers is most of what `bgt import` does on the host.  Records, and their complaints, are handed over in file order. */
typedef struct { kstring_t line; rec_t rec; int ret; char err[256]; } pf_slot_t;
struct prefetch_s {
    reader_t *r;
    pf_slot_t *slot[2];
    int n[2], cap, cur, at, n_threads;
    int eof;                                  /* the filler met the end of the file */
    int started;                              /* a filler thread is running on batch cur ^ 1 */
    pthread_t filler;
    int fill_batch, next;                     /* the parse workers' shared cursor (a batch at a time) */
    pthread_mutex_t lock;
};

static void *pf_parse_worker(void *p)
{
    prefetch_t *pf = (prefetch_t*)p;
    pf_slot_t *b = pf->slot[pf->fill_batch];
    const int n = pf->n[pf->fill_batch];
    for (;;) {
        int i;
        pthread_mutex_lock(&pf->lock);
        i = pf->next; pf->next += 4;
        pthread_mutex_unlock(&pf->lock);
        if (i >= n) break;
        for (int k = i; k < i + 4 && k < n; ++k) { b[k].err[0] = 0; b[k].ret = parse_text_line(pf->r, b[k].line.s, &b[k].rec, b[k].err); }
    }
    return NULL;
}

static void *pf_fill(void *p)
{
    prefetch_t *pf = (prefetch_t*)p;
    const int w = pf->fill_batch;
    pf_slot_t *b = pf->slot[w];
    size_t bytes = 0;
    int n = 0, t, n_started = 0;
    pthread_t th[64];
    while (n < pf->cap && bytes < ((size_t)32 << 20)) {
        if (gz_getline(pf->r->tf, &b[n].line) < 0) { pf->eof = 1; break; }
        bytes += b[n].line.l;
        ++n;
    }
    pf->n[w] = n;
    pf->next = 0;
    for (t = 1; t < pf->n_threads && t * 8 < n; ++t) if (pthread_create(&th[n_started], NULL, pf_parse_worker, pf) == 0) ++n_started;
    pf_parse_worker(pf);
    for (t = 0; t < n_started; ++t) pthread_join(th[t], NULL);
    return NULL;
}

static prefetch_t *pf_init(reader_t *r)
{
    prefetch_t *pf = (prefetch_t*)calloc(1, sizeof(*pf));
    const char *e = getenv("BGT_THREADS");
    long nt = e ? atol(e) : sysconf(_SC_NPROCESSORS_ONLN);
    pf->r = r; pf->cap = 1024;
    pf->n_threads = nt < 1 ? 1 : nt > 16 ? 16 : (int)nt;
    pf->slot[0] = (pf_slot_t*)calloc((size_t)pf->cap, sizeof(pf_slot_t));
    pf->slot[1] = (pf_slot_t*)calloc((size_t)pf->cap, sizeof(pf_slot_t));
    pthread_mutex_init(&pf->lock, NULL);
    return pf;
}

static void pf_destroy(prefetch_t *pf)
{
    int w, i;
    if (!pf) return;
    if (pf->started) pthread_join(pf->filler, NULL);
    for (w = 0; w < 2; ++w) {
        for (i = 0; i < pf->cap; ++i) {
            pf_slot_t *s = &pf->slot[w][i];
            free(s->line.s); free(s->rec.allele); free(s->rec.cigar.s); free(s->rec.gt); free(s->rec.pool.s);
        }
        free(pf->slot[w]);
    }
    pthread_mutex_destroy(&pf->lock);
    free(pf);
}

/* text record -> rec_t.  Returns 0, -1 at EOF, <-1 on a record this importer cannot take. */
static int read_text_record(reader_t *r, rec_t *c)
{
    prefetch_t *pf = r->pf;
    pf_slot_t *s;
    if (pf == NULL) pf = r->pf = pf_init(r);
    while (pf->at == pf->n[pf->cur]) {                            /* this batch is used up: take the one filled meanwhile */
        if (pf->started) { pthread_join(pf->filler, NULL); pf->started = 0; }
        else if (pf->eof) return -1;
        else { pf->fill_batch = pf->cur ^ 1; pf_fill(pf); }         /* the first batch (or no thread to be had): nothing to overlap with */
        pf->cur ^= 1; pf->at = 0;
        if (!pf->eof) {                                             /* have the next one filled while this one is used */
            pf->fill_batch = pf->cur ^ 1;
            if (pthread_create(&pf->filler, NULL, pf_fill, pf) == 0) pf->started = 1;
        }
    }
    s = &pf->slot[pf->cur][pf->at++];
    if (s->ret < -1) { fputs(s->err, stderr); return s->ret; }
    { rec_t tmp = *c; *c = s->rec; s->rec = tmp; }                /* the record changes hands; its old storage is reused by the slot */
    return 0;
}

/* typed values of a BCF record */
static int tv_bytes(int t) { return t == 1 || t == 7 ? 1 : t == 2 ? 2 : (t == 3 || t == 5) ? 4 : 0; }
static int32_t tv_int(const uint8_t *p, int t)
{
    if (t == 1) return *(const int8_t*)p;
    if (t == 2) { int16_t v; memcpy(&v, p, 2); return v; }
    { int32_t v; memcpy(&v, p, 4); return v; }
}

static int read_bcf_record(reader_t *r, rec_t *c)
{
    const uint8_t *p, *q;
    int i, n, type, ret;
    if ((ret = bcf_read1_stream(r->bf, r->b)) != 0) return ret < -1 ? -2 : -1;
    c->rid = r->b->rid; c->pos = r->b->pos; c->rlen = r->b->rlen;
    c->pool.l = 0;
    p = (const uint8_t*)r->b->shared.s;
    n = bcf_dec_size(p, &q, &type); p = q + n;                               /* ID */
    rec_set_alleles(c, (int)r->b->n_allele);
    {
        size_t off[1024];
        if (c->n_allele > 1024) return -2;
        for (i = 0; i < c->n_allele; ++i) { n = bcf_dec_size(p, &q, &type); off[i] = c->pool.l; ks_putn(&c->pool, (const char*)q, (size_t)n); ks_putc(&c->pool, 0); p = q + n; }
        for (i = 0; i < c->n_allele; ++i) c->allele[i] = c->pool.s + off[i];
    }
    n = bcf_dec_size(p, &q, &type);                                          /* FILTER */
    c->filtered = !(n == 0 || (n == 1 && tv_int(q, type) == 0));
    p = q + (size_t)n * tv_bytes(type);
    c->cigar.l = 0; if (c->cigar.s) c->cigar.s[0] = 0;
    for (i = 0; i < (int)r->b->n_info; ++i) {
        int kt, key;
        bcf_dec_size(p, &q, &kt); key = tv_int(q, kt); p = q + tv_bytes(kt);
        n = bcf_dec_size(p, &q, &type);
        if (key == r->id_cigar && type == BCF_BT_CHAR) ks_putn(&c->cigar, (const char*)q, (size_t)n);
        p = q + (size_t)n * tv_bytes(type);
    }
    c->n_sample = (int)r->b->n_sample;
    if (c->n_allele > 127) { fprintf(stderr, "[E::%s] %d alleles in one record: allele numbers are kept in 7 bits\n", __func__, c->n_allele); return -2; }
    if ((size_t)c->n_sample * 2 > c->m_gt) { c->m_gt = (size_t)c->n_sample * 2; c->gt = (int8_t*)realloc(c->gt, c->m_gt); }
    p = (const uint8_t*)r->b->indiv.s;
    {
        const uint8_t *end = p + r->b->indiv.l;                              /* the record's own lengths are not trusted */
        for (i = 0; i < (int)r->b->n_fmt; ++i) {
            int kt, key, k;
            size_t bytes;
            if (p + 2 > end) goto truncated;
            bcf_dec_size(p, &q, &kt);
            if (q + tv_bytes(kt) + 1 > end) goto truncated;
            key = tv_int(q, kt); p = q + tv_bytes(kt);
            n = bcf_dec_size(p, &q, &type);
            bytes = (size_t)(n < 0 ? 0 : n) * (size_t)tv_bytes(type) * (size_t)c->n_sample;
            if (n < 0 || q > end || bytes > (size_t)(end - q)) goto truncated;
            if (key == r->id_gt) {
                if (n != 2) { fprintf(stderr, "[E::%s] only diploid genotypes can be imported\n", __func__); return -2; }
                for (k = 0; k < 2 * c->n_sample; ++k) {
                    const int al = (tv_int(q + (size_t)k * tv_bytes(type), type) >> 1) - 1;      /* -1 = missing */
                    if (al >= c->n_allele) {                                 /* (the text reader refuses it too) */
                        fprintf(stderr, "[E::%s] a genotype names allele %d of a record with %d alleles\n", __func__, al, c->n_allele);
                        return -2;
                    }
                    c->gt[k] = (int8_t)(al < 0 ? -1 : al);
                }
            }
            p = q + bytes;
        }
    }
    return 0;
truncated:
    fprintf(stderr, "[E::%s] a record's FORMAT block is shorter than its own field sizes say\n", __func__);
    return -2;
}

/* the next record that passes FILTER (or any, with -F) */
static int read_record(reader_t *r, rec_t *c)
{
    int ret;
    do ret = r->is_bcf ? read_bcf_record(r, c) : read_text_record(r, c);
    while (ret == 0 && !r->keep_flt && c->filtered);
    return ret;
}

/* ---------------------------------------------------------------------------------------------------------------
 * atomizer
 * --------------------------------------------------------------------------------------------------------------- */
typedef struct {
    int rid, pos, rlen, anum, has_multi, from_new, l_ref, l_alt;
    char *ref, *alt;                          /* one allocation: ref NUL alt NUL */
    uint8_t *gt; int n_gt;
} atom_t;
typedef struct { int n, m; atom_t *a; } atom_v;

static int atom_cmp(const atom_t *a, const atom_t *b)                /* reference atomic.h:36-42 */
{
    if (a->rid != b->rid) return a->rid - b->rid;
    if (a->pos != b->pos) return a->pos - b->pos;
    if (a->rlen != b->rlen) return a->rlen - b->rlen;
    return strcmp(a->alt, b->alt);
}
static int atom_cmp_sort(const void *x, const void *y)                /* equal alleles: the buffered one first */
{
    const atom_t *a = (const atom_t*)x, *b = (const atom_t*)y;
    const int c = atom_cmp(a, b);
    return c ? c : a->from_new - b->from_new;
}

static void add_atom(atom_v *v, int rid, int pos, int rlen, int anum, const char *ref, int l_ref, const char *alt, int l_alt)
{
    atom_t *p;
    if (v->n == v->m) {
        const int old = v->m;
        v->m = v->m ? v->m << 1 : 4;
        v->a = (atom_t*)realloc(v->a, (size_t)v->m * sizeof(atom_t));
        memset(v->a + old, 0, (size_t)(v->m - old) * sizeof(atom_t));
    }
    p = &v->a[v->n++];
    p->rid = rid; p->pos = pos; p->rlen = rlen; p->anum = anum; p->from_new = 1; p->has_multi = 0;
    p->ref = (char*)realloc(p->ref, (size_t)l_ref + (size_t)l_alt + 2);
    memcpy(p->ref, ref, (size_t)l_ref); p->ref[l_ref] = 0;
    p->alt = p->ref + l_ref + 1;
    memcpy(p->alt, alt, (size_t)l_alt); p->alt[l_alt] = 0;
    p->l_ref = l_ref; p->l_alt = l_alt;
}

/* genotypes of the new atoms of record c; duplicates go to the back; returns the number of distinct atoms
 * (reference atomic.c:15-76) */
static int atoms_finish(const rec_t *c, atom_v *v)
{
    const int n = v->n;
    int i, k, has_dup = 0;
    int *eq = (int*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int)), *tr = (int*)malloc((size_t)c->n_allele * sizeof(int));
    qsort(v->a, (size_t)n, sizeof(atom_t), atom_cmp_sort);
    for (i = 1, eq[0] = 0; i < n; ++i) {                           /* eq[k]: the first atom equal to atom k */
        eq[i] = atom_cmp(&v->a[i - 1], &v->a[i]) ? i : eq[i - 1];
        if (eq[i] == eq[i - 1]) has_dup = 1;
    }
    tr[0] = 0;
    for (k = 0; k < n; ++k) {
        atom_t *ak = &v->a[k];
        int s;
        if (eq[k] != k || !ak->from_new) continue;                 /* a duplicate, or buffered from an earlier record */
        ak->has_multi = 0;
        for (i = 1; i < c->n_allele; ++i) tr[i] = 0;               /* allele number of the record -> code of this atom */
        for (i = 0; i < n; ++i) {
            const atom_t *ai = &v->a[i];
            if (!ai->from_new) continue;
            if (eq[i] == eq[k]) tr[ai->anum] = 1;                  /* the same atomic allele */
            else if (ai->pos < ak->pos + ak->rlen && ak->pos < ai->pos + ai->rlen) tr[ai->anum] = 3;   /* overlapping */
        }
        ak->gt = (uint8_t*)realloc(ak->gt, (size_t)c->n_sample * 2);
        ak->n_gt = 2 * c->n_sample;
        for (s = 0; s < 2 * c->n_sample; ++s) {
            const int code = c->gt[s] < 0 ? 2 : tr[c->gt[s]];
            ak->gt[s] = (uint8_t)code;
            if (code == 3) ak->has_multi = 1;
        }
    }
    if (has_dup) {                                                 /* distinct atoms first (in order), duplicates behind */
        atom_t *swap = (atom_t*)malloc((size_t)n * sizeof(atom_t));
        int j = 0, b = n - 1;
        memcpy(swap, v->a, (size_t)n * sizeof(atom_t));
        for (i = 0; i < n; ++i) { if (eq[i] == i) v->a[j++] = swap[i]; else v->a[b--] = swap[i]; }
        free(swap);
        v->n = j;
    }
    free(eq); free(tr);
    return v->n;
}

/* record -> atoms appended to v (reference atomic.c:98-179) */
static int atomize(const bcf_hdr_t *h, const rec_t *c, atom_v *v)
{
    const char *ref = c->allele[0], *pc = c->cigar.l ? c->cigar.s : NULL;
    const int l_ref = (int)strlen(ref);
    kstring_t cg = {0, 0, 0};
    int i;
    for (i = 0; i < v->n; ++i) v->a[i].from_new = 0;
    for (i = 1; i < c->n_allele; ++i) {
        const char *alt = c->allele[i], *p;
        const int l_alt = (int)strlen(alt);
        int x = 0, y = 0;
        if (c->rlen != l_ref || (alt[0] == '<' && l_alt > 0 && alt[l_alt - 1] == '>')) {      /* symbolic: kept whole */
            add_atom(v, c->rid, c->pos, c->rlen, i, ref, l_ref, alt, l_alt);
            continue;
        }
        cg.l = 0;
        if (pc) {                                                  /* this allele's part of INFO/CIGAR */
            const char *e = pc;
            while (*e && *e != ',') ++e;
            if (e == pc) { fprintf(stderr, "[E::%s] incomplete CIGAR\n", __func__); free(cg.s); return -1; }
            ks_putn(&cg, pc, (size_t)(e - pc));
            pc = *e ? e + 1 : e;
        } else if (l_alt == c->rlen) ks_printf(&cg, "%dM", c->rlen);
        else {
            const int l = l_alt - c->rlen;
            int rest;
            ks_puts(&cg, "1M");
            if (l > 0) { ks_printf(&cg, "%dI", l); rest = c->rlen - 1; }
            else { ks_printf(&cg, "%dD", -l); rest = l_alt - 1; }
            if (rest) ks_printf(&cg, "%dM", rest);
        }
        {   /* the walk below indexes ref[] and alt[] by what the CIGAR says: it must consume exactly the two alleles */
            long cx = 0, cy = 0;
            int bad = 0;
            for (p = cg.s; *p && !bad; ++p) {
                char *e;
                const long l = strtol(p, &e, 10);
                if (e == p || l < 0 || l > INT32_MAX) { bad = 1; break; }
                p = e;
                if (*p == 'M' || *p == '=' || *p == 'X') { cx += l; cy += l; }
                else if (*p == 'I') cy += l;
                else if (*p == 'D') cx += l;
                else bad = 1;
                if (*p == 0) break;
            }
            if (bad || cx != l_ref || cy != l_alt) {
                fprintf(stderr, "[E::%s] CIGAR '%s' does not span REF (%d) and ALT (%d) at %s:%d\n", __func__, cg.s, l_ref, l_alt,
                        h->id[BCF_DT_CTG][c->rid].key, c->pos + 1);
                free(cg.s);
                return -1;
            }
        }
        for (p = cg.s; *p; ++p) {
            char *e;
            const int l = (int)strtol(p, &e, 10);
            int j;
            p = e;
            if (*p == 'M' || *p == '=' || *p == 'X') {
                for (j = 0; j < l; ++j)
                    if (ref[x + j] != alt[y + j]) add_atom(v, c->rid, c->pos + x + j, 1, i, ref + x + j, 1, alt + y + j, 1);
                x += l; y += l;
            } else if (*p == 'I') {
                if (x == 0 || y == 0)
                    fprintf(stderr, "[W::%s] invalid insertion (%d,%d) at %s:%d\n", __func__, x, y, h->id[BCF_DT_CTG][c->rid].key, c->pos + 1);
                else add_atom(v, c->rid, c->pos + x - 1, 1, i, ref + x - 1, 1, alt + y - 1, l + 1);
                y += l;
            } else if (*p == 'D') {
                if (x == 0 || y == 0) { fprintf(stderr, "[E::%s] deletion at the first base of %s:%d\n", __func__, h->id[BCF_DT_CTG][c->rid].key, c->pos + 1); free(cg.s); return -1; }
                add_atom(v, c->rid, c->pos + x - 1, l + 1, i, ref + x - 1, l + 1, alt + y - 1, 1);
                x += l;
            } else if (*p == 0) break;
        }
    }
    free(cg.s);
    atoms_finish(c, v);
    return 0;
}

/* the buffer that puts atoms of successive records in order (reference atomic.c:190-262) */
typedef struct { reader_t *in; atom_v a; rec_t cur; int start, no_more, failed; } atombuf_t;

static int atombuf_advance(atombuf_t *ab)      /* atomize the look-ahead record, fetch the next one */
{
    int ret;
    if (atomize(ab->in->h, &ab->cur, &ab->a) < 0) { ab->failed = 1; return -1; }
    ret = read_record(ab->in, &ab->cur);
    if (ret < -1) { ab->failed = 1; return -1; }
    if (ret < 0) ab->no_more = 1;
    return 0;
}

static atombuf_t *atombuf_init(reader_t *in)
{
    atombuf_t *ab = (atombuf_t*)calloc(1, sizeof(*ab));
    const int ret = read_record(in, &ab->cur);
    ab->in = in;
    if (ret < -1) ab->failed = 1;
    if (ret == 0) atombuf_advance(ab); else ab->no_more = 1;
    return ab;
}

static const atom_t *atombuf_read(atombuf_t *ab)
{
    if (ab->failed) return NULL;
    if (ab->start == ab->a.n) {
        if (ab->no_more) return NULL;
        ab->a.n = ab->start = 0;
        if (atombuf_advance(ab) < 0) return NULL;
    }
    for (;;) {
        const atom_t *f = &ab->a.a[ab->start];
        if (ab->no_more || f->rid < ab->cur.rid || (f->rid == ab->cur.rid && f->pos < ab->cur.pos)) return &ab->a.a[ab->start++];
        if (ab->start != 0) {                                      /* served atoms leave the front (their storage is kept) */
            atom_t *tmp = (atom_t*)malloc((size_t)ab->start * sizeof(atom_t));
            memcpy(tmp, ab->a.a, (size_t)ab->start * sizeof(atom_t));
            memmove(ab->a.a, ab->a.a + ab->start, (size_t)(ab->a.n - ab->start) * sizeof(atom_t));
            ab->a.n -= ab->start;
            memcpy(ab->a.a + ab->a.n, tmp, (size_t)ab->start * sizeof(atom_t));
            ab->start = 0;
            free(tmp);
        }
        if (atombuf_advance(ab) < 0) return NULL;
    }
}

static void atombuf_destroy(atombuf_t *ab)
{
    int i;
    for (i = 0; i < ab->a.m; ++i) { free(ab->a.a[i].ref); free(ab->a.a[i].gt); }
    free(ab->a.a); free(ab->cur.allele); free(ab->cur.cigar.s); free(ab->cur.gt); free(ab->cur.pool.s);
    free(ab);
}

/* ---------------------------------------------------------------------------------------------------------------
 * bgt import
 * --------------------------------------------------------------------------------------------------------------- */
/* header of the site-only output (reference import.c:48-55, vcf.c:1044-1072, :210-231): the input's text up to the INFO
 * column of #CHROM, plus a FORMAT/GT line if the input has none, plus INFO/_row -- new lines go right before #CHROM */
static bcf_hdr_t *site_header(const bcf_hdr_t *h0)
{
    kstring_t s = {0, 0, 0};
    const char *chrom = NULL, *p;
    bcf_hdr_t *h;
    int i, n_added = 1;
    for (p = h0->text; (p = strstr(p, "#CHROM\t")) != NULL; ++p) if (p == h0->text || p[-1] == '\n') { chrom = p; break; }
    if (chrom == NULL || chrom < h0->text || (size_t)(chrom - h0->text) > (size_t)INT32_MAX) return NULL;
    for (p = chrom, i = 0; i < 8 && (p = strchr(p, '\t')) != NULL; ++i) ++p;      /* p: behind the tab after INFO, or NULL */
    ks_putn(&s, h0->text, (size_t)(chrom - h0->text));
    if (bcf_id2int(h0, BCF_DT_ID, "GT") < 0) { ks_puts(&s, "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n"); ++n_added; }
    ks_puts(&s, "##INFO=<ID=_row,Number=1,Type=Integer,Description=\"row number\">\n");
    if (p) ks_putn(&s, chrom, (size_t)(p - 1 - chrom)); else ks_puts(&s, chrom);
    h = bcf_hdr_init();
    h->text = s.s; h->l_text = (int32_t)s.l + 1; h->m_text = (int32_t)s.m;
    bcf_hdr_parse(h);
    /* The reference's bcf_hdr_append (vcf.c:210-231) grows the text by the line and its newline but l_text by the line
     * only, so the length it writes into the BCF is one short per appended line: no terminating NUL after one line, the
     * last character of "#CHROM ... INFO" cut after two.  Readers cope; the bytes must match. */
    h->l_text = (int32_t)s.l + 1 - n_added;
    return h;
}

static int flush_encoder(bgth_encoder_t *enc, FILE *fp)
{
    uint8_t *chunk = NULL;
    const int64_t n = bgth_encoder_take(enc, &chunk);
    if (n < 0) return -1;
    if (n > 0 && fwrite(chunk, 1, (size_t)n, fp) != (size_t)n) { bgth_encoder_free_image(chunk); return -1; }
    bgth_encoder_free_image(chunk);
    return 0;
}

int main_import(int argc, char *argv[])
{
    int c, clevel = -1, is_vcf = 0, keep_flt = 0, j, rc = 1;
    const char *fn_ref = NULL, *prefix;
    char *fn;
    reader_t *in = NULL;
    atombuf_t *ab = NULL;
    bcf_hdr_t *h0 = NULL;
    bgth_encoder_t *enc = NULL, *enc1 = NULL;             /* enc1: -1, the one-plane file prefix.pb1 (import.c:72-74) */
    FILE *fp_pbf = NULL, *fp_bcf = NULL, *fp_pb1 = NULL;
    uint8_t *rows1 = NULL;
    int gen_pb1 = 0;
    bgzw_t *bz = NULL;
    csi_writer_t *ix = NULL;
    bcf1_t *b = NULL;
    uint8_t *rows = NULL;
    int64_t n = 0, n_buf = 0, cap_rows = 0;
    int m = 0;
    const atom_t *a;

    optind = 0;
    while ((c = getopt(argc, argv, "1l:SFt:")) >= 0) {
        switch (c) {
        case 'l': clevel = atoi(optarg); break;
        case 'S': is_vcf = 1; break;
        case 't': fn_ref = optarg; is_vcf = 1; break;
        case 'F': keep_flt = 1; break;
        case '1': gen_pb1 = 1; break;                                 /* also write prefix.pb1: one plane, bit = (genotype code == 1) */
        default: break;
        }
    }
    if (argc - optind < 2) {
        fprintf(stderr, "Usage: bgt import [options] <out-prefix> <in.bcf>|<in.vcf>|<in.vcf.gz>\n");
        fprintf(stderr, "Options:\n  -S           input is VCF\n  -t FILE      list of reference names and lengths [null]\n  -F           keep filtered variants\n  -1           also write <out-prefix>.pb1 (one bit plane: the ALT allele)\n");
        return 1;
    }
    prefix = argv[optind];
    fn = (char*)malloc(strlen(prefix) + 16);
    b = bcf_init1();
    for (j = optind + 1; j < argc; ++j) {
        if ((in = reader_open(argv[j], !is_vcf, fn_ref, keep_flt)) == NULL) goto done;
        if (in->h->n[BCF_DT_SAMPLE] <= 0) { fprintf(stderr, "[E::%s] '%s' has no samples\n", __func__, argv[j]); goto done; }
        ab = atombuf_init(in);
        if (j == optind + 1) {                                        /* outputs are laid out after the first input */
            FILE *fp;
            int i, depth;
            int64_t max_len = 0, s;
            m = 2 * in->h->n[BCF_DT_SAMPLE];
            if ((h0 = site_header(in->h)) == NULL) { fprintf(stderr, "[E::%s] malformed header\n", __func__); goto done; }
            sprintf(fn, "%s.spl", prefix);
            if ((fp = fopen(fn, "wb")) == NULL) { fprintf(stderr, "[E::%s] cannot create '%s'\n", __func__, fn); goto done; }
            for (i = 0; i < in->h->n[BCF_DT_SAMPLE]; ++i) { fputs(in->h->id[BCF_DT_SAMPLE][i].key, fp); fputc('\n', fp); }
            fclose(fp);
            if ((enc = bgth_encoder_open(m, 2, 13, 0)) == NULL) { fprintf(stderr, "[E::%s] %s\n", __func__, bgth_encoder_last_error()); goto done; }
            sprintf(fn, "%s.pbf", prefix);
            if ((fp_pbf = fopen(fn, "wb")) == NULL) { fprintf(stderr, "[E::%s] cannot create '%s'\n", __func__, fn); goto done; }
            if (gen_pb1) {                                            /* reference import.c:72-74: pbf_open_w(prefix.pb1, m, 1, 13) */
                if ((enc1 = bgth_encoder_open(m, 1, 13, 0)) == NULL) { fprintf(stderr, "[E::%s] %s\n", __func__, bgth_encoder_last_error()); goto done; }
                sprintf(fn, "%s.pb1", prefix);
                if ((fp_pb1 = fopen(fn, "wb")) == NULL) { fprintf(stderr, "[E::%s] cannot create '%s'\n", __func__, fn); goto done; }
            }
            sprintf(fn, "%s.bcf", prefix);
            if ((fp_bcf = fopen(fn, "wb")) == NULL) { fprintf(stderr, "[E::%s] cannot create '%s'\n", __func__, fn); goto done; }
            bz = bgzw_open(fp_bcf, clevel >= 0 && clevel <= 9 ? clevel : -1);
            bcf_hdr_write_stream(bz, h0);
            for (i = 0; i < h0->n[BCF_DT_CTG]; ++i) if (max_len < (int64_t)h0->id[BCF_DT_CTG][i].val->info[0]) max_len = h0->id[BCF_DT_CTG][i].val->info[0];
            if (max_len == 0) max_len = ((int64_t)1 << 31) - 1;
            max_len += 256;
            for (depth = 0, s = 1 << 14; max_len > s; ++depth, s <<= 3) {}      /* reference vcf.c:1013-1014 */
            ix = csi_writer_init(h0->n[BCF_DT_CTG], 14, depth, bgzw_tell(bz));
            cap_rows = ((int64_t)64 << 20) / m; if (cap_rows < 64) cap_rows = 64; if (cap_rows > 16384) cap_rows = 16384;
            rows = (uint8_t*)malloc((size_t)cap_rows * (size_t)m);
            if (gen_pb1) rows1 = (uint8_t*)malloc((size_t)cap_rows * (size_t)m);
        } else if (2 * in->h->n[BCF_DT_SAMPLE] != m) { fprintf(stderr, "[E::%s] '%s' has a different number of samples\n", __func__, argv[j]); goto done; }
        while ((a = atombuf_read(ab)) != NULL) {
            int32_t val = (int32_t)n;
            uint64_t off0;
            if (a->n_gt != m) { fprintf(stderr, "[E::%s] internal: atom with %d genotypes\n", __func__, a->n_gt); goto done; }
            memcpy(rows + (size_t)n_buf * (size_t)m, a->gt, (size_t)m);        /* code = bit 0 plane 0, bit 1 plane 1 (import.c:96-97) */
            if (rows1) {                                                       /* bit1[i] = (a->gt[i] == 1) (import.c:98) */
                uint8_t *dst = rows1 + (size_t)n_buf * (size_t)m;
                int i;
                for (i = 0; i < m; ++i) dst[i] = a->gt[i] == 1;
            }
            if (++n_buf == cap_rows) {
                if (bgth_encoder_write(enc, rows, n_buf) < 0 || flush_encoder(enc, fp_pbf) < 0) { fprintf(stderr, "[E::%s] %s\n", __func__, bgth_encoder_last_error()); goto done; }
                if (enc1 && (bgth_encoder_write(enc1, rows1, n_buf) < 0 || flush_encoder(enc1, fp_pb1) < 0)) { fprintf(stderr, "[E::%s] %s\n", __func__, bgth_encoder_last_error()); goto done; }
                n_buf = 0;
            }
            bcf_set_site(b, a->rid, a->pos, a->rlen, a->ref, a->l_ref, a->alt, a->l_alt, a->has_multi ? "<M>" : NULL);
            bcf_append_info_ints(h0, b, "_row", 1, &val);
            off0 = bgzw_tell(bz);
            bcf_write1_stream(bz, b);
            csi_writer_push(ix, a->rid, a->pos, a->pos + a->rlen, off0, bgzw_tell(bz));
            ++n;
        }
        if (ab->failed) goto done;
        atombuf_destroy(ab); ab = NULL;
        reader_close(in); in = NULL;
    }
    if (n_buf > 0 && bgth_encoder_write(enc, rows, n_buf) < 0) { fprintf(stderr, "[E::%s] %s\n", __func__, bgth_encoder_last_error()); goto done; }
    if (n_buf > 0 && enc1 && bgth_encoder_write(enc1, rows1, n_buf) < 0) { fprintf(stderr, "[E::%s] %s\n", __func__, bgth_encoder_last_error()); goto done; }
    {
        uint8_t *tail = NULL;
        const int64_t nt = bgth_encoder_finish(enc, &tail);
        if (nt < 0 || fwrite(tail, 1, (size_t)nt, fp_pbf) != (size_t)nt) { fprintf(stderr, "[E::%s] %s\n", __func__, bgth_encoder_last_error()); goto done; }
        bgth_encoder_free_image(tail);
    }
    if (enc1) {
        uint8_t *tail = NULL;
        const int64_t nt = bgth_encoder_finish(enc1, &tail);
        if (nt < 0 || fwrite(tail, 1, (size_t)nt, fp_pb1) != (size_t)nt) { fprintf(stderr, "[E::%s] %s\n", __func__, bgth_encoder_last_error()); goto done; }
        bgth_encoder_free_image(tail);
    }
    bgzw_close(bz); bz = NULL;
    fclose(fp_bcf); fp_bcf = NULL;
    sprintf(fn, "%s.bcf.csi", prefix);
    if (csi_writer_save(ix, fn) < 0) { fprintf(stderr, "[E::%s] cannot write '%s'\n", __func__, fn); goto done; }
    rc = 0;
done:
    if (ab) atombuf_destroy(ab);
    if (in) reader_close(in);
    if (bz) bgzw_close(bz);
    if (fp_bcf) fclose(fp_bcf);
    if (fp_pbf) fclose(fp_pbf);
    if (fp_pb1) fclose(fp_pb1);
    if (enc) bgth_encoder_close(enc);
    if (enc1) bgth_encoder_close(enc1);
    free(rows1);
    if (ix) csi_writer_destroy(ix);
    if (h0) bcf_hdr_destroy(h0);
    if (b) bcf_destroy1(b);
    free(rows); free(fn);
    return rc;
}
