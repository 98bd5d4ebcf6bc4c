/* bcf2.c -- see bcf2.h.  Dictionary rules follow the reference (vcf.c:29-215): the ID dictionary
 * starts with the implicit FILTER "PASS" (id 0) and numbers FILTER/INFO/FORMAT IDs in order of first
 * appearance; contigs and samples are numbered in order. */
#include <ctype.h>
#include <math.h>
#include "bcf2.h"

/* ---- a small open-addressing string->index map ---- */
typedef struct { char **key; bcf_idinfo_t *val; int n, cap; int *slot; int nslot; } dict_t;

static uint32_t hash_str(const char *s) { uint32_t h = 2166136261u; while (*s) h = (h ^ (uint8_t)*s++) * 16777619u; return h; }

static void dict_rehash(dict_t *d)
{
    int i;
    d->nslot = d->nslot ? d->nslot * 2 : 64;
    d->slot = (int*)realloc(d->slot, (size_t)d->nslot * sizeof(int));
    for (i = 0; i < d->nslot; ++i) d->slot[i] = -1;
    for (i = 0; i < d->n; ++i) {
        uint32_t h = hash_str(d->key[i]) & (uint32_t)(d->nslot - 1);
        while (d->slot[h] >= 0) h = (h + 1) & (uint32_t)(d->nslot - 1);
        d->slot[h] = i;
    }
}
static int dict_get(const dict_t *d, const char *k)
{
    uint32_t h;
    if (d->nslot == 0) return -1;
    h = hash_str(k) & (uint32_t)(d->nslot - 1);
    while (d->slot[h] >= 0) {
        if (strcmp(d->key[d->slot[h]], k) == 0) return d->slot[h];
        h = (h + 1) & (uint32_t)(d->nslot - 1);
    }
    return -1;
}
static int dict_put(dict_t *d, const char *k, int len, int *absent)
{
    char *s = (char*)malloc((size_t)len + 1);
    int i;
    memcpy(s, k, (size_t)len); s[len] = 0;
    if ((i = dict_get(d, s)) >= 0) { free(s); *absent = 0; return i; }
    if (d->n == d->cap) {
        d->cap = d->cap ? d->cap * 2 : 16;
        d->key = (char**)realloc(d->key, (size_t)d->cap * sizeof(char*));
        d->val = (bcf_idinfo_t*)realloc(d->val, (size_t)d->cap * sizeof(bcf_idinfo_t));
    }
    d->key[d->n] = s;
    d->val[d->n].info[0] = d->val[d->n].info[1] = d->val[d->n].info[2] = 15;
    d->val[d->n].id = d->n;
    ++d->n;
    if (d->n * 2 > d->nslot) dict_rehash(d);
    else {
        uint32_t h = hash_str(s) & (uint32_t)(d->nslot - 1);
        while (d->slot[h] >= 0) h = (h + 1) & (uint32_t)(d->nslot - 1);
        d->slot[h] = d->n - 1;
    }
    *absent = 1;
    return d->n - 1;
}

bcf_hdr_t *bcf_hdr_init(void)
{
    bcf_hdr_t *h = (bcf_hdr_t*)calloc(1, sizeof(*h));
    int i;
    for (i = 0; i < 3; ++i) h->dict[i] = calloc(1, sizeof(dict_t));
    return h;
}

void bcf_hdr_destroy(bcf_hdr_t *h)
{
    int i, j;
    if (!h) return;
    for (i = 0; i < 3; ++i) {
        dict_t *d = (dict_t*)h->dict[i];
        if (d) { for (j = 0; j < d->n; ++j) free(d->key[j]); free(d->key); free(d->val); free(d->slot); free(d); }
        free(h->id[i]);
    }
    free(h->mem.s); free(h->text); free(h);
}

/* one "##KIND=<...>" line: find ID=... (and length=... for contigs) outside quoted strings */
static void parse_meta_line(bcf_hdr_t *h, const char *p, const char *end)
{
    int kind = -1, absent, id_len = 0;
    const char *q, *id = NULL;
    long ctg_len = -1;
    if (end - p < 3 || p[0] != '#' || p[1] != '#') return;
    p += 2;
    for (q = p; q < end && *q != '='; ++q) {}
    if (q == end) return;
    if (q - p == 4 && !strncmp(p, "INFO", 4)) kind = 1;
    else if (q - p == 6 && !strncmp(p, "FILTER", 6)) kind = 0;
    else if (q - p == 6 && !strncmp(p, "FORMAT", 6)) kind = 2;
    else if (q - p == 6 && !strncmp(p, "contig", 6)) kind = 3;
    else return;
    for (; q < end && *q != '<'; ++q) {}
    if (q == end) return;
    p = q + 1;
    while (p < end && *p != '>') {
        const char *val;
        for (q = p; q < end && *q != '='; ++q) {}
        if (q == end) break;
        val = q + 1;
        if (val < end && *val == '"') {
            for (q = val + 1; q < end && *q != '"'; ++q) if (*q == '\\' && q + 1 < end) ++q;
            p = q + 1; if (p < end && *p == ',') ++p;
            continue;
        }
        {
            const char *ve;
            for (ve = val; ve < end && *ve != ',' && *ve != '>'; ++ve) {}
            if (q - p == 2 && !strncmp(p, "ID", 2)) { id = val; id_len = (int)(ve - val); }
            else if (q - p == 6 && !strncmp(p, "length", 6) && isdigit((unsigned char)*val)) ctg_len = strtol(val, NULL, 10);
            p = ve + 1;
        }
    }
    if (!id) return;
    if (kind == 3) {
        dict_t *d = (dict_t*)h->dict[BCF_DT_CTG];
        int i;
        if (ctg_len <= 0) return;                       /* ref vcf.c:98-100: a contig needs a positive length */
        i = dict_put(d, id, id_len, &absent);
        if (absent) d->val[i].info[0] = (uint32_t)ctg_len;
    } else {
        dict_t *d = (dict_t*)h->dict[BCF_DT_ID];
        dict_put(d, id, id_len, &absent);
    }
}

int bcf_hdr_parse(bcf_hdr_t *h)
{
    static const char pass[] = "##FILTER=<ID=PASS,Description=\"All filters passed\">";
    const char *p = h->text, *end = h->text + h->l_text;
    int w;
    parse_meta_line(h, pass, pass + sizeof(pass) - 1);
    while (p < end && *p) {
        const char *q = p;
        while (q < end && *q && *q != '\n') ++q;
        if (p[0] == '#' && p[1] == '#') parse_meta_line(h, p, q);
        else if (p[0] == '#') {                        /* #CHROM line: samples follow the 9th column */
            dict_t *d = (dict_t*)h->dict[BCF_DT_SAMPLE];
            const char *s = p, *t;
            int col = 0, absent;
            for (t = p;; ++t) {
                if (t == q || *t == '\t') {
                    if (++col > 9) dict_put(d, s, (int)(t - s), &absent);
                    s = t + 1;
                    if (t == q) break;
                }
            }
        }
        if (q >= end || *q == 0) break;
        p = q + 1;
    }
    for (w = 0; w < 3; ++w) {
        dict_t *d = (dict_t*)h->dict[w];
        int i;
        h->n[w] = d->n;
        h->id[w] = (bcf_idpair_t*)realloc(h->id[w], (size_t)(d->n ? d->n : 1) * sizeof(bcf_idpair_t));
        for (i = 0; i < d->n; ++i) { h->id[w][i].key = d->key[i]; h->id[w][i].val = &d->val[i]; }
    }
    return 0;
}

int bcf_id2int(const bcf_hdr_t *h, int which, const char *id) { return dict_get((const dict_t*)h->dict[which], id); }

bcf_hdr_t *bcf_hdr_read_stream(bgzr_t *fp)
{
    uint8_t magic[5];
    bcf_hdr_t *h;
    if (bgzr_read(fp, magic, 5) != 5 || memcmp(magic, "BCF\2\2", 5) != 0) return NULL;
    h = bcf_hdr_init();
    if (bgzr_read(fp, &h->l_text, 4) != 4 || h->l_text <= 0) { bcf_hdr_destroy(h); return NULL; }
    h->m_text = h->l_text;
    h->text = (char*)malloc((size_t)h->l_text + 1);
    if (bgzr_read(fp, h->text, (size_t)h->l_text) != h->l_text) { bcf_hdr_destroy(h); return NULL; }
    h->text[h->l_text] = 0;
    bcf_hdr_parse(h);
    return h;
}

/* ---- records ---- */
bcf1_t *bcf_init1(void) { return (bcf1_t*)calloc(1, sizeof(bcf1_t)); }

void bcf_destroy1(bcf1_t *v)
{
    if (!v) return;
    free(v->d.id); free(v->d.allele); free(v->d.flt); free(v->d.info); free(v->d.fmt);
    free(v->shared.s); free(v->indiv.s); free(v);
}

int bcf_read1_stream(bgzr_t *fp, bcf1_t *v)
{
    uint32_t x[8];
    long got = bgzr_read(fp, x, 32);
    if (got == 0) return -1;
    if (got != 32 || x[0] < 24) return -2;
    x[0] -= 24;
    v->shared.l = 0; ks_need(&v->shared, x[0]);
    v->indiv.l = 0; ks_need(&v->indiv, x[1]);
    memcpy(v, x + 2, 16);
    v->n_allele = x[6] >> 16; v->n_info = x[6] & 0xffff;
    v->n_fmt = x[7] >> 24; v->n_sample = x[7] & 0xffffff;
    if (bgzr_read(fp, v->shared.s, x[0]) != (long)x[0]) return -2;
    if (bgzr_read(fp, v->indiv.s, x[1]) != (long)x[1]) return -2;
    v->shared.l = x[0]; v->indiv.l = x[1];
    v->unpacked = 0; v->unpack_ptr = NULL;
    return 0;
}

/* typed values (BCF2 spec 6.3.3): descriptor byte = length<<4 | type, length 15 = "a typed int follows" */
void bcf_enc_size(kstring_t *s, int size, int type)
{
    if (size < 15) { ks_putc(s, size << 4 | type); return; }
    ks_putc(s, 15 << 4 | type);
    if (size < 128) { ks_putc(s, 1 << 4 | BCF_BT_INT8); ks_putc(s, size); }
    else if (size < 32768) { int16_t x = (int16_t)size; ks_putc(s, 1 << 4 | BCF_BT_INT16); ks_putn(s, &x, 2); }
    else { int32_t x = size; ks_putc(s, 1 << 4 | BCF_BT_INT32); ks_putn(s, &x, 4); }
}

void bcf_enc_int1(kstring_t *s, int32_t x)
{
    if (x == INT32_MIN) { bcf_enc_size(s, 1, BCF_BT_INT8); ks_putc(s, INT8_MIN); }
    else if (x <= INT8_MAX && x > INT8_MIN) { bcf_enc_size(s, 1, BCF_BT_INT8); ks_putc(s, x); }
    else if (x <= INT16_MAX && x > INT16_MIN) { int16_t z = (int16_t)x; bcf_enc_size(s, 1, BCF_BT_INT16); ks_putn(s, &z, 2); }
    else { bcf_enc_size(s, 1, BCF_BT_INT32); ks_putn(s, &x, 4); }
}

/* vector of ints in the narrowest type that holds every value above that type's reserved codes
 * (missing = MIN, end-of-vector = MIN+1), ref vcf.c:430-459 */
void bcf_enc_vint(kstring_t *s, int n, const int32_t *a)
{
    int32_t mx = INT32_MIN + 1, mn = INT32_MAX;
    int i;
    if (n == 0) { bcf_enc_size(s, 0, BCF_BT_NULL); return; }
    if (n == 1) { bcf_enc_int1(s, a[0]); return; }
    for (i = 0; i < n; ++i) {
        if (a[i] == INT32_MIN || a[i] == INT32_MIN + 1) continue;
        if (mx < a[i]) mx = a[i];
        if (mn > a[i]) mn = a[i];
    }
    if (mx <= INT8_MAX && mn > INT8_MIN + 1) {
        bcf_enc_size(s, n, BCF_BT_INT8);
        for (i = 0; i < n; ++i) ks_putc(s, a[i] == INT32_MIN + 1 ? INT8_MIN + 1 : a[i] == INT32_MIN ? INT8_MIN : a[i]);
    } else if (mx <= INT16_MAX && mn > INT16_MIN + 1) {
        bcf_enc_size(s, n, BCF_BT_INT16);
        for (i = 0; i < n; ++i) {
            int16_t x = (int16_t)(a[i] == INT32_MIN + 1 ? INT16_MIN + 1 : a[i] == INT32_MIN ? INT16_MIN : a[i]);
            ks_putn(s, &x, 2);
        }
    } else {
        bcf_enc_size(s, n, BCF_BT_INT32);
        for (i = 0; i < n; ++i) ks_putn(s, &a[i], 4);
    }
}

void bcf_enc_vchar(kstring_t *s, int l, const char *a) { bcf_enc_size(s, l, BCF_BT_CHAR); ks_putn(s, a, (size_t)l); }

int bcf_append_info_ints(const bcf_hdr_t *h, bcf1_t *b, const char *key, int n, const int32_t *vals)
{
    int id = bcf_id2int(h, BCF_DT_ID, key);
    if (id < 0) return -1;
    b->n_info = b->n_info + 1;
    bcf_enc_int1(&b->shared, id);
    bcf_enc_vint(&b->shared, n, vals);
    return 0;
}

void bcf_set_site(bcf1_t *b, int rid, int pos, int rlen, const char *ref, int l_ref, const char *alt, int l_alt,
                  const char *alt2)
{
    b->rid = rid; b->pos = pos; b->rlen = rlen; b->qual = 0;
    b->n_info = 0; b->n_fmt = 0; b->n_sample = 0;
    b->n_allele = alt2 ? 3 : 2;
    b->shared.l = b->indiv.l = 0;
    bcf_enc_size(&b->shared, 0, BCF_BT_CHAR);
    bcf_enc_vchar(&b->shared, l_ref, ref);
    bcf_enc_vchar(&b->shared, l_alt, alt);
    if (alt2) bcf_enc_vchar(&b->shared, (int)strlen(alt2), alt2);
    bcf_enc_vint(&b->shared, 0, 0);
    b->unpacked = 0;
}

/* ---- decoding for text output ---- */
static const int type_bytes[8] = {0, 1, 2, 4, 0, 4, 0, 1};

static int32_t dec_int(const uint8_t *p, int type)
{
    if (type == BCF_BT_INT8) return *(const int8_t*)p;
    if (type == BCF_BT_INT16) { int16_t v; memcpy(&v, p, 2); return v; }
    { int32_t v; memcpy(&v, p, 4); return v; }
}
static int dec_size(const uint8_t *p, const uint8_t **q, int *type)
{
    *type = *p & 15;
    if (*p >> 4 != 15) { *q = p + 1; return *p >> 4; }
    { int t = p[1] & 15; int32_t n = dec_int(p + 2, t); *q = p + 2 + type_bytes[t]; return n; }
}

/* a typed array as VCF text (ref vcf.c bcf_fmt_array semantics: missing -> '.', vector end stops) */
static void fmt_array(kstring_t *s, int n, int type, const uint8_t *p)
{
    int j;
    if (n == 0) { ks_putc(s, '.'); return; }
    if (type == BCF_BT_CHAR) {
        for (j = 0; j < n && p[j]; ++j) ks_putc(s, p[j]);
        return;
    }
    for (j = 0; j < n; ++j) {
        if (type == BCF_BT_FLOAT) {
            float f; uint32_t u;
            memcpy(&f, p + 4 * j, 4); memcpy(&u, p + 4 * j, 4);
            if (u == 0x7F800002u) break;
            if (j) ks_putc(s, ',');
            if (u == 0x7F800001u) ks_putc(s, '.'); else ks_printf(s, "%g", f);
        } else {
            int32_t v = dec_int(p + j * type_bytes[type], type);
            int32_t miss = type == BCF_BT_INT8 ? INT8_MIN : type == BCF_BT_INT16 ? INT16_MIN : INT32_MIN;
            if (v == miss + 1) break;
            if (j) ks_putc(s, ',');
            if (v == miss) ks_putc(s, '.'); else ks_puti(s, v);
        }
    }
}

/* size + type of a typed BCF2 value at p; *q = first byte of its payload (for readers outside this file) */
int bcf_dec_size(const uint8_t *p, const uint8_t **q, int *type) { return dec_size(p, q, type); }

int vcf_format1(const bcf_hdr_t *h, const bcf1_t *v, kstring_t *s)
{
    const uint8_t *p = (const uint8_t*)v->shared.s, *q;
    int i, n, type;
    s->l = 0;
    ks_puts(s, h->id[BCF_DT_CTG][v->rid].key);
    ks_putc(s, '\t'); ks_puti(s, (long long)v->pos + 1);
    ks_putc(s, '\t');                                   /* ID */
    n = dec_size(p, &q, &type);
    if (n == 0) ks_putc(s, '.'); else ks_putn(s, q, (size_t)n);
    p = q + n;
    for (i = 0; i < (int)v->n_allele; ++i) {            /* REF, ALT */
        n = dec_size(p, &q, &type);
        ks_putc(s, i == 0 ? '\t' : i == 1 ? '\t' : ',');
        ks_putn(s, q, (size_t)n);
        p = q + n;
    }
    if (v->n_allele == 0) ks_puts(s, "\t.");
    if (v->n_allele < 2) ks_puts(s, "\t.");
    ks_putc(s, '\t');                                   /* QUAL */
    { uint32_t u; memcpy(&u, &v->qual, 4); if (u == 0x7F800001u) ks_putc(s, '.'); else ks_printf(s, "%g", v->qual); }
    ks_putc(s, '\t');                                   /* FILTER */
    n = dec_size(p, &q, &type);
    if (n == 0) ks_putc(s, '.');
    for (i = 0; i < n; ++i) {
        if (i) ks_putc(s, ';');
        ks_puts(s, h->id[BCF_DT_ID][dec_int(q + i * type_bytes[type], type)].key);
    }
    p = q + n * type_bytes[type];
    ks_putc(s, '\t');                                   /* INFO */
    if (v->n_info == 0) ks_putc(s, '.');
    for (i = 0; i < (int)v->n_info; ++i) {
        int key, kt;
        n = dec_size(p, &q, &kt); key = dec_int(q, kt); p = q + type_bytes[kt]; (void)n;
        n = dec_size(p, &q, &type);
        if (i) ks_putc(s, ';');
        ks_puts(s, h->id[BCF_DT_ID][key].key);
        if (n > 0) { ks_putc(s, '='); fmt_array(s, n, type, q); }
        p = q + n * type_bytes[type];
    }
    if (v->n_sample && v->n_fmt) {                      /* FORMAT + samples */
        struct { int id, n, type; const uint8_t *p; } f[16];
        int j, nf = v->n_fmt < 16 ? (int)v->n_fmt : 16;
        p = (const uint8_t*)v->indiv.s;
        for (i = 0; i < nf; ++i) {
            int kt;
            dec_size(p, &q, &kt); f[i].id = dec_int(q, kt); p = q + type_bytes[kt];
            f[i].n = dec_size(p, &q, &f[i].type); f[i].p = q;
            p = q + (size_t)f[i].n * type_bytes[f[i].type] * v->n_sample;
            ks_putc(s, i ? ':' : '\t');
            ks_puts(s, h->id[BCF_DT_ID][f[i].id].key);
        }
        for (j = 0; j < (int)v->n_sample; ++j) {
            ks_putc(s, '\t');
            for (i = 0; i < nf; ++i) {
                const uint8_t *x = f[i].p + (size_t)j * f[i].n * type_bytes[f[i].type];
                if (i) ks_putc(s, ':');
                if (strcmp(h->id[BCF_DT_ID][f[i].id].key, "GT") == 0) {
                    int l;
                    for (l = 0; l < f[i].n && (int8_t)x[l] != INT8_MIN + 1; ++l) {
                        if (l) ks_putc(s, "/|"[x[l] & 1]);
                        if ((int8_t)x[l] >> 1) ks_puti(s, ((int8_t)x[l] >> 1) - 1); else ks_putc(s, '.');
                    }
                    if (l == 0) ks_putc(s, '.');
                } else fmt_array(s, f[i].n, f[i].type, x);
            }
        }
    }
    return 0;
}

void vcf_hdr_write_text(FILE *fp, const bcf_hdr_t *h)
{
    int l = h->l_text;
    while (l && h->text[l - 1] == 0) --l;
    if (l && h->text[l - 1] == '\n') --l;
    fwrite(h->text, 1, (size_t)l, fp);
    fputc('\n', fp);
}

void bcf_hdr_write_stream(bgzw_t *fp, const bcf_hdr_t *h)
{
    bgzw_write(fp, "BCF\2\2", 5);
    bgzw_write(fp, &h->l_text, 4);
    bgzw_write(fp, h->text, (size_t)h->l_text);
}

int bcf_write1_stream(bgzw_t *fp, const bcf1_t *v)
{
    uint32_t x[8];
    x[0] = (uint32_t)v->shared.l + 24; x[1] = (uint32_t)v->indiv.l;
    memcpy(x + 2, v, 16);
    x[6] = (uint32_t)v->n_allele << 16 | v->n_info;
    x[7] = (uint32_t)v->n_fmt << 24 | v->n_sample;
    bgzw_write(fp, x, 32);
    bgzw_write(fp, v->shared.s, v->shared.l);
    return bgzw_write(fp, v->indiv.s, v->indiv.l);
}
