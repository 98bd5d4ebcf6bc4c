/* metadata.c -- see metadata.h; behaviour of reference fmf.c:19-100 (parser) and :140-156 (row test). */
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include "metadata.h"

static int intern(char ***tab, int *n, int *m, const char *s)
{
    int i;
    for (i = 0; i < *n; ++i) if (strcmp((*tab)[i], s) == 0) return i;   /* few distinct keys/values */
    if (*n == *m) { *m = *m ? *m * 2 : 8; *tab = (char**)realloc(*tab, (size_t)*m * sizeof(char*)); }
    (*tab)[*n] = strdup(s);
    return (*n)++;
}

static void add_row(fmf_t *f, char *line)
{
    char *p = line, *field;
    int col = 0, n_tab = 0;
    fmf1_t *u;
    for (p = line; *p; ++p) n_tab += *p == '\t';
    if (f->n_rows == f->m_rows) { f->m_rows = f->m_rows ? f->m_rows * 2 : 16; f->rows = (fmf1_t*)realloc(f->rows, (size_t)f->m_rows * sizeof(fmf1_t)); }
    u = &f->rows[f->n_rows++];
    memset(u, 0, sizeof(*u));
    u->m_meta = n_tab; u->meta = (fmf_meta_t*)calloc((size_t)(n_tab ? n_tab : 1), sizeof(fmf_meta_t));
    for (field = p = line;; ++p) {
        if (*p == 0 || *p == '\t') {
            const int last = *p == 0;
            *p = 0;
            if (col == 0) u->name = strdup(field);
            else {
                fmf_meta_t *m = &u->meta[u->n_meta++];
                char *c = strchr(field, ':');
                if (c) *c = 0;
                m->key = (uint32_t)intern(&f->keys, &f->n_keys, &f->m_keys, field);
                m->v.i = 0;
                if (c && p - c >= 3) {                                  /* "key:T:value" */
                    if (c[1] == 'i') { m->type = FMF_INT; m->v.i = (int32_t)strtol(c + 3, NULL, 0); }
                    else if (c[1] == 'f') { m->type = FMF_REAL; m->v.r = (float)strtod(c + 3, NULL); }
                    else { m->type = FMF_STR; m->v.s = (uint32_t)intern(&f->vals, &f->n_vals, &f->m_vals, c + 3); }
                } else m->type = FMF_FLAG;
            }
            ++col; field = p + 1;
            if (last) break;
        }
    }
}

fmf_t *fmf_read(const char *fn)
{
    gzFile fp = gzopen(fn, "r");
    fmf_t *f;
    size_t cap = 1 << 16, len = 0;
    char *line;
    int c;
    if (!fp) return NULL;
    f = (fmf_t*)calloc(1, sizeof(*f));
    line = (char*)malloc(cap);
    while ((c = gzgetc(fp)) != -1) {
        if (c == '\n') { line[len] = 0; if (len) add_row(f, line); len = 0; continue; }
        if (len + 2 > cap) { cap *= 2; line = (char*)realloc(line, cap); }
        line[len++] = (char)c;
    }
    if (len) { line[len] = 0; add_row(f, line); }
    free(line);
    gzclose(fp);
    return f;
}

void fmf_destroy(fmf_t *f)
{
    int i;
    if (!f) return;
    for (i = 0; i < f->n_keys; ++i) free(f->keys[i]);
    for (i = 0; i < f->n_vals; ++i) free(f->vals[i]);
    for (i = 0; i < f->n_rows; ++i) { free(f->rows[i].name); free(f->rows[i].meta); }
    free(f->rows); free(f->keys); free(f->vals); free(f);
}

/* bind the row's metadata as variables, evaluate; an unbound variable (error) means "no" (ref fmf.c:155) */
int fmf_test(const fmf_t *f, int row, kexpr_t *ke)
{
    const fmf1_t *u;
    int i, err, yes;
    if (row >= f->n_rows) return 0;
    u = &f->rows[row];
    ke_unset(ke);
    for (i = 0; i < u->n_meta; ++i) {
        const fmf_meta_t *m = &u->meta[i];
        ke_set_str(ke, "_ROW_", u->name);
        if (m->type == FMF_STR) ke_set_str(ke, f->keys[m->key], f->vals[m->v.s]);
        else if (m->type == FMF_INT) ke_set_int(ke, f->keys[m->key], m->v.i);
        else if (m->type == FMF_REAL) ke_set_int(ke, f->keys[m->key], (int64_t)m->v.r);   /* sic: ref fmf.c:152 binds reals as ints */
    }
    yes = !!ke_eval_int(ke, &err);
    return !(err || !yes);
}
