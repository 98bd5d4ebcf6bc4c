/* bed_filter.c -- `-B FILE [-e]`: keep (or drop) the sites that overlap an interval of a BED file.
 * Restates the behaviour of the reference's bedidx.c (bed_read :95-143, bed_overlap :85-93, used at
 * bgt.c:320-325): whitespace-separated tokens, line = chrom [beg [end]] + ignored rest; a single number N
 * means the 1-based position N, i.e. [N-1, N); intervals with beg < 0 or end <= beg are dropped; a site
 * [pos, pos+rlen) overlaps an interval [b, e) iff e > pos and b < pos + rlen.  The reference finds the
 * candidates through a 8 kb linear index; here every chromosome keeps its intervals sorted by start with a
 * running maximum of the ends, and a binary search bounds the scan -- same answers. */
#include <ctype.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

typedef struct { char *chr; int n, m; int32_t *beg, *end, *maxend; } bedchr_t;
typedef struct { int n, m; bedchr_t *c; } bedset_t;

static bedchr_t *bed_chr(bedset_t *h, const char *name, int add)
{
    int i;
    for (i = 0; i < h->n; ++i) if (strcmp(h->c[i].chr, name) == 0) return &h->c[i];
    if (!add) return NULL;
    if (h->n == h->m) { h->m = h->m ? h->m << 1 : 8; h->c = (bedchr_t*)realloc(h->c, (size_t)h->m * sizeof(bedchr_t)); }
    memset(&h->c[h->n], 0, sizeof(bedchr_t));
    h->c[h->n].chr = strdup(name);
    return &h->c[h->n++];
}

/* next whitespace-delimited token of the stream; *delim = the character that ended it (-1 at end of file) */
static int next_token(gzFile fp, char *buf, int cap, int *delim)
{
    int c, l = 0;
    while ((c = gzgetc(fp)) != -1 && !isspace(c)) if (l < cap - 1) buf[l++] = (char)c;
    buf[l] = 0;
    *delim = c;
    return (c == -1 && l == 0) ? -1 : l;
}

static int cmp_iv(const void *a, const void *b)
{
    const int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
    return x < y ? -1 : x > y;
}

void *bed_read(const char *fn)
{
    gzFile fp = strcmp(fn, "-") ? gzopen(fn, "r") : gzdopen(0, "r");
    bedset_t *h;
    char tok[1024];
    int delim, i, j;
    if (fp == NULL) return NULL;
    h = (bedset_t*)calloc(1, sizeof(*h));
    while (next_token(fp, tok, sizeof(tok), &delim) >= 0) {           /* the chromosome name */
        int beg = -1, end = -1;
        bedchr_t *p = bed_chr(h, tok, 1);
        if (delim != '\n' && delim != -1) {
            if (next_token(fp, tok, sizeof(tok), &delim) > 0 && isdigit((unsigned char)tok[0])) {
                beg = atoi(tok);
                if (delim != '\n' && delim != -1 && next_token(fp, tok, sizeof(tok), &delim) > 0 && isdigit((unsigned char)tok[0])) {
                    end = atoi(tok);
                    if (end < beg) end = -1;
                }
            }
        }
        while (delim != '\n' && delim != -1) delim = gzgetc(fp);      /* the rest of the line */
        if (end < 0 && beg > 0) { end = beg; beg = beg - 1; }         /* one column: a 1-based position */
        if (beg >= 0 && end > beg) {
            if (p->n == p->m) {
                p->m = p->m ? p->m << 1 : 4;
                p->beg = (int32_t*)realloc(p->beg, (size_t)p->m * 4);
                p->end = (int32_t*)realloc(p->end, (size_t)p->m * 4);
            }
            p->beg[p->n] = beg; p->end[p->n++] = end;
        }
    }
    gzclose(fp);
    for (i = 0; i < h->n; ++i) {                                       /* sort by start, running maximum of ends */
        bedchr_t *p = &h->c[i];
        int64_t *key = (int64_t*)malloc((size_t)(p->n ? p->n : 1) * 8);
        for (j = 0; j < p->n; ++j) key[j] = (int64_t)p->beg[j] << 32 | (uint32_t)p->end[j];
        qsort(key, (size_t)p->n, 8, cmp_iv);
        p->maxend = (int32_t*)malloc((size_t)(p->n ? p->n : 1) * 4);
        for (j = 0; j < p->n; ++j) {
            p->beg[j] = (int32_t)(key[j] >> 32); p->end[j] = (int32_t)(uint32_t)key[j];
            p->maxend[j] = j && p->maxend[j - 1] > p->end[j] ? p->maxend[j - 1] : p->end[j];
        }
        free(key);
    }
    return h;
}

int bed_overlap(const void *_h, const char *chr, int beg, int end)
{
    const bedset_t *h = (const bedset_t*)_h;
    const bedchr_t *p;
    int lo, hi;
    if (h == NULL || (p = bed_chr((bedset_t*)h, chr, 0)) == NULL || p->n == 0) return 0;
    /* intervals 0..hi-1 start before `end`; one of them overlaps iff the largest end among them exceeds `beg` */
    lo = 0; hi = p->n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (p->beg[mid] < end) lo = mid + 1; else hi = mid; }
    return lo > 0 && p->maxend[lo - 1] > beg;
}

void bed_destroy(void *_h)
{
    bedset_t *h = (bedset_t*)_h;
    int i;
    if (h == NULL) return;
    for (i = 0; i < h->n; ++i) { free(h->c[i].chr); free(h->c[i].beg); free(h->c[i].end); free(h->c[i].maxend); }
    free(h->c); free(h);
}
