/* csi_writer.c -- see csi_writer.h.  Written from the CSI specification (bins of 2^(min_shift+3k) bases, bin number =
 * offset of its level + position >> shift of its level). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bgzf_io.h"
#include "csi_writer.h"

#define REC_SHIFT 10

typedef struct { uint32_t bin; uint64_t beg, end; } chunk_t;
typedef struct {
    chunk_t *chunks; int64_t n_chunks, m_chunks;
    uint64_t *lin; int64_t n_lin, m_lin;               /* first record offset per 1 << min_shift window */
    uint32_t cur_bin; uint64_t cur_beg, cur_end; int open;
    uint64_t off_beg, off_end, n_rec;
} ref_t;
struct csi_writer_s {
    int n_ref, min_shift, depth;
    ref_t *ref;
    uint64_t *ridx; int64_t n_ridx, m_ridx, n_rec;
    uint64_t first_offset;
};

static int reg2bin(int64_t beg, int64_t end, int min_shift, int depth)
{
    int l, s = min_shift, t = ((1 << depth * 3) - 1) / 7;
    for (--end, l = depth; l > 0; --l, s += 3, t -= 1 << l * 3)
        if (beg >> s == end >> s) return t + (int)(beg >> s);
    return 0;
}
static int64_t bin_first_window(int bin, int depth)                 /* leftmost leaf window under a bin */
{
    int l = 0, b = bin;
    for (; b; ++l, b = (b - 1) >> 3) {}
    return (int64_t)(bin - ((1 << l * 3) - 1) / 7) << (depth - l) * 3;
}

csi_writer_t *csi_writer_init(int n_ref, int min_shift, int depth, uint64_t first_offset)
{
    csi_writer_t *w = (csi_writer_t*)calloc(1, sizeof(*w));
    w->n_ref = n_ref; w->min_shift = min_shift; w->depth = depth; w->first_offset = first_offset;
    w->ref = (ref_t*)calloc((size_t)(n_ref > 0 ? n_ref : 1), sizeof(ref_t));
    return w;
}

static void close_chunk(ref_t *r)
{
    if (!r->open) return;
    if (r->n_chunks == r->m_chunks) { r->m_chunks = r->m_chunks ? r->m_chunks * 2 : 256; r->chunks = (chunk_t*)realloc(r->chunks, (size_t)r->m_chunks * sizeof(chunk_t)); }
    r->chunks[r->n_chunks].bin = r->cur_bin; r->chunks[r->n_chunks].beg = r->cur_beg; r->chunks[r->n_chunks++].end = r->cur_end;
    r->open = 0;
}

void csi_writer_push(csi_writer_t *w, int rid, int64_t beg, int64_t end, uint64_t off0, uint64_t off1)
{
    ref_t *r;
    int64_t win, we;
    uint32_t bin;
    if ((w->n_rec & ((1 << REC_SHIFT) - 1)) == 0) {
        if (w->n_ridx == w->m_ridx) { w->m_ridx = w->m_ridx ? w->m_ridx * 2 : 1024; w->ridx = (uint64_t*)realloc(w->ridx, (size_t)w->m_ridx * 8); }
        w->ridx[w->n_ridx++] = off0;
    }
    ++w->n_rec;
    if (rid < 0 || rid >= w->n_ref) return;
    if (end <= beg) end = beg + 1;
    r = &w->ref[rid];
    if (r->n_rec++ == 0) r->off_beg = off0;
    r->off_end = off1;
    win = beg >> w->min_shift; we = (end - 1) >> w->min_shift;
    if (we >= r->m_lin) {
        const int64_t old = r->m_lin;
        r->m_lin = we + 1 > 2 * old ? we + 1 : 2 * old;
        r->lin = (uint64_t*)realloc(r->lin, (size_t)r->m_lin * 8);
        memset(r->lin + old, 0xff, (size_t)(r->m_lin - old) * 8);
    }
    for (; win <= we; ++win) if (r->lin[win] == (uint64_t)-1) r->lin[win] = off0;
    if (we + 1 > r->n_lin) r->n_lin = we + 1;
    bin = (uint32_t)reg2bin(beg, end, w->min_shift, w->depth);
    if (!r->open || bin != r->cur_bin) { close_chunk(r); r->open = 1; r->cur_bin = bin; r->cur_beg = off0; }
    r->cur_end = off1;
}

static int cmp_chunk(const void *a, const void *b)
{
    const chunk_t *x = (const chunk_t*)a, *y = (const chunk_t*)b;
    return x->bin != y->bin ? (x->bin < y->bin ? -1 : 1) : (x->beg < y->beg ? -1 : x->beg > y->beg);
}

int csi_writer_save(csi_writer_t *w, const char *path)
{
    FILE *fp = fopen(path, "wb");
    bgzw_t *bz;
    const int n_bins = ((1 << (3 * w->depth + 3)) - 1) / 7;
    int32_t x[3], n_ref = w->n_ref, rec_shift = REC_SHIFT, n_r = (int32_t)w->n_ridx;
    uint64_t zero = 0, n_rec = (uint64_t)w->n_rec;
    int k;
    if (fp == NULL) return -1;
    bz = bgzw_open(fp, -1);
    x[0] = w->min_shift; x[1] = w->depth; x[2] = 0;
    bgzw_write(bz, "CSI\1", 4); bgzw_write(bz, x, 12); bgzw_write(bz, &n_ref, 4);
    for (k = 0; k < w->n_ref; ++k) {
        ref_t *r = &w->ref[k];
        int32_t n_bin = 0;
        int64_t i, j;
        uint64_t prev = r->off_beg;
        close_chunk(r);
        for (i = 0; i < r->n_lin; ++i) { if (r->lin[i] == (uint64_t)-1) r->lin[i] = prev; else prev = r->lin[i]; }   /* fill gaps */
        qsort(r->chunks, (size_t)r->n_chunks, sizeof(chunk_t), cmp_chunk);
        for (i = 0; i < r->n_chunks; ++i) if (i == 0 || r->chunks[i].bin != r->chunks[i - 1].bin) ++n_bin;
        n_bin += r->n_rec > 0;                                                   /* + the statistics pseudo-bin */
        bgzw_write(bz, &n_bin, 4);
        for (i = 0; i < r->n_chunks; i = j) {
            const int64_t fw = bin_first_window((int)r->chunks[i].bin, w->depth);
            const uint64_t loff = fw < r->n_lin ? r->lin[fw] : r->off_end;
            int32_t nc;
            for (j = i; j < r->n_chunks && r->chunks[j].bin == r->chunks[i].bin; ++j) {}
            nc = (int32_t)(j - i);
            bgzw_write(bz, &r->chunks[i].bin, 4); bgzw_write(bz, &loff, 8); bgzw_write(bz, &nc, 4);
            for (; i < j; ++i) { bgzw_write(bz, &r->chunks[i].beg, 8); bgzw_write(bz, &r->chunks[i].end, 8); }
        }
        if (r->n_rec > 0) {                                                      /* pseudo-bin: file span, #records */
            uint32_t pb = (uint32_t)n_bins + 1; int32_t two = 2;
            bgzw_write(bz, &pb, 4); bgzw_write(bz, &zero, 8); bgzw_write(bz, &two, 4);
            bgzw_write(bz, &r->off_beg, 8); bgzw_write(bz, &r->off_end, 8);
            bgzw_write(bz, &r->n_rec, 8); bgzw_write(bz, &zero, 8);
        }
    }
    bgzw_write(bz, &zero, 8);                                                    /* records without coordinates */
    bgzw_write(bz, "RNI\1", 4); bgzw_write(bz, &n_rec, 8); bgzw_write(bz, &rec_shift, 4);
    bgzw_write(bz, &n_r, 4); bgzw_write(bz, w->ridx, (size_t)w->n_ridx * 8);
    bgzw_close(bz);
    fclose(fp);
    return 0;
}

void csi_writer_destroy(csi_writer_t *w)
{
    int k;
    if (!w) return;
    for (k = 0; k < w->n_ref; ++k) { free(w->ref[k].chunks); free(w->ref[k].lin); }
    free(w->ref); free(w->ridx); free(w);
}
