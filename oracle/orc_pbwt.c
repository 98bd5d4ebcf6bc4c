/*
 * orc_pbwt.c -- CPU ORACLE (test infrastructure only; see orc.h): PBWT run-length code, full-row
 * codec, rank-tracking subset decoder and the PBF container, restated from the behaviour of the
 * reference's pbwt.c.  Written from the format/algorithm description, not
 * transcribed: data structures and control flow are this repo's own, results are bit-identical
 * (pinned by tests/test_oracle_golden.py against the compiled reference).
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "orc.h"

/* ------------------------------------------------------------------------------------------------
 * Run-length byte code.  ref pbwt.c:12-21 (the 128-entry table) is exactly
 *     code = byte>>1 ; len = (code & 15) << (4 * (code >> 4))
 * bit = byte & 1.  A zero byte terminates a row (pbwt.c:73, :81 loop on *q).
 * ---------------------------------------------------------------------------------------------- */
uint32_t orc_rle_len(uint8_t byte)
{
    uint32_t code = byte >> 1;
    return (code & 15u) << (4u * (code >> 4));
}

/* ref pbwt.c:24-36: a run shorter than 16 is one byte; otherwise one byte per non-zero hex digit of
 * the length, most significant digit first, all carrying the run's bit. */
int orc_rle_put_run(uint8_t *dst, uint32_t len, int bit)
{
    int n = 0, digit;
    if (len < 16) { dst[0] = (uint8_t)(len << 1 | (bit & 1)); return 1; }
    for (digit = 7; digit >= 0; --digit) {
        uint32_t d = (len >> (4 * digit)) & 15u;
        if (d) dst[n++] = (uint8_t)(((uint32_t)digit * 16u + d) << 1 | (bit & 1));
    }
    return n;
}

/* ref pbwt.c:39-50: maximal runs of equal bits, left to right; a 0 terminator is stored after the
 * last byte but not counted.  dst may alias bits (the encoder never overtakes the reader). */
int orc_rle_encode(int m, const uint8_t *bits, uint8_t *dst)
{
    int j = 0, n = 0;
    while (j < m) {
        uint8_t v = bits[j];
        int e = j + 1;
        while (e < m && bits[e] == v) ++e;
        n += orc_rle_put_run(dst + n, (uint32_t)(e - j), v);
        j = e;
    }
    dst[n] = 0;
    return n;
}

/* ref pbwt.c:73-74 / :133-134 */
int64_t orc_rle_count_ones(const uint8_t *rle, int n)
{
    int64_t ones = 0;
    int k;
    for (k = 0; k < n && rle[k]; ++k)
        if (rle[k] & 1) ones += orc_rle_len(rle[k]);
    return ones;
}

/* ------------------------------------------------------------------------------------------------
 * Full-row codec.  ref pbwt.c:92-105 (state: two int32[m] permutations + m+1 byte row, identity
 * start), :107-119 (swap then core), :57-66 (encode core), :69-90 (decode core).
 * ---------------------------------------------------------------------------------------------- */
orc_codec_t *orc_codec_new(int m)
{
    orc_codec_t *c = (orc_codec_t*)calloc(1, sizeof(*c));
    int j;
    c->m = m;
    c->perm = (int32_t*)malloc((size_t)m * 4);
    c->prev = (int32_t*)malloc((size_t)m * 4);
    c->bits = (uint8_t*)calloc((size_t)m + 1, 1);
    for (j = 0; j < m; ++j) c->perm[j] = j, c->prev[j] = 0;
    return c;
}

void orc_codec_free(orc_codec_t *c)
{
    if (!c) return;
    free(c->perm); free(c->prev); free(c->bits); free(c);
}

static void codec_flip(orc_codec_t *c)
{
    int32_t *t = c->perm; c->perm = c->prev; c->prev = t;
}

/* B_k (RLE) + S_{k-1} -> A_k and S_k.  A_k[S_{k-1}[j]] = B_k[j]; S_k = S_{k-1} stably partitioned by
 * B_k, zeros first.  Constant rows leave the permutation untouched (ref pbwt.c:75-77). */
void orc_codec_decode(orc_codec_t *c, const uint8_t *rle, int n)
{
    const int m = c->m;
    int64_t ones = orc_rle_count_ones(rle, n);
    int k, at = 0;
    int32_t *zero_dst, *one_dst;
    codec_flip(c);
    if (ones == 0 || ones == m) {
        memcpy(c->perm, c->prev, (size_t)m * 4);
        memset(c->bits, ones == m, (size_t)m);
        return;
    }
    zero_dst = c->perm;
    one_dst  = c->perm + (m - ones);
    memset(c->bits, 0, (size_t)m);
    for (k = 0; k < n && rle[k]; ++k) {
        const uint32_t len = orc_rle_len(rle[k]);
        const int32_t *src = c->prev + at;
        uint32_t i;
        if (rle[k] & 1) {
            for (i = 0; i < len; ++i) { c->bits[src[i]] = 1; one_dst[i] = src[i]; }
            one_dst += len;
        } else {
            memcpy(zero_dst, src, (size_t)len * 4);
            zero_dst += len;
        }
        at += (int)len;
    }
}

/* A_k + S_{k-1} -> B_k (RLE into dst, m+1 bytes) and S_k.  ref pbwt.c:57-66. */
int orc_codec_encode(orc_codec_t *c, const uint8_t *a, uint8_t *dst)
{
    const int m = c->m;
    int j, ones = 0;
    int32_t *zero_dst, *one_dst;
    codec_flip(c);
    for (j = 0; j < m; ++j) ones += (c->bits[j] = (a[c->prev[j]] != 0));
    zero_dst = c->perm;
    one_dst  = c->perm + (m - ones);
    for (j = 0; j < m; ++j) {
        if (c->bits[j]) *one_dst++ = c->prev[j];
        else *zero_dst++ = c->prev[j];
    }
    return orc_rle_encode(m, c->bits, dst);
}

/* ------------------------------------------------------------------------------------------------
 * Subset decoding by rank tracking (LF-mapping).  ref pbwt.c:340-347 (initial ranks from a
 * permutation: invert, look up, order by rank) and :129-170 (one row).
 * t[] is kept ordered by rank; t[x].slot is the output position of the tracked column.
 * ---------------------------------------------------------------------------------------------- */
void orc_track_init(int m, const int32_t *perm, int n_sub, const int32_t *cols, orc_track_t *t)
{
    /* rank_of[col] then a counting placement (ranks are distinct, so "sort by rank" is unambiguous) */
    int32_t *rank_of = (int32_t*)malloc((size_t)m * 4);
    int32_t *slot_at = (int32_t*)malloc((size_t)m * 4);
    int j, x = 0;
    for (j = 0; j < m; ++j) rank_of[perm[j]] = j, slot_at[j] = -1;
    for (j = 0; j < n_sub; ++j) slot_at[rank_of[cols[j]]] = j;
    for (j = 0; j < m; ++j)
        if (slot_at[j] >= 0) t[x].rank = (uint32_t)j, t[x].slot = (uint32_t)slot_at[j], ++x;
    free(rank_of); free(slot_at);
}

void orc_track_decode(int m, int n_sub, orc_track_t *t, const uint8_t *rle, int n, uint8_t *a)
{
    int64_t ones = orc_rle_count_ones(rle, n);
    orc_track_t *hold;
    int k, x = 0, nz = 0, no = 0;
    uint32_t zeros_seen = 0, ones_seen = 0;
    if (ones == 0 || ones == m) {            /* ref pbwt.c:135-138: ranks stay as they are */
        memset(a, ones == m, (size_t)n_sub);
        return;
    }
    memset(a, 0, (size_t)n_sub);
    hold = (orc_track_t*)malloc((size_t)n_sub * sizeof(*hold));
    for (k = 0; k < n && rle[k] && x < n_sub; ++k) {
        const uint32_t len = orc_rle_len(rle[k]);
        const uint32_t beg = zeros_seen + ones_seen;
        const int bit = rle[k] & 1;
        /* new rank of position p inside this run: zeros go to [0,m-ones), ones after them, each
         * keeping their relative order (ref pbwt.c:148-153) */
        const uint32_t base = bit ? (uint32_t)(m - ones) + ones_seen : zeros_seen;
        while (x < n_sub && t[x].rank >= beg && t[x].rank < beg + len) {
            orc_track_t e = t[x++];
            e.rank = base + (e.rank - beg);
            if (bit) { a[e.slot] = 1; hold[no++] = e; }
            else t[nz++] = e;                /* nz <= x-1 always: in-place compaction is safe */
        }
        if (bit) ones_seen += len; else zeros_seen += len;
    }
    /* entries never reached only exist if the row is malformed; the reference leaves them behind the
     * zeros too (pbwt.c:145 loop ends at p==end or terminator) -- keep them in place */
    while (x < n_sub) t[nz++] = t[x++];
    memcpy(t + nz, hold, (size_t)no * sizeof(*hold));
    free(hold);
}

/* ------------------------------------------------------------------------------------------------
 * PBF container over a memory image.  Layout (ref pbwt.c:199-219 header, :288-311 records,
 * :264-277 footer):  "PBF\1" m g shift | { ['S' g*m int32] 'B' g*(int32 len, bytes) }* |
 * 'I' int64 n, int32 n_idx, uint64 idx[n_idx], uint64 offset_of_'I'
 * ---------------------------------------------------------------------------------------------- */
struct orc_pbf_s {
    const uint8_t *buf; size_t len, pos;
    int32_t m, g, shift;
    int64_t n, next;                 /* rows in file; next row to be read (ref "k") */
    int32_t n_idx; uint64_t *idx;
    orc_codec_t **codec;
    const uint8_t **ret;
    int n_sub; int32_t *cols; orc_track_t **track; uint8_t **sub_bits;
    uint8_t *row;                    /* m+1 scratch for one RLE string + terminator */
};

static int take(orc_pbf_t *p, void *dst, size_t n)
{
    if (p->pos + n > p->len) return -1;
    memcpy(dst, p->buf + p->pos, n);
    p->pos += n;
    return 0;
}

orc_pbf_t *orc_pbf_open(const uint8_t *buf, size_t len)       /* ref pbwt.c:221-262 */
{
    orc_pbf_t *p;
    int32_t hdr[3];
    int i;
    if (len < 16 || memcmp(buf, "PBF\1", 4) != 0) return NULL;
    p = (orc_pbf_t*)calloc(1, sizeof(*p));
    p->buf = buf; p->len = len;
    memcpy(hdr, buf + 4, 12);
    p->m = hdr[0]; p->g = hdr[1]; p->shift = hdr[2];
    p->codec = (orc_codec_t**)calloc((size_t)p->g, sizeof(void*));
    p->ret = (const uint8_t**)calloc((size_t)p->g, sizeof(void*));
    p->track = (orc_track_t**)calloc((size_t)p->g, sizeof(void*));
    p->sub_bits = (uint8_t**)calloc((size_t)p->g, sizeof(void*));
    for (i = 0; i < p->g; ++i) { p->codec[i] = orc_codec_new(p->m); p->ret[i] = p->codec[i]->bits; }
    p->row = (uint8_t*)malloc((size_t)p->m + 1);
    if (len >= 24) {                                          /* footer: last 8 bytes point at 'I' */
        uint64_t off;
        memcpy(&off, buf + len - 8, 8);
        if (off + 13 <= len && buf[off] == 'I') {
            memcpy(&p->n, buf + off + 1, 8);
            memcpy(&p->n_idx, buf + off + 9, 4);
            p->idx = (uint64_t*)malloc((size_t)(p->n_idx > 0 ? p->n_idx : 1) * 8);
            memcpy(p->idx, buf + off + 13, (size_t)p->n_idx * 8);
        }
    }
    p->pos = 16;
    return p;
}

void orc_pbf_close(orc_pbf_t *p)
{
    int i;
    if (!p) return;
    for (i = 0; i < p->g; ++i) { orc_codec_free(p->codec[i]); free(p->track[i]); free(p->sub_bits[i]); }
    free(p->codec); free(p->ret); free(p->track); free(p->sub_bits);
    free(p->idx); free(p->cols); free(p->row); free(p);
}

int orc_pbf_m(const orc_pbf_t *p) { return p->m; }
int orc_pbf_g(const orc_pbf_t *p) { return p->g; }
int orc_pbf_shift(const orc_pbf_t *p) { return p->shift; }
int64_t orc_pbf_n(const orc_pbf_t *p) { return p->n; }
int64_t orc_pbf_tell(const orc_pbf_t *p) { return p->next; }
const int32_t *orc_pbf_perm(const orc_pbf_t *p, int plane) { return p->codec[plane]->perm; }

static int subset_on(const orc_pbf_t *p) { return p->n_sub > 0 && p->n_sub < p->m; }
int orc_pbf_subset_width(const orc_pbf_t *p) { return subset_on(p) ? p->n_sub : p->m; }

/* ref pbwt.c:374-388: n_sub<=0 or >=m means "decode everything"; otherwise remember the column
 * list and derive tracked ranks from the CURRENT permutation of every plane. */
int orc_pbf_subset(orc_pbf_t *p, int n_sub, const int32_t *cols)
{
    int g;
    if (n_sub <= 0 || n_sub >= p->m || cols == NULL) n_sub = 0;
    p->n_sub = n_sub;
    for (g = 0; g < p->g; ++g) p->ret[g] = p->codec[g]->bits;
    if (n_sub == 0) return 0;
    p->cols = (int32_t*)realloc(p->cols, (size_t)n_sub * 4);
    memcpy(p->cols, cols, (size_t)n_sub * 4);
    for (g = 0; g < p->g; ++g) {
        p->track[g] = (orc_track_t*)realloc(p->track[g], (size_t)n_sub * sizeof(orc_track_t));
        p->sub_bits[g] = (uint8_t*)realloc(p->sub_bits[g], (size_t)n_sub);
        orc_track_init(p->m, p->codec[g]->perm, n_sub, p->cols, p->track[g]);
        p->ret[g] = p->sub_bits[g];
    }
    return 0;
}

/* ref pbwt.c:313-337: an optional 'S' record refreshes the full permutations (NOT the tracked
 * ranks), then a 'B' record holds one RLE string per plane; anything else ends the stream. */
const uint8_t **orc_pbf_read(orc_pbf_t *p)
{
    uint8_t tag;
    int g;
    if (take(p, &tag, 1) < 0) return NULL;
    if (tag == 'S') {
        for (g = 0; g < p->g; ++g)
            if (take(p, p->codec[g]->perm, (size_t)p->m * 4) < 0) return NULL;
        if (take(p, &tag, 1) < 0) return NULL;
    }
    if (tag != 'B') return NULL;
    for (g = 0; g < p->g; ++g) {
        int32_t l;
        if (take(p, &l, 4) < 0 || l < 0 || l > p->m) return NULL;
        if (take(p, p->row, (size_t)l) < 0) return NULL;
        p->row[l] = 0;
        if (subset_on(p)) orc_track_decode(p->m, p->n_sub, p->track[g], p->row, l, p->sub_bits[g]);
        else orc_codec_decode(p->codec[g], p->row, l);
    }
    for (g = 0; g < p->g; ++g) p->ret[g] = subset_on(p) ? p->sub_bits[g] : p->codec[g]->bits;
    ++p->next;
    return p->ret;
}

/* ref pbwt.c:349-372: same row = no-op; up to 1<<shift rows ahead = decode forward through the
 * intermediate rows; otherwise restart from the checkpoint of the target's block (re-deriving the
 * tracked ranks) and decode forward inside the block. */
int orc_pbf_seek(orc_pbf_t *p, int64_t row)
{
    int64_t in_block, i;
    int g;
    if (row == p->next) return 0;
    if (row > p->next && row - p->next <= ((int64_t)1 << p->shift)) {
        while (p->next < row) if (orc_pbf_read(p) == NULL) return -1;
        return 0;
    }
    if (p->idx == NULL || row < 0 || row >= p->n) return -1;
    p->pos = (size_t)p->idx[row >> p->shift];
    if (p->pos >= p->len || p->buf[p->pos] != 'S') return -2;
    ++p->pos;
    for (g = 0; g < p->g; ++g) {
        if (take(p, p->codec[g]->perm, (size_t)p->m * 4) < 0) return -2;
        if (subset_on(p)) orc_track_init(p->m, p->codec[g]->perm, p->n_sub, p->cols, p->track[g]);
    }
    p->next = row >> p->shift << p->shift;
    in_block = row - p->next;
    for (i = 0; i < in_block; ++i) if (orc_pbf_read(p) == NULL) return -1;
    return 0;
}

/* ---------------- writer ---------------- */
struct orc_pbw_s {
    int32_t m, g, shift;
    int64_t n;
    orc_codec_t **codec;
    uint8_t *out; size_t len, cap;
    uint64_t *idx; int32_t n_idx, cap_idx;
    uint8_t *row;
};

static void put(orc_pbw_t *w, const void *src, size_t n)
{
    if (w->len + n > w->cap) {
        while (w->len + n > w->cap) w->cap = w->cap ? w->cap * 2 : 4096;
        w->out = (uint8_t*)realloc(w->out, w->cap);
    }
    memcpy(w->out + w->len, src, n);
    w->len += n;
}

orc_pbw_t *orc_pbw_new(int m, int g, int shift)               /* ref pbwt.c:199-219 */
{
    orc_pbw_t *w = (orc_pbw_t*)calloc(1, sizeof(*w));
    int32_t hdr[3];
    int i;
    w->m = m; w->g = g; w->shift = shift;
    w->codec = (orc_codec_t**)calloc((size_t)g, sizeof(void*));
    for (i = 0; i < g; ++i) w->codec[i] = orc_codec_new(m);
    w->row = (uint8_t*)malloc((size_t)m + 1);
    hdr[0] = m; hdr[1] = g; hdr[2] = shift;
    put(w, "PBF\1", 4); put(w, hdr, 12);
    return w;
}

int orc_pbw_row(orc_pbw_t *w, uint8_t *const *planes)         /* ref pbwt.c:288-311 */
{
    int g;
    if ((w->n & (((int64_t)1 << w->shift) - 1)) == 0) {
        uint64_t at = w->len;
        if (w->n_idx == w->cap_idx) {
            w->cap_idx = w->cap_idx ? w->cap_idx * 2 : 8;
            w->idx = (uint64_t*)realloc(w->idx, (size_t)w->cap_idx * 8);
        }
        w->idx[w->n_idx++] = at;
        put(w, "S", 1);
        for (g = 0; g < w->g; ++g) put(w, w->codec[g]->perm, (size_t)w->m * 4);
    }
    put(w, "B", 1);
    for (g = 0; g < w->g; ++g) {
        int32_t l = orc_codec_encode(w->codec[g], planes[g], w->row);
        put(w, &l, 4); put(w, w->row, (size_t)l);
    }
    ++w->n;
    return 0;
}

size_t orc_pbw_finish(orc_pbw_t *w, uint8_t **out)            /* ref pbwt.c:264-277 */
{
    uint64_t off = w->len;
    size_t len;
    int g;
    put(w, "I", 1); put(w, &w->n, 8); put(w, &w->n_idx, 4);
    put(w, w->idx, (size_t)w->n_idx * 8); put(w, &off, 8);
    *out = w->out; len = w->len;
    for (g = 0; g < w->g; ++g) orc_codec_free(w->codec[g]);
    free(w->codec); free(w->idx); free(w->row); free(w);
    return len;
}
