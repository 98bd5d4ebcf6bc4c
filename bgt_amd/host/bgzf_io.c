/* bgzf_io.c -- BGZF = a series of gzip members of at most 64 KiB, each carrying its own compressed size in
 * a "BC" extra field (SAM/BAM spec 4.1; the reference's bgzf.c implements the same container). */
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <zlib.h>
#include "bgzf_io.h"

#define BLOCK_DATA 0xff00          /* uncompressed bytes per block, as the reference writer uses */
#define BLOCK_MAX  0x10000

struct bgzr_s {
    FILE *fp;
    uint8_t *in, *out;
    size_t out_len, out_pos;
    int eof;
    long blk_coff;                 /* file offset of the block in `out` */
};

bgzr_t *bgzr_open(const char *path)
{
    FILE *fp = fopen(path, "rb");
    bgzr_t *r;
    int c0, c1;
    if (!fp) return NULL;
    c0 = fgetc(fp); c1 = fgetc(fp);
    if (c0 != 0x1f || c1 != 0x8b) { fclose(fp); return NULL; }
    rewind(fp);
    r = (bgzr_t*)calloc(1, sizeof(*r));
    r->fp = fp;
    r->in = (uint8_t*)malloc(BLOCK_MAX);
    r->out = (uint8_t*)malloc(BLOCK_MAX);
    return r;
}

static int next_block(bgzr_t *r)
{
    uint8_t hdr[12], *p;
    unsigned xlen, bsize = 0, i;
    size_t rest;
    z_stream zs;
    r->blk_coff = ftell(r->fp);
    if (fread(hdr, 1, 12, r->fp) != 12) { r->eof = 1; return 0; }
    if (hdr[0] != 0x1f || hdr[1] != 0x8b || !(hdr[3] & 4)) return -1;
    xlen = hdr[10] | hdr[11] << 8;
    if (fread(r->in, 1, xlen, r->fp) != xlen) return -1;
    for (i = 0; i + 4 <= xlen;) {                       /* find the BC subfield */
        unsigned slen = r->in[i + 2] | r->in[i + 3] << 8;
        if (r->in[i] == 'B' && r->in[i + 1] == 'C' && slen == 2) bsize = (r->in[i + 4] | r->in[i + 5] << 8) + 1u;
        i += 4 + slen;
    }
    if (bsize < 12 + xlen + 8) return -1;
    rest = bsize - 12 - xlen;                           /* deflate data + crc32 + isize */
    if (fread(r->in, 1, rest, r->fp) != rest) return -1;
    p = r->in;
    memset(&zs, 0, sizeof(zs));
    zs.next_in = p; zs.avail_in = (uInt)(rest - 8);
    zs.next_out = r->out; zs.avail_out = BLOCK_MAX;
    if (inflateInit2(&zs, -15) != Z_OK) return -1;
    if (inflate(&zs, Z_FINISH) != Z_STREAM_END) { inflateEnd(&zs); return -1; }
    inflateEnd(&zs);
    r->out_len = zs.total_out;
    r->out_pos = 0;
    return 1;
}

long bgzr_read(bgzr_t *r, void *dst, size_t n)
{
    size_t got = 0;
    while (got < n) {
        size_t k;
        if (r->out_pos == r->out_len) {
            int s;
            if (r->eof) break;
            s = next_block(r);
            if (s < 0) return -1;
            if (s == 0) break;
            continue;
        }
        k = r->out_len - r->out_pos;
        if (k > n - got) k = n - got;
        memcpy((uint8_t*)dst + got, r->out + r->out_pos, k);
        r->out_pos += k; got += k;
    }
    return (long)got;
}

/* virtual file offsets as in the BGZF index formats: compressed offset of a block << 16 | offset inside it */
int bgzr_seek(bgzr_t *r, uint64_t voff)
{
    if (fseek(r->fp, (long)(voff >> 16), SEEK_SET) != 0) return -1;
    r->out_len = r->out_pos = 0; r->eof = 0;
    if ((voff & 0xffff) == 0) return 0;                 /* the block is loaded by the next read */
    if (next_block(r) <= 0 || (voff & 0xffff) > r->out_len) return -1;
    r->out_pos = (size_t)(voff & 0xffff);
    return 0;
}

uint64_t bgzr_tell(bgzr_t *r)
{
    if (r->out_len == 0 && r->out_pos == 0) return (uint64_t)ftell(r->fp) << 16;      /* nothing loaded yet */
    return (uint64_t)r->blk_coff << 16 | (uint64_t)r->out_pos;
}

void bgzr_close(bgzr_t *r)
{
    if (!r) return;
    fclose(r->fp); free(r->in); free(r->out); free(r);
}

/* ---------------- writer ---------------- */
/* Blocks are independent gzip members, so a batch of full blocks can be deflated by several threads and
 * written in order: the bytes are the same as from the one-block-at-a-time writer (bgzw_threads). */
#define BATCH_BLOCKS 64
struct bgzw_s {
    FILE *fp;
    int level, n_thr;
    uint8_t *buf, *cbuf;
    size_t fill;
    uint64_t coff;
    uint8_t *pend, *cpend;         /* batch mode: BATCH_BLOCKS raw blocks and their compressed forms */
    uint32_t clen[BATCH_BLOCKS];
    int n_pend;
};

bgzw_t *bgzw_open(FILE *fp, int level)
{
    bgzw_t *w = (bgzw_t*)calloc(1, sizeof(*w));
    w->fp = fp;
    w->level = (level < 0 || level > 9) ? Z_DEFAULT_COMPRESSION : level;
    w->buf = (uint8_t*)malloc(BLOCK_MAX);
    w->cbuf = (uint8_t*)malloc(BLOCK_MAX);
    w->n_thr = 1;
    return w;
}

void bgzw_threads(bgzw_t *w, int n)
{
    if (n > 16) n = 16;
    if (n > 1 && w->pend == NULL) {
        w->pend = (uint8_t*)malloc((size_t)BATCH_BLOCKS * BLOCK_MAX);
        w->cpend = (uint8_t*)malloc((size_t)BATCH_BLOCKS * BLOCK_MAX);
    }
    w->n_thr = n > 1 ? n : 1;
}

/* one BGZF member for n bytes; returns its size or 0 */
static uint32_t deflate_block(const uint8_t *src, uint32_t n, int level, uint8_t *dst)
{
    static const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    z_stream zs;
    uint32_t crc, total;
    memset(&zs, 0, sizeof(zs));
    zs.next_in = (Bytef*)src; zs.avail_in = n;
    zs.next_out = dst + 18; zs.avail_out = BLOCK_MAX - 18 - 8;
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return 0;
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { deflateEnd(&zs); return 0; }
    deflateEnd(&zs);
    total = (uint32_t)zs.total_out + 18 + 8;
    memcpy(dst, head, 16);
    dst[16] = (uint8_t)((total - 1) & 0xff); dst[17] = (uint8_t)((total - 1) >> 8);
    crc = (uint32_t)crc32(crc32(0L, NULL, 0), src, n);
    memcpy(dst + total - 8, &crc, 4);
    memcpy(dst + total - 4, &n, 4);
    return total;
}

static int flush_block(bgzw_t *w)
{
    const uint32_t total = deflate_block(w->buf, (uint32_t)w->fill, w->level, w->cbuf);
    if (total == 0 || fwrite(w->cbuf, 1, total, w->fp) != total) return -1;
    w->coff += total;
    w->fill = 0;
    return 0;
}

typedef struct { bgzw_t *w; int first, step; } batch_job_t;

static void *batch_worker(void *arg)
{
    const batch_job_t *j = (const batch_job_t*)arg;
    int i;
    for (i = j->first; i < j->w->n_pend; i += j->step)
        j->w->clen[i] = deflate_block(j->w->pend + (size_t)i * BLOCK_MAX, BLOCK_DATA, j->w->level, j->w->cpend + (size_t)i * BLOCK_MAX);
    return NULL;
}

static int flush_batch(bgzw_t *w)
{
    pthread_t tid[16];
    batch_job_t job[16];
    int i, n = w->n_thr < w->n_pend ? w->n_thr : w->n_pend, rc = 0;
    if (w->n_pend == 0) return 0;
    for (i = 0; i < n; ++i) { job[i].w = w; job[i].first = i; job[i].step = n; }
    for (i = 1; i < n; ++i) if (pthread_create(&tid[i], NULL, batch_worker, &job[i]) != 0) { batch_worker(&job[i]); tid[i] = 0; }
    batch_worker(&job[0]);
    for (i = 1; i < n; ++i) if (tid[i]) pthread_join(tid[i], NULL);
    for (i = 0; i < w->n_pend; ++i) {
        if (w->clen[i] == 0 || fwrite(w->cpend + (size_t)i * BLOCK_MAX, 1, w->clen[i], w->fp) != w->clen[i]) { rc = -1; break; }
        w->coff += w->clen[i];
    }
    w->n_pend = 0;
    return rc;
}

int bgzw_write(bgzw_t *w, const void *src, size_t n)
{
    const uint8_t *p = (const uint8_t*)src;
    while (n) {
        size_t k = BLOCK_DATA - w->fill;
        if (k > n) k = n;
        memcpy(w->buf + w->fill, p, k);
        w->fill += k; p += k; n -= k;
        if (w->fill == BLOCK_DATA) {
            if (w->n_thr > 1) {                            /* park the full block; deflate a batch at a time */
                memcpy(w->pend + (size_t)w->n_pend * BLOCK_MAX, w->buf, BLOCK_DATA);
                w->fill = 0;
                if (++w->n_pend == BATCH_BLOCKS && flush_batch(w) < 0) return -1;
            } else if (flush_block(w) < 0) return -1;
        }
    }
    return 0;
}

uint64_t bgzw_tell(const bgzw_t *cw)
{
    bgzw_t *w = (bgzw_t*)cw;
    if (w->n_pend) flush_batch(w);                       /* offsets are only known once the parked blocks are written */
    return w->coff << 16 | (uint64_t)w->fill;
}

int bgzw_close(bgzw_t *w)
{
    int rc = 0;
    if (!w) return 0;
    /* the BGZF end-of-file marker is one fixed empty member (SAM spec 4.1.2), whatever the level */
    static const uint8_t eof_marker[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0,
                                           0, 0, 0, 0, 0, 0, 0, 0};
    if (flush_batch(w) < 0) rc = -1;
    if (w->fill && flush_block(w) < 0) rc = -1;
    if (fwrite(eof_marker, 1, 28, w->fp) != 28) rc = -1;
    fflush(w->fp);
    free(w->buf); free(w->cbuf); free(w->pend); free(w->cpend); free(w);
    return rc;
}
