/* pbfview_cli.c -- `bgt pbfview`: the command-line behaviour of the reference's codec-level tool (pbfview.c) over the device
 * codec.  Same options, same bytes on stdout; the program is this repo's own: a job description filled from an option
 * table, one of two row SOURCES (PIM text | a .pbf through bgth_reader_read) pumping into one of two row SINKS (PIM text |
 * a .pbf through bgth_encoder_*).
 *
 *   decode   <in.pbf>             -> PIM text ("PIM1 m g", then one line of m integers per row: sum of plane k << k)
 *            -c COL (repeatable)  -> only these columns, in the order given (pbf_subset, pbwt.c:374-388)
 *            -r ROW, -n COUNT     -> from row ROW on (pbf_seek, pbwt.c:349-372), at most COUNT rows
 *   encode   -S <in.pim> -b       -> PBF on stdout, 'S' records every 1 << shift rows (-s, default 13)
 *   recode   <in.pbf> -b [-c ..]  -> the decoded rows written again
 *
 * Decoding is rank tracking on the device, encoding keeps the PBWT order on the device; the host only parses and prints
 * integers.  Any number of bit planes up to the encoder's eight: every plane is a PBWT of its own (pbwt.c:211-213). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <unistd.h>
#include "../../include/bgt_hip.h"

#define PV_ROWS_PER_WRITE 4096                                       /* rows handed to the encoder at a time */

/* ---- the job ---- */
typedef struct {
    int         text_in, pbf_out, shift;
    int64_t     first_row, max_rows;                                 /* max_rows < 0: all */
    int32_t    *cols;
    int         n_cols, cap_cols;
    const char *input;
} pv_job_t;

typedef struct { char key; const char *arg; const char *help; } pv_opt_t;
static const pv_opt_t PV_OPTS[] = {
    {'S', NULL,  "input is PIM (portable integer matrix format)"},
    {'b', NULL,  "output PBF (positional BWT format)"},
    {'s', "INT", "write S array every 1<<INT rows (effective with -b) [13]"},
    {'r', "INT", "start decoding from row INT (effective w/o -S) [0]"},
    {'n', "INT", "read INT rows starting from -r (effective w/o -S) [inf]"},
    {'c', "INT", "decode column INT (there can be multiple -c; effective w/o -S) [inf]"},
};
#define PV_N_OPTS ((int)(sizeof(PV_OPTS) / sizeof(PV_OPTS[0])))

static int pv_usage(void)
{
    int i;
    fprintf(stderr, "Usage: pbfview [options] <in.pbf>|<in.pim>\nOptions:\n");
    for (i = 0; i < PV_N_OPTS; ++i)
        fprintf(stderr, "  -%c %-5s %s\n", PV_OPTS[i].key, PV_OPTS[i].arg ? PV_OPTS[i].arg : "", PV_OPTS[i].help);
    return 1;
}

static void pv_set(pv_job_t *job, int key, const char *arg)
{
    switch (key) {
    case 'S': job->text_in = 1; break;
    case 'b': job->pbf_out = 1; break;
    case 's': job->shift = atoi(arg); break;
    case 'r': job->first_row = atol(arg); break;
    case 'n': job->max_rows = atol(arg); break;
    case 'c':
        if (job->n_cols == job->cap_cols) {
            job->cap_cols = job->cap_cols ? 2 * job->cap_cols : 16;
            job->cols = (int32_t*)realloc(job->cols, (size_t)job->cap_cols * sizeof(int32_t));
        }
        job->cols[job->n_cols++] = (int32_t)atol(arg);
        break;
    default: break;                                                  /* getopt has complained */
    }
}

static int pv_parse(pv_job_t *job, int argc, char **argv)
{
    char spec[2 * PV_N_OPTS + 1];
    int i, l = 0, c;
    for (i = 0; i < PV_N_OPTS; ++i) { spec[l++] = PV_OPTS[i].key; if (PV_OPTS[i].arg) spec[l++] = ':'; }
    spec[l] = 0;
    memset(job, 0, sizeof(*job));
    job->shift = 13; job->max_rows = -1;
    while ((c = getopt(argc, argv, spec)) >= 0) pv_set(job, c, optarg);
    if (optind >= argc) return -1;
    job->input = argv[optind];
    return 0;
}

/* ---- the sink: rows of m cells, as text or into the device encoder ---- */
typedef struct {
    bgth_encoder_t *enc;                                             /* NULL: PIM text on stdout */
    uint8_t        *codes;                                           /* rows waiting for the encoder */
    int64_t         held;
    int             m, g, col;                                       /* col: cells of the current row so far */
} pv_sink_t;

static int sink_open(pv_sink_t *s, const pv_job_t *job, int m, int g)
{
    memset(s, 0, sizeof(*s));
    s->m = m; s->g = g;
    if (!job->pbf_out) { printf("PIM1 %d %d\n", m, g); return 0; }
    if ((s->enc = bgth_encoder_open(m, g, job->shift, 0)) == NULL) { fprintf(stderr, "[E::%s] %s\n", __func__, bgth_encoder_last_error()); return -1; }
    s->codes = (uint8_t*)malloc((size_t)m * PV_ROWS_PER_WRITE);
    return s->codes ? 0 : -1;
}

static int sink_drain(pv_sink_t *s)
{
    if (s->enc && s->held > 0 && bgth_encoder_write(s->enc, s->codes, s->held) < 0) {
        fprintf(stderr, "[E::%s] %s\n", __func__, bgth_encoder_last_error());
        return -1;
    }
    s->held = 0;
    return 0;
}

/* one cell of the current row: text goes out at once (an unfinished last row of a PIM echo stays visible, as in the
 * reference), codes wait for the row to be complete */
static void sink_cell(pv_sink_t *s, long long v)
{
    if (s->enc) s->codes[(size_t)s->held * s->m + s->col] = (uint8_t)(v & ((1 << s->g) - 1));   /* (a byte per cell: g <= 8, the encoder's limit) */
    else printf(s->col ? " %lld" : "%lld", v);
    ++s->col;
}

static int sink_end_row(pv_sink_t *s)
{
    s->col = 0;
    if (!s->enc) { putchar('\n'); return 0; }
    return ++s->held == PV_ROWS_PER_WRITE ? sink_drain(s) : 0;
}

static int sink_close(pv_sink_t *s, int ok)
{
    int rc = 0;
    if (s->enc) {
        if (ok && sink_drain(s) < 0) { ok = 0; rc = -1; }
        if (ok) {
            uint8_t *img = NULL;
            const int64_t len = bgth_encoder_finish(s->enc, &img);
            if (len < 0) { fprintf(stderr, "[E::%s] %s\n", __func__, bgth_encoder_last_error()); rc = -1; }
            else { fwrite(img, 1, (size_t)len, stdout); bgth_encoder_free_image(img); }
        }
        bgth_encoder_close(s->enc);
    }
    free(s->codes);
    fflush(stdout);
    return rc;
}

/* ---- source 1: PIM text ---- */
static int pump_text(const pv_job_t *job)
{
    FILE *fp = strcmp(job->input, "-") ? fopen(job->input, "r") : stdin;
    char magic[256];
    pv_sink_t sink;
    long cell = 0;
    int m = 0, g = 0, ok = 1;
    if (fp == NULL) { fprintf(stderr, "[E::%s] cannot open '%s'\n", __func__, job->input); return 1; }
    if (fscanf(fp, "%255s%d%d", magic, &m, &g) != 3 || m <= 0 || g <= 0) { fprintf(stderr, "[E::%s] not a PIM file\n", __func__); return 1; }
    if (sink_open(&sink, job, m, g) < 0) { sink_close(&sink, 0); return 1; }
    /* Rows are read cell by cell.  The end of the file is noticed only by the read AFTER the last number (the stream's
     * end-of-file flag, tested before every cell as the reference tests it): that read converts nothing and leaves `cell`
     * as it was, so after a file that ends in a newline the echo shows the last value once more, without a newline -- the
     * reference's output, byte for byte; a row that is not complete never reaches the encoder. */
    while (ok) {
        while (sink.col < m && !feof(fp)) {
            if (fscanf(fp, "%ld", &cell) == 0) { fprintf(stderr, "[E::%s] not a number in the PIM input\n", __func__); ok = 0; break; }
            sink_cell(&sink, cell);
        }
        if (!ok || sink.col < m) break;
        if (sink_end_row(&sink) < 0) ok = 0;
    }
    if (fp != stdin) fclose(fp);
    return sink_close(&sink, ok) < 0 || !ok;
}

/* ---- source 2: a .pbf through the device reader ---- */
static int pump_pbf(const pv_job_t *job)
{
    bgth_pbf_t *in = bgth_pbf_open(job->input, 0);
    bgth_reader_t *rd = in ? bgth_reader_create(in) : NULL;
    pv_sink_t sink;
    const uint8_t **planes;
    int64_t left = job->max_rows < 0 ? INT64_MAX : job->max_rows;
    int m, g, j, ok = 1, opened = 0;
    if (rd == NULL) { fprintf(stderr, "[E::%s] %s\n", __func__, bgth_last_error()); if (in) bgth_pbf_close(in); return 1; }
    g = bgth_pbf_get_g(in);
    if (job->n_cols > 0 && bgth_reader_select(rd, job->n_cols, job->cols, NULL, 1) < 0) { fprintf(stderr, "[E::%s] %s\n", __func__, bgth_last_error()); ok = 0; }
    /* the width of a row: the columns named, or -- when the list names every column of the file, which pbf_subset treats as
     * no subset at all (pbwt.c:377) -- still the number named */
    m = job->n_cols > 0 && job->n_cols < bgth_pbf_get_m(in) ? job->n_cols : bgth_reader_width(rd);
    if (ok) { opened = 1; if (sink_open(&sink, job, m, g) < 0) ok = 0; }
    if (ok && job->first_row > 0) {
        /* pbf_seek from row 0 (pbwt.c:349-372): a target inside the file is reached; one behind the end but inside the forward
         * window of 1 << shift rows is "reached" by reading into the end -- nothing is left; one behind both makes the seek
         * fail and leaves the reader at row 0 */
        if (job->first_row < bgth_pbf_get_n(in)) bgth_reader_seek(rd, job->first_row);
        else if (job->first_row <= ((int64_t)1 << bgth_pbf_get_shift(in))) left = 0;
    }
    for (; ok && left > 0 && (planes = bgth_reader_read(rd)) != NULL; --left) {
        for (j = 0; j < m; ++j) {
            long long v = 0;
            int k;
            for (k = 0; k < g; ++k) v |= (long long)planes[k][j] << k;    /* (any number of planes: pbfview.c is g-agnostic) */
            sink_cell(&sink, v);
        }
        if (sink_end_row(&sink) < 0) ok = 0;
    }
    if (opened && sink_close(&sink, ok) < 0) ok = 0;
    bgth_reader_destroy(rd);
    bgth_pbf_close(in);
    return !ok;
}

int main_pbfview(int argc, char **argv)
{
    pv_job_t job;
    int rc;
    if (pv_parse(&job, argc, argv) < 0) { free(job.cols); return pv_usage(); }
    rc = job.text_in ? pump_text(&job) : pump_pbf(&job);
    free(job.cols);
    return rc;
}
