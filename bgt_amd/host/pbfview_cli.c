/* pbfview_cli.c -- `bgt pbfview`: the reference's codec-level tool (pbfview.c:7-101) over the device codec.
 *
 *   decode   <in.pbf>             -> PIM text ("PIM1 m g", then one line of m integers per row: sum of plane k << k)
 *            -c COL (repeatable)  -> only these columns, in the order given (pbf_subset, pbwt.c:374-388)
 *            -r ROW, -n COUNT     -> from row ROW on (pbf_seek, pbwt.c:349-372), at most COUNT rows
 *   encode   -S <in.pim> -b       -> PBF on stdout, 'S' records every 1 << shift rows (-s, default 13)
 *   recode   <in.pbf> -b [-c ..]  -> the decoded rows written again
 *
 * Decoding is bgth_reader_read (rank tracking on the device), encoding bgth_encoder_* (PBWT order kept on the device);
 * the host only parses and prints integers.  The encoder takes one or two bit planes; the reader holds exactly BGT's two
 * (import.c:68) and refuses other files with a message. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <unistd.h>
#include "../../include/bgt_hip.h"

#define ROWS_PER_WRITE 4096

static int flush_rows(bgth_encoder_t *enc, const uint8_t *codes, int64_t n)
{
    if (n > 0 && bgth_encoder_write(enc, codes, n) < 0) { fprintf(stderr, "[E::%s] %s\n", __func__, bgth_encoder_last_error()); return -1; }
    return 0;
}

static int finish_image(bgth_encoder_t *enc)
{
    uint8_t *img = NULL;
    const int64_t len = bgth_encoder_finish(enc, &img);
    if (len < 0) { fprintf(stderr, "[E::%s] %s\n", __func__, bgth_encoder_last_error()); return -1; }
    fwrite(img, 1, (size_t)len, stdout);
    bgth_encoder_free_image(img);
    return 0;
}

int main_pbfview(int argc, char **argv)
{
    int c, in_txt = 0, out_pbf = 0, m_sub = 0, n_sub = 0, shift = 13, rc = 0;
    int32_t *sub = NULL;
    int64_t row_start = 0, n_rec = -1;
    bgth_encoder_t *enc = NULL;
    uint8_t *codes = NULL;
    while ((c = getopt(argc, argv, "Sbc:r:n:s:")) >= 0) {
        if (c == 'S') in_txt = 1;
        else if (c == 'b') out_pbf = 1;
        else if (c == 'r') row_start = atol(optarg);
        else if (c == 'n') n_rec = atol(optarg);
        else if (c == 's') shift = atoi(optarg);
        else if (c == 'c') {
            if (n_sub == m_sub) { m_sub = m_sub ? m_sub << 1 : 4; sub = (int32_t*)realloc(sub, (size_t)m_sub * sizeof(int32_t)); }
            sub[n_sub++] = (int32_t)atol(optarg);
        }
    }
    if (argc == optind) {
        fprintf(stderr, "Usage: pbfview [options] <in.pbf>|<in.pim>\n");
        fprintf(stderr, "Options:\n");
        fprintf(stderr, "  -S       input is PIM (portable integer matrix format)\n");
        fprintf(stderr, "  -b       output PBF (positional BWT format)\n");
        fprintf(stderr, "  -s INT   write S array every 1<<INT rows (effective with -b) [%d]\n", shift);
        fprintf(stderr, "  -r INT   start decoding from row INT (effective w/o -S) [0]\n");
        fprintf(stderr, "  -n INT   read INT rows starting from -r (effective w/o -S) [inf]\n");
        fprintf(stderr, "  -c INT   decode column INT (there can be multiple -c; effective w/o -S) [inf]\n");
        return 1;
    }
    if (n_rec < 0) n_rec = INT64_MAX;
    if (in_txt) {                                               /* PIM text in (ref pbfview.c:41-70) */
        char magic[256];
        FILE *fp = strcmp(argv[optind], "-") ? fopen(argv[optind], "r") : stdin;
        int m = 0, g = 0, i;
        int64_t n_buf = 0;
        if (fp == NULL) { fprintf(stderr, "[E::%s] cannot open '%s'\n", __func__, argv[optind]); return 1; }
        if (fscanf(fp, "%255s%d%d", magic, &m, &g) != 3 || m <= 0 || g <= 0) { fprintf(stderr, "[E::%s] not a PIM file\n", __func__); return 1; }
        if (out_pbf) {
            if (g > 2) { fprintf(stderr, "[E::%s] the device codec writes one or two bit planes, the input has %d\n", __func__, g); return 1; }
            if ((enc = bgth_encoder_open(m, g, shift, 0)) == NULL) { fprintf(stderr, "[E::%s] %s\n", __func__, bgth_encoder_last_error()); return 1; }
            codes = (uint8_t*)malloc((size_t)m * ROWS_PER_WRITE);
        } else printf("PIM1 %d %d\n", m, g);
        {
        long x = 0;
        while (!feof(fp)) {
            /* the reference's loop, quirk included: end-of-file is only noticed one read late, so after a file that ends in
             * a newline the echo prints the last value once more (no newline after it); the encoder sees no such row */
            for (i = 0; i < m; ++i) {
                if (feof(fp)) break;
                if (fscanf(fp, "%ld", &x) == 0) { fprintf(stderr, "[E::%s] not a number in the PIM input\n", __func__); rc = 1; break; }
                if (enc) codes[(size_t)n_buf * m + i] = (uint8_t)(x & ((1 << g) - 1));
                else { if (i) putchar(' '); printf("%ld", x); }
            }
            if (i < m) break;                                   /* an incomplete last row is dropped, as the reference does */
            if (enc) { if (++n_buf == ROWS_PER_WRITE) { if (flush_rows(enc, codes, n_buf) < 0) { rc = 1; break; } n_buf = 0; } }
            else putchar('\n');
        }
        }
        if (enc && rc == 0 && flush_rows(enc, codes, n_buf) < 0) rc = 1;
        if (fp != stdin) fclose(fp);
    } else {                                                    /* PBF in (ref pbfview.c:71-96) */
        bgth_pbf_t *in = bgth_pbf_open(argv[optind], 0);
        bgth_reader_t *rd;
        const uint8_t **a;
        int m, g, j, k;
        int64_t i, n_buf = 0;
        if (in == NULL) { fprintf(stderr, "[E::%s] %s\n", __func__, bgth_last_error()); return 1; }
        if ((rd = bgth_reader_create(in)) == NULL) { fprintf(stderr, "[E::%s] %s\n", __func__, bgth_last_error()); bgth_pbf_close(in); return 1; }
        g = bgth_pbf_get_g(in);
        if (n_sub > 0 && bgth_reader_select(rd, n_sub, sub, NULL, 1) < 0) { fprintf(stderr, "[E::%s] %s\n", __func__, bgth_last_error()); rc = 1; }
        m = bgth_reader_width(rd);                              /* (all the columns when the list names every one of them, pbwt.c:377) */
        if (n_sub > 0 && n_sub < bgth_pbf_get_m(in)) m = n_sub;
        if (rc == 0 && out_pbf) {
            if ((enc = bgth_encoder_open(m, g, shift, 0)) == NULL) { fprintf(stderr, "[E::%s] %s\n", __func__, bgth_encoder_last_error()); rc = 1; }
            codes = (uint8_t*)malloc((size_t)m * ROWS_PER_WRITE);
        } else if (rc == 0) printf("PIM1 %d %d\n", m, g);
        if (rc == 0 && row_start > 0) {                         /* pbf_seek from row 0 (pbwt.c:349-372) */
            if (row_start < bgth_pbf_get_n(in)) bgth_reader_seek(rd, row_start);
            else if (row_start <= ((int64_t)1 << bgth_pbf_get_shift(in))) n_rec = 0;   /* read forward into the end: nothing is left */
            /* else: behind the end and behind the forward window: the seek fails and the reader stays at row 0 (:359) */
        }
        for (i = 0; rc == 0 && i < n_rec && (a = bgth_reader_read(rd)) != NULL; ++i) {
            if (!enc) {
                for (j = 0; j < m; ++j) {
                    unsigned long long x = 0;
                    for (k = 0; k < g; ++k) x |= (unsigned long long)a[k][j] << k;
                    if (j) putchar(' ');
                    printf("%llu", x);
                }
                putchar('\n');
            } else {
                uint8_t *dst = codes + (size_t)n_buf * m;
                for (j = 0; j < m; ++j) dst[j] = (uint8_t)(a[0][j] | (g > 1 ? a[1][j] << 1 : 0));
                if (++n_buf == ROWS_PER_WRITE) { if (flush_rows(enc, codes, n_buf) < 0) rc = 1; n_buf = 0; }
            }
        }
        if (enc && rc == 0 && flush_rows(enc, codes, n_buf) < 0) rc = 1;
        bgth_reader_destroy(rd);
        bgth_pbf_close(in);
    }
    if (enc) {
        if (rc == 0 && finish_image(enc) < 0) rc = 1;
        bgth_encoder_close(enc);
    }
    fflush(stdout);
    free(codes); free(sub);
    return rc;
}
