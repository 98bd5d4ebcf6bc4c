/* Test harness (not product code): drives the READER API of bgt.h exactly as a caller of the reference would, and
 * prints everything the calls hand back.  The same source is compiled twice -- against bgt_amd/lib/libbgt.so (MI355X)
 * and against oracle/_ref/libbgt_ref.so (the compiled reference; same struct layouts, tests/test_host_shell.py) -- and
 * the two outputs must be identical.
 *
 *   api_dump read <prefix> [region|-] [start]      bgt_read() to the end (reference bgt.c:347-356): row number, the
 *                                                  record's fixed fields, shared / indiv bytes, and its VCF line
 *   api_dump server <max_gt> <prefix> [prefix..] -- [-g] [-C] [-S] [-H] [-f e] [-r reg] [-i n] [-n n] [-t fmt] [-a al] [-s e]..
 *                                                  the call sequence of one query of bgt-server.go:220-373 (bgs_query):
 *                                                  flags, setters in the server's order, bgtm_prepare, bgtm_test_mgs,
 *                                                  header, bgtm_read until n_read > n or n_gt_read > max_gt, -H/-S text
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bgt.h"

int vcf_format1(const bcf_hdr_t *h, const bcf1_t *v, kstring_t *s);

static void hex(const char *tag, const char *s, size_t l)
{
    size_t i;
    printf(" %s=%zu:", tag, l);
    for (i = 0; i < l; ++i) printf("%02x", (unsigned char)s[i]);
}

static int do_read(int argc, char **argv)
{
    bgt_file_t *f = bgt_open(argv[0]);
    bgt_t *r;
    bcf1_t *b;
    kstring_t s = {0, 0, 0};
    int row, n = 0;
    if (f == NULL) { printf("open failed\n"); return 1; }
    r = bgt_reader_init(f);
    if (argc > 1 && strcmp(argv[1], "-") != 0) printf("set_region=%d\n", bgt_set_region(r, argv[1]));
    if (argc > 2) printf("set_start=%d\n", bgt_set_start(r, atol(argv[2])));
    b = bcf_init1();
    while ((row = bgt_read(r, b)) >= 0) {
        uint32_t q;
        memcpy(&q, &b->qual, 4);
        if (n++ == 0) printf("header=%s", r->h_out->text);
        printf("row=%d rid=%d pos=%d rlen=%d qual=%08x n_info=%d n_allele=%d n_fmt=%d n_sample=%d", row, b->rid, b->pos, b->rlen,
               q, (int)b->n_info, (int)b->n_allele, (int)b->n_fmt, (int)b->n_sample);
        hex("shared", b->shared.s, b->shared.l);
        hex("indiv", b->indiv.s, b->indiv.l);
        s.l = 0;
        vcf_format1(r->h_out, b, &s);
        printf(" vcf=%s\n", s.s);
    }
    printf("end=%d after %d sites\n", row < -1 ? -2 : row, n);
    free(s.s);
    bcf_destroy1(b);
    bgt_reader_destroy(r);
    bgt_close(f);
    return 0;
}

static int do_server(int argc, char **argv)
{
    const uint64_t max_gt = strtoull(argv[0], NULL, 10);
    bgt_file_t *files[16];
    bgtm_t *bm;
    bcf1_t *b;
    kstring_t s = {0, 0, 0};
    int i, n_files = 0, flag = BGT_F_NO_GT, n_read = 0, max_read = 2147483647, ret = 0, vcf_out = 1;
    for (i = 1; i < argc && strcmp(argv[i], "--") != 0 && n_files < 16; ++i) {
        if ((files[n_files] = bgt_open(argv[i])) == NULL) { printf("open failed: %s\n", argv[i]); return 1; }
        ++n_files;
    }
    bm = bgtm_reader_init(n_files, files);                              /* bgt-server.go:233 */
    bgtm_set_mgs(bm, getenv("API_DUMP_MGS") ? atoi(getenv("API_DUMP_MGS")) : 1);   /* :235 (the server's -g; 1 when unset) */
    {   /* flags first (:237-254), then the setters in the server's order: f r i n t a s */
        int k;
        const char *order = "fritnas";
        for (k = i + 1; k < argc; ++k) {
            if (strcmp(argv[k], "-g") == 0) flag &= ~BGT_F_NO_GT;
            else if (strcmp(argv[k], "-C") == 0 || strcmp(argv[k], "-s") == 0) flag |= BGT_F_SET_AC;
            else if (strcmp(argv[k], "-S") == 0) flag |= BGT_F_CNT_AL;
            else if (strcmp(argv[k], "-H") == 0) flag |= BGT_F_CNT_HAP;
        }
        bgtm_set_flag(bm, flag);
        if (flag & (BGT_F_CNT_AL | BGT_F_CNT_HAP)) vcf_out = 0;
        for (; *order; ++order)
            for (k = i + 1; k + 1 < argc; ++k) {
                if (argv[k][0] != '-' || argv[k][1] != *order || argv[k][2]) continue;
                if (*order == 'f') printf("set_flt_site=%d\n", bgtm_set_flt_site(bm, argv[k + 1]));
                else if (*order == 'r') printf("set_region=%d\n", bgtm_set_region(bm, argv[k + 1]));
                else if (*order == 'i') printf("set_start=%d\n", bgtm_set_start(bm, atol(argv[k + 1])));
                else if (*order == 'n') max_read = atoi(argv[k + 1]);
                else if (*order == 't') { printf("set_table=%d\n", bgtm_set_table(bm, argv[k + 1])); vcf_out = 0; }
                else if (*order == 'a') printf("set_alleles=%d\n", bgtm_set_alleles(bm, argv[k + 1], NULL, NULL));
                else if (*order == 's') printf("add_group=%d\n", bgtm_add_group(bm, argv[k + 1]));
            }
    }
    bgtm_prepare(bm);                                                   /* :322 */
    printf("test_mgs=%d n_out=%d n_groups=%d\n", bgtm_test_mgs(bm), bm->n_out, bm->n_groups);
    if (vcf_out) printf("%s\n", bm->h_out->text);                       /* :329-332 reads bm.h_out.text */
    b = bcf_init1();
    for (;;) {                                                          /* :334-352 */
        if (n_read > max_read || bm->n_gt_read > max_gt) break;
        if ((ret = bgtm_read(bm, b)) < 0) break;
        if (vcf_out) { s.l = 0; vcf_format1(bm->h_out, b, &s); printf("%s\n", s.s); }   /* bgtm_format_bcf1, :24-29 */
        if (!(flag & BGT_F_NO_GT) && bm->n_out > 0) {                   /* bgt.h:70: the merged site's two byte planes (bgt.c:829-842) */
            hex("a0", (const char*)bm->a[0], (size_t)bm->n_out << 1);
            hex("a1", (const char*)bm->a[1], (size_t)bm->n_out << 1);
            printf("\n");
        }
        else if (bm->n_fields > 0) printf("%s\n", bm->tbl_line.s);      /* :348 */
        ++n_read;
    }
    if (!vcf_out && bm->n_aal > 0) {                                    /* :355-368 */
        if (flag & BGT_F_CNT_HAP) { int n_hap; bgt_hapcnt_t *hc = bgtm_hapcnt(bm, &n_hap); char *t = bgtm_hapcnt_print_destroy(bm, n_hap, hc); if (t) fputs(t, stdout); free(t); }
        if (flag & BGT_F_CNT_AL) { char *t = bgtm_alcnt_print(bm); if (t) fputs(t, stdout); free(t); }   /* (NULL: no sample carries them all) */
    }
    if (!vcf_out && bm->n_aal > 0) {                                    /* the arrays themselves (bgt.c:859-876 fills them; bgt.h:112-113) */
        int j;
        if ((flag & BGT_F_CNT_AL) && bm->alcnt) { printf("alcnt[%d]:", bm->n_out); for (j = 0; j < bm->n_out; ++j) printf(" %d", bm->alcnt[j]); printf("\n"); }
        if ((flag & BGT_F_CNT_HAP) && bm->hap) { printf("hap[%d]:", 2 * bm->n_out); for (j = 0; j < 2 * bm->n_out; ++j) printf(" %llx", (unsigned long long)bm->hap[j]); printf("\n"); }
    }
    if (n_read > max_read || bm->n_gt_read > max_gt) printf("*\n");
    printf("records=%d n_gt_read=%llu n_aal=%d\n", n_read, (unsigned long long)bm->n_gt_read, bm->n_aal);
    free(s.s);
    bcf_destroy1(b);
    bgtm_reader_destroy(bm);
    for (i = 0; i < n_files; ++i) bgt_close(files[i]);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 3 && strcmp(argv[1], "read") == 0) return do_read(argc - 2, argv + 2);
    if (argc >= 4 && strcmp(argv[1], "server") == 0) return do_server(argc - 2, argv + 2);
    fprintf(stderr, "usage: api_dump read <prefix> [region|-] [start] | api_dump server <max_gt> <prefix>.. -- [options]\n");
    return 2;
}
