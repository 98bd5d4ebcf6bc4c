/* trio_writer.c -- write a complete synthetic BGT database (prefix.pbf / .bcf / .bcf.csi / .spl) for the
 * benchmark shapes of BASELINE.json.  The genotype matrix comes from the PBWT-domain generator and its
 * checkpoints from the GPU (bgth_pbf_from_rle); this file adds what `bgt import` writes beside it
 * (reference import.c:55-117): the site-only BCF with INFO/_row, its CSI index (min_shift 14) with the
 * record-number trailer "RNI\1" every 1024 records (hts.h:71, hts.c:541-547), and the sample list.
 * The index is a plain (uncompressed-binning) but valid CSI: the reference loads and queries it. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/bgt_reader.h"
#include "../../include/bgt_hip.h"
#include "../../include/bgt_synth.h"

#define MIN_SHIFT 14
#define REC_SHIFT 10

static int reg2bin(int64_t beg, int64_t end, int depth)
{
    int l, s = MIN_SHIFT, t = ((1 << depth * 3) - 1) / 7;
    for (--end, l = depth; l > 0; --l, s += 3, t -= 1 << l * 3)
        if (beg >> s == end >> s) return t + (int)(beg >> s);
    return 0;
}
static int bin_bot(int bin, int depth)                          /* first leaf window under a bin */
{
    int l = 0, b = bin;
    for (; b; ++l, b = (b - 1) >> 3) {}
    return (bin - ((1 << l * 3) - 1) / 7) << (depth - l) * 3;
}

typedef struct { uint32_t bin; uint64_t beg, end; } chunk_t;
static int cmp_chunk(const void *a, const void *b)
{
    const chunk_t *x = (const chunk_t*)a, *y = (const chunk_t*)b;
    return x->bin != y->bin ? (x->bin < y->bin ? -1 : 1) : (x->beg < y->beg ? -1 : x->beg > y->beg);
}

int bgt_synth_trio(const char *prefix, int n_samples, int64_t n_sites, uint64_t seed, int device)
{
    const int m = 2 * n_samples, shift = 13;
    const int64_t ctg_len = 135006516;
    char *fn = (char*)malloc(strlen(prefix) + 16);
    bgth_synth_t *sy;
    bgth_pbf_t *img;
    FILE *fp;
    bgzw_t *bz;
    bcf1_t *b;
    kstring_t h = {0, 0, 0};
    int depth, row_key, rc = -1;
    int64_t r, s, max_len;
    uint64_t off_first, last_off, *lin = NULL, *ridx = NULL;
    int64_t n_lin = 0, n_ridx = 0, n_chunks = 0, cap_chunks = 0;
    chunk_t *chunks = NULL;
    uint32_t save_bin = 0xffffffffu;
    uint64_t save_off = 0;
    bcf_hdr_t *hdr;

    /* ---- .pbf: rows in the PBWT domain, checkpoints on the device ---- */
    if ((sy = bgth_synth_rows(m, 0, n_sites, seed, 0)) == NULL) goto done;
    img = bgth_pbf_from_rle(m, 2, shift, n_sites, bgth_synth_rle(sy), bgth_synth_len(sy), device);
    bgth_synth_free(sy);
    if (img == NULL) { fprintf(stderr, "[E::%s] %s\n", __func__, bgth_last_error()); goto done; }
    sprintf(fn, "%s.pbf", prefix);
    if (bgth_pbf_save(img, fn) < 0) { fprintf(stderr, "[E::%s] %s\n", __func__, bgth_last_error()); bgth_pbf_close(img); goto done; }
    bgth_pbf_close(img);

    /* ---- .spl ---- */
    sprintf(fn, "%s.spl", prefix);
    if ((fp = fopen(fn, "w")) == NULL) goto done;
    for (r = 0; r < n_samples; ++r) fprintf(fp, "S%06lld\tpop:Z:%c\tidx:i:%lld\n", (long long)r, "ABC"[r % 3], (long long)r);
    fclose(fp);

    /* ---- .bcf (+ index bookkeeping while writing) ---- */
    ks_puts(&h, "##fileformat=VCFv4.1\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n");
    ks_printf(&h, "##contig=<ID=11,length=%lld>\n", (long long)ctg_len);
    ks_puts(&h, "##INFO=<ID=_row,Number=1,Type=Integer,Description=\"row number\">\n");
    ks_puts(&h, "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO");
    hdr = bcf_hdr_init();
    hdr->text = h.s; hdr->l_text = (int32_t)h.l + 1; hdr->m_text = (int32_t)h.m;
    bcf_hdr_parse(hdr);
    row_key = bcf_id2int(hdr, BCF_DT_ID, "_row");
    max_len = ctg_len + 256;
    for (depth = 0, s = 1 << MIN_SHIFT; max_len > s; ++depth, s <<= 3) {}       /* ref vcf.c:1013-1014 */
    sprintf(fn, "%s.bcf", prefix);
    if ((fp = fopen(fn, "wb")) == NULL) { bcf_hdr_destroy(hdr); goto done; }
    bz = bgzw_open(fp, -1);
    bcf_hdr_write_stream(bz, hdr);
    off_first = last_off = bgzw_tell(bz);
    n_lin = (ctg_len >> MIN_SHIFT) + 1;
    lin = (uint64_t*)malloc((size_t)n_lin * 8);
    memset(lin, 0xff, (size_t)n_lin * 8);
    ridx = (uint64_t*)malloc((size_t)((n_sites >> REC_SHIFT) + 2) * 8);
    b = bcf_init1();
    for (r = 0; r < n_sites; ++r) {
        int32_t pos1, n_allele, row = (int32_t)r;
        char ref, alt;
        int bin;
        int64_t w;
        bgth_synth_site(seed, r, &pos1, &ref, &alt, &n_allele);
        bcf_set_site(b, 0, pos1 - 1, 1, &ref, 1, &alt, 1, n_allele > 2 ? "<M>" : NULL);
        bcf_append_info_ints(hdr, b, "_row", 1, &row);
        (void)row_key;
        /* index: this record starts at last_off */
        w = (int64_t)(pos1 - 1) >> MIN_SHIFT;
        if (lin[w] == (uint64_t)-1) lin[w] = last_off;
        bin = reg2bin(pos1 - 1, pos1, depth);
        if ((uint32_t)bin != save_bin) {
            if (save_bin != 0xffffffffu) {
                if (n_chunks == cap_chunks) { cap_chunks = cap_chunks ? cap_chunks * 2 : 1024; chunks = (chunk_t*)realloc(chunks, (size_t)cap_chunks * sizeof(chunk_t)); }
                chunks[n_chunks].bin = save_bin; chunks[n_chunks].beg = save_off; chunks[n_chunks++].end = last_off;
            }
            save_bin = (uint32_t)bin; save_off = last_off;
        }
        if ((r & ((1 << REC_SHIFT) - 1)) == 0) ridx[n_ridx++] = last_off;
        bcf_write1_stream(bz, b);
        last_off = bgzw_tell(bz);
    }
    if (save_bin != 0xffffffffu) {
        if (n_chunks == cap_chunks) { cap_chunks = cap_chunks ? cap_chunks * 2 : 1024; chunks = (chunk_t*)realloc(chunks, (size_t)cap_chunks * sizeof(chunk_t)); }
        chunks[n_chunks].bin = save_bin; chunks[n_chunks].beg = save_off; chunks[n_chunks++].end = last_off;
    }
    bcf_destroy1(b);
    bgzw_close(bz);
    fclose(fp);

    /* ---- .bcf.csi ---- */
    {
        const int n_bins = ((1 << (3 * depth + 3)) - 1) / 7;
        int32_t x[3] = {MIN_SHIFT, depth, 0}, n_ref = 1, n_bin = 0, rec_shift = REC_SHIFT, n_r = (int32_t)n_ridx;
        uint64_t zero = 0, prev = off_first;
        int64_t i, j;
        for (i = 0; i < n_lin; ++i) { if (lin[i] == (uint64_t)-1) lin[i] = prev; else prev = lin[i]; }   /* fill gaps */
        qsort(chunks, (size_t)n_chunks, sizeof(chunk_t), cmp_chunk);
        for (i = 0; i < n_chunks; ++i) if (i == 0 || chunks[i].bin != chunks[i - 1].bin) ++n_bin;
        sprintf(fn, "%s.bcf.csi", prefix);
        if ((fp = fopen(fn, "wb")) == NULL) { bcf_hdr_destroy(hdr); goto done; }
        bz = bgzw_open(fp, -1);
        bgzw_write(bz, "CSI\1", 4); bgzw_write(bz, x, 12);
        bgzw_write(bz, &n_ref, 4);
        n_bin += n_sites > 0;                                                    /* + the statistics pseudo-bin */
        bgzw_write(bz, &n_bin, 4);
        for (i = 0; i < n_chunks; i = j) {
            uint64_t loff = lin[bin_bot((int)chunks[i].bin, depth)];
            int32_t nc;
            for (j = i; j < n_chunks && chunks[j].bin == chunks[i].bin; ++j) {}
            nc = (int32_t)(j - i);
            bgzw_write(bz, &chunks[i].bin, 4); bgzw_write(bz, &loff, 8); bgzw_write(bz, &nc, 4);
            for (; i < j; ++i) { bgzw_write(bz, &chunks[i].beg, 8); bgzw_write(bz, &chunks[i].end, 8); }
        }
        if (n_sites > 0) {                                                       /* pseudo-bin n_bins+1: file span, #mapped */
            uint32_t pb = (uint32_t)n_bins + 1; int32_t two = 2; uint64_t nm = (uint64_t)n_sites;
            bgzw_write(bz, &pb, 4); bgzw_write(bz, &zero, 8); bgzw_write(bz, &two, 4);
            bgzw_write(bz, &off_first, 8); bgzw_write(bz, &last_off, 8);
            bgzw_write(bz, &nm, 8); bgzw_write(bz, &zero, 8);
        }
        bgzw_write(bz, &zero, 8);                                                /* records without coordinates */
        bgzw_write(bz, "RNI\1", 4); bgzw_write(bz, &n_sites, 8); bgzw_write(bz, &rec_shift, 4);
        bgzw_write(bz, &n_r, 4); bgzw_write(bz, ridx, (size_t)n_ridx * 8);
        bgzw_close(bz);
        fclose(fp);
    }
    bcf_hdr_destroy(hdr);
    rc = 0;
done:
    free(fn); free(lin); free(ridx); free(chunks);
    return rc;
}
