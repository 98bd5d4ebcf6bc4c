/*
 * orc_scan.c -- CPU ORACLE (test infrastructure only; see orc.h): the allele-count reduction of
 * bgtm_cal_info and the per-site scan loop that `bgt view -G [-s ...] [-f ...]` drives.
 */
#include <stdlib.h>
#include <string.h>
#include "orc.h"

/* ref bgt.c:735-757.  Code of haplotype i is a1[i]<<1 | a0[i] (0 REF, 1 ALT, 2 missing, 3 <M>).
 * AN = n(0)+n(1)+n(3), AC = n(1), AC<M> = n(3); per group the same over that group's samples
 * (group[] is per SAMPLE, 1-based; haplotype i belongs to sample i>>1).  Per-group numbers are only
 * produced when there are at least two groups (ref :740), and then the overall numbers are their
 * sums (ref :749) -- which equals the plain histogram because every selected sample has a group. */
void orc_allele_counts(int n_hap, const uint8_t *a0, const uint8_t *a1,
                       const uint32_t *group, int n_groups, int32_t *out)
{
    int32_t tot[4] = {0, 0, 0, 0};
    int i, g;
    if (n_groups > 1 && group) {
        int32_t (*h)[4] = (int32_t(*)[4])calloc((size_t)n_groups, sizeof(*h));
        for (i = 0; i < n_hap; ++i) ++h[group[i >> 1] - 1][a1[i] << 1 | a0[i]];
        for (g = 0; g < n_groups; ++g) {
            out[3 + 3 * g] = h[g][0] + h[g][1] + h[g][3];
            out[4 + 3 * g] = h[g][1];
            out[5 + 3 * g] = h[g][3];
            tot[0] += h[g][0]; tot[1] += h[g][1]; tot[2] += h[g][2]; tot[3] += h[g][3];
        }
        free(h);
    } else {
        for (i = 0; i < n_hap; ++i) ++tot[a1[i] << 1 | a0[i]];
    }
    out[0] = tot[0] + tot[1] + tot[3];
    out[1] = tot[1];
    out[2] = tot[3];
}

/* The loop of view.c:151 / bgt.c:797-878 reduced to its genotype arithmetic for one database:
 * seek (ref bgt.c:341 pbf_seek) + read (ref :342) + histogram (ref :852).  Subset selection must have
 * been installed with orc_pbf_subset() beforehand (ref bgt.c:239-243). */
int64_t orc_scan(orc_pbf_t *p, int64_t row0, int64_t row1, const uint32_t *group, int n_groups,
                 int32_t *counts, uint8_t *gt)
{
    const int g_out = n_groups > 1 ? n_groups : 0;
    const int stride = 3 * (1 + g_out);
    int64_t r;
    if (orc_pbf_g(p) != 2) return -1;
    for (r = row0; r < row1; ++r) {
        const uint8_t **a;
        int n_hap, i;
        if (orc_pbf_seek(p, r) < 0) return -2;
        if ((a = orc_pbf_read(p)) == NULL) return -3;
        n_hap = orc_pbf_subset_width(p);
        orc_allele_counts(n_hap, a[0], a[1], group, n_groups, counts + (r - row0) * stride);
        if (gt) {
            const int nb = (n_hap + 3) / 4;
            uint8_t *dst = gt + (r - row0) * (int64_t)nb;
            memset(dst, 0, (size_t)nb);
            for (i = 0; i < n_hap; ++i)
                dst[i >> 2] |= (uint8_t)((a[1][i] << 1 | a[0][i]) << ((i & 3) * 2));
        }
    }
    return row1 - row0;
}
