// Host side of the gfx950 kernels: launch geometry (cost model) and the small companion kernels.
// The scan kernel itself lives in scan_device.inc.h and is instantiated by scan_nt.hip, once per launch geometry.
#include "scan_kernels.h"
#include <stdlib.h>

namespace bgth {

// wave helpers used by the companion kernels below are not needed here
// ----------------------------------------------------------------------------------------------------
// launch geometry
// ----------------------------------------------------------------------------------------------------
static const int kLdsBytes = 160 * 1024;

#define BGTH_GEOMS(X) \
    X(256, 2) X(256, 4) X(256, 8) X(256, 12) X(256, 16) X(256, 20) \
    X(512, 4) X(512, 8) X(512, 10) X(512, 12) X(512, 16) X(512, 20) X(512, 24) X(512, 32) X(512, 40) X(512, 48) \
    X(1024, 4) X(1024, 8) X(1024, 10) X(1024, 12) X(1024, 16) X(1024, 20) X(1024, 24) X(1024, 28) X(1024, 32) X(1024, 36) X(1024, 40) \
    X(512, 64) X(512, 80) X(512, 98)          /* team mode only (scan_wide.hip) */

static bool team_only(int nt, int cpt) { return nt == 512 && cpt > 48; }

struct GeomEntry { int nt, cpt; };
static const GeomEntry kGeoms[] = {
#define X(nt, cpt) {nt, cpt},
    BGTH_GEOMS(X)
#undef X
};

static int lds_need(int nw, int K, int G, int threads, int nbuf = 1)
{
    const int cnt = G > 1 ? K * G * 3 * 4 : K * (threads / 64) * 8;
    return nbuf * (16 * K * ((nw + 2) & ~1) + cnt + 2 * K * 4) + 64;
}

// Cost model (cycles per decoded row on one CU; the kernel is VALU-bound at one wave-instruction per
// 4 cycles and SIMD, measured on MI355X):
//   phase B   8 VALU per lookup -> cpt * 2 planes * 8 * 4 cycles * (threads/256 waves per SIMD)
//   phase A   ~(120 + 0.4 * nw) wave-instructions per plane-row, two plane-rows per row, built by
//             (threads/64) waves over a batch of K rows
// Workgroups (blocks x slices) are spread over the 256 CUs; a CU's workgroups share its VALUs.
static int wpp_for(int nt, int K)
{
    int wpp = 1;
    while (wpp * 2 * 2 * K <= nt / 64 && wpp < 8) wpp *= 2;         // waves per plane-row
    return wpp;
}

static long model_cost(int nw, int n_chunks, int n_blk, int nt, int cpt, int K, int *slices_out)
{
    const int nwave = nt / 64, cap = nwave * cpt;
    const int slices = (n_chunks + cap - 1) / cap;
    const int wpp = wpp_for(nt, K);
    // Team kernels (one workgroup per CU): 2 waves per SIMD fill the VALU less well than 4 (measured).  Pipelined narrow
    // kernels share a CU between workgroups, and there the same columns on HALF the waves win: every wave pays ~3 columns' worth
    // per row beside its lookups (statement set-up, counts), so 256 x 20 beats 512 x 10 at m = 5,008 (6.26 against 6.92 ms per
    // 2 M sites), 512 x 20 beats 1024 x 10 at m = 10,000 (5.75 / 6.36 per 1 M), 256 x 8 beats 512 x 4 at m = 2,000 (4.23 /
    // 4.86) -- up to ~24 columns: 512 x 40 at m = 20,000 loses to 1024 x 20 (13.8 against 11.0 ms; scripts/sweep.py, 2026-09-30).
    // (Only where ONE slice holds the selection: with several, the old prices keep the wide-cohort choices -- team kernels,
    // directory path -- where they were measured: m = 34,000 as 512 x 24 x 3 slices runs at 1.97 T lookups/s, 3.0 T on the directory path.)
    // (... and only where EIGHT rows per batch fit the LDS twice over, m <= 20,480 -- where it was measured.  Selections of wider
    // cohorts keep the old prices: every 4th sample of 40,000 haplotypes 5.0 ms as 1024 x 10 against 5.9 as 512 x 20, and a sparse
    // selection of a wide cohort keeps the team geometry that sends it to the plane-split kernels.)
    const bool one_slice = wpp == 1 && slices == 1 && 2 * (16 * 8 * ((nw + 2) & ~1)) <= kLdsBytes;
    long tB;
    if (one_slice) tB = (long)(cpt + 3) * 64 * (nt / 256) * (cpt > 24 ? 130 : 100) / 100;
    else tB = (long)cpt * 64 * (nt / 256) * (nt == 1024 ? 100 : 115) / 100;
    const long lat = 8 + 5 * (nt / 256);                 // cycles per dependent instruction of a building wave
    long tA;
    if (one_slice) {
        // pipelined narrow mode: the builds of one workgroup run beside the walks of the CU's others, so a row's two
        // plane-rows cost their instructions' issue slots (4 cycles each over 4 SIMDs), whatever the workgroup's size --
        // with the (cpt + 3) above this reproduces the measured ratios at m = 2,000 / 5,008 / 10,000 (0.84 / 0.905 / 0.90)
        tA = 2 * (120 + (long)(nw * 4) / 10);
    } else if (wpp == 1) {
        const int rounds = (2 * K + nwave - 1) / nwave;  // plane-rows per wave and batch
        tA = (long)rounds * (120 + (long)(nw * 4) / 10) * lat / K;
    } else {
        // team mode (wide cohorts), per row and wave: chunks of the string -> toggles (~260 instructions per
        // 256-byte chunk), directory trips of 256 words (~75), clearing the row, four barriers.  A wave gets one
        // instruction per 4 cycles x waves per SIMD.
        const int chunks = (40 + (nw * 10) / 65 + 255) / 256, ntrip = (nw + 255) / 256;
        const long instr = (long)((chunks + wpp - 1) / wpp) * 260 + (long)((ntrip + wpp - 1) / wpp) * 60 +
                           3 * (nw / (64 * wpp) + 1) + 150;
        const long per_instr = nt >= 512 ? 13 : 12;                         // cycles, measured (m = 200 k, MI355X)
        tA = (instr * per_instr + 1600) / K;
    }
    const long wgs = (long)n_blk * slices;
    const long per_cu = (wgs + 255) / 256;
    *slices_out = slices;
    return per_cu * (tA + tB);
}

bool choose_geometry(int m, int n_chunks, int G, int n_blk, int want_threads, int want_cpt, int want_K,
                     Geometry *g, bool allow_tog)
{
    const int nw = (m + 31) / 32;
    if (lds_need(nw, 1, G, 1024) > kLdsBytes) return false;
    int best = -1, best_K = 1;
    long best_cost = 0;
    for (int i = 0; i < (int)(sizeof(kGeoms) / sizeof(kGeoms[0])); ++i) {
        const int nt = kGeoms[i].nt, cpt = kGeoms[i].cpt;
        if (want_threads && nt != want_threads) continue;
        if (want_cpt && cpt != want_cpt) continue;
        // 1024 x 28 ... 40 (round 5): for selections ONE such workgroup holds, 24,577-40,960 columns (m = 28,000: 4.16 T lookups/s
        // against 3.31 on the directory path, 36,000: 4.03 / 3.54); wider selections keep the choices they were measured with
        if (nt == 1024 && cpt > 24 && !want_cpt && n_chunks > (nt / 64) * cpt) continue;
        // rows per batch: as many as fit the LDS, at most one per wave (a wave builds <= 2 plane-rows)
        int K = want_K > 0 ? want_K : nt / 64;
        if (K > nt / 64) K = nt / 64;
        while (K > 1 && lds_need(nw, K, G, nt) > kLdsBytes) --K;
        if (team_only(nt, cpt)) {                        // instantiated for team mode only: rows per batch few enough
            while (K > 1 && wpp_for(nt, K) == 1) --K;    // for teams of waves (mid-width cohorts: one wide column slice
        }                                                // in team mode can beat three narrow ones)
        int slices;
        const long cost = model_cost(nw, n_chunks, n_blk, nt, cpt, K, &slices);
        // ties: fewer idle slots, then more threads (more waves to hide LDS latency)
        const long waste = (long)slices * (nt / 64) * cpt - n_chunks;
        const long key = cost * 4096 + (waste < 4095 ? waste : 4095);
        if (best < 0 || key < best_cost || (key == best_cost && nt > kGeoms[best].nt)) best = i, best_cost = key, best_K = K;
    }
    if (best < 0) return false;
    g->threads = kGeoms[best].nt;
    g->cpt     = kGeoms[best].cpt;
    const int cap = g->threads / 64 * g->cpt;
    g->slices  = (n_chunks + cap - 1) / cap;
    // narrow cohort: two batch buffers, a wave builds its own plane-rows (wpp 1), batches are pipelined;
    // wide cohort (two buffers with enough rows do not fit): one buffer, teams of waves per plane-row
    {
        const int nt = g->threads;
        int K2 = want_K > 0 ? want_K : nt / 64;
        if (K2 > nt / 64) K2 = nt / 64;
        while (K2 > 1 && lds_need(nw, K2, G, nt, 2) > kLdsBytes) --K2;
        // a CU must hold 16 waves of these workgroups (4 per SIMD): rows per batch few enough that 1024 / nt of them share its
        // LDS -- but never fewer than 4 rows (m = 12,000, 512 x 24: K 8 = 97 KB, one workgroup per CU, 9.4 ms per 1 M sites;
        // K 6 = 73 KB, two per CU, 7.1 ms)
        if (want_K <= 0 && nt < 1024 && !team_only(nt, g->cpt)) {
            int K3 = K2;
            while (K3 > 4 && lds_need(nw, K3, G, nt, 2) > kLdsBytes * nt / 1024) --K3;
            if (lds_need(nw, K3, G, nt, 2) <= kLdsBytes * nt / 1024) K2 = K3;
        }
        if (!team_only(nt, g->cpt) && lds_need(nw, K2, G, nt, 2) <= kLdsBytes && wpp_for(nt, K2) == 1) { g->K = K2; g->nbuf = 2; g->wpp = 1; }
        else { g->K = best_K; g->nbuf = 1; g->wpp = wpp_for(nt, best_K); if (g->wpp == 1) g->wpp = 2; }
        if (g->nbuf == 1 && 2 * g->K * g->wpp > nt / 64) {            // a team per plane-row must exist
            while (g->K > 1 && 2 * g->K * g->wpp > nt / 64) --g->K;
            while (g->wpp > 1 && 2 * g->K * g->wpp > nt / 64) g->wpp /= 2;
        }
    }
    g->lds_bytes = (lds_need(nw, g->K, G, g->threads, g->nbuf) + 15) & ~15;
    // team mode: a separate toggle array (two barriers per batch instead of four) when the LDS has room for it
    g->tog_off = 0;
    if (g->nbuf == 1 && allow_tog) {
        const int tog_bytes = 2 * g->K * 4 * ((nw + 4) & ~3);
        if (g->lds_bytes + tog_bytes <= kLdsBytes) { g->tog_off = g->lds_bytes; g->lds_bytes += tog_bytes; }
    }
    g->workgroups = ((n_blk + 7) / 8) * 8 * g->slices;
    return true;
}

// one translation unit (scan_nt.hip) per geometry
#define X(C_) hipError_t launch_scan_nt256_c##C_(const ScanArgs &a, const Geometry &g, hipStream_t s);
BGTH_CPT_256(X)
#undef X
#define X(C_) hipError_t launch_scan_nt512_c##C_(const ScanArgs &a, const Geometry &g, hipStream_t s);
BGTH_CPT_512(X)
#undef X
#define X(C_) hipError_t launch_scan_nt1024_c##C_(const ScanArgs &a, const Geometry &g, hipStream_t s);
BGTH_CPT_1024(X)
#undef X
hipError_t launch_scan_wide(const ScanArgs &a, const Geometry &g, hipStream_t s);

hipError_t launch_scan(const ScanArgs &a, const Geometry &g, hipStream_t s)
{
    if (team_only(g.threads, g.cpt)) return launch_scan_wide(a, g, s);
#define X(C_) if (g.threads == 256 && g.cpt == C_) return launch_scan_nt256_c##C_(a, g, s);
    BGTH_CPT_256(X)
#undef X
#define X(C_) if (g.threads == 512 && g.cpt == C_) return launch_scan_nt512_c##C_(a, g, s);
    BGTH_CPT_512(X)
#undef X
#define X(C_) if (g.threads == 1024 && g.cpt == C_) return launch_scan_nt1024_c##C_(a, g, s);
    BGTH_CPT_1024(X)
#undef X
    return hipErrorInvalidConfiguration;
}

// ----------------------------------------------------------------------------------------------------
// small companion kernels
// ----------------------------------------------------------------------------------------------------
// reference bgt.c:745-756: AN = n0+n1+n3 = haplotypes - n2 ; AC = n1 ; AC<M> = n3 ; totals = sum of groups
__global__ void finalize_kernel(const int32_t *raw, int32_t *out, const int32_t *group_haps,
                                int64_t n_rows, int G)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const int Gx = G > 1 ? G : 0;
    const int32_t *src = raw + r * G * 3;
    int32_t *dst = out + r * (1 + Gx) * 3;
    int32_t an = 0, ac = 0, am = 0;
    for (int g = 0; g < G; ++g) {
        const int32_t gan = group_haps[g] - src[3 * g + 1], gac = src[3 * g], gam = src[3 * g + 2];
        an += gan; ac += gac; am += gam;
        if (Gx) { dst[3 * (1 + g)] = gan; dst[3 * (1 + g) + 1] = gac; dst[3 * (1 + g) + 2] = gam; }
    }
    dst[0] = an; dst[1] = ac; dst[2] = am;
}

hipError_t launch_finalize(const int32_t *raw, int32_t *out, const int32_t *group_haps, int64_t n_rows,
                           int G, hipStream_t s)
{
    if (n_rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, s,
                       raw, out, group_haps, n_rows, G);
    return hipGetLastError();
}

// reference pbwt.c:343: invS[S[i]] = i.  The permutations come from a file: an entry outside 0..m-1 must not become a
// store outside the image (HBM is shared with every other image and reader of the process), so it is dropped and
// reported; verify_inverse_kernel then proves the record was a permutation (inv was pre-filled with -1).
__global__ void invert_kernel(const int32_t *perm, int32_t *inv, int m, int64_t total, int *bad)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t base = i / m * m;
    const int32_t p = perm[i];
    if (p < 0 || p >= m) { if (bad) atomicOr(bad, 1); return; }
    inv[base + p] = (int32_t)(i - base);
}

// out[rec][rank0[rec][col]] = rank1[rec][col] -- the plane-1 ranks of every (sub-)checkpoint in the order of its plane-0 ranks, so
// that slots laid out in plane-0 rank order start without a gather.  Product path: the table behind ScanArgs::order0, which every
// whole-cohort counts scan uses (measured in profiles/r05_lds).  rank0[rec] must be a permutation (files: launch_invert checks the
// 'S' records; bgth_pbf_rebase checks its input), or slots of `out` stay unwritten.
__global__ void plane1_by_plane0_kernel(const int32_t *rank, int32_t *out, int m, int64_t total)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t rec = i / m;
    const int32_t col = (int32_t)(i - rec * m);
    out[rec * m + rank[rec * 2 * m + col]] = rank[rec * 2 * m + m + col];
}
// A sparse selection of a wide cohort (C3: 10,000 of 200,000 columns) starts every sub-block by gathering its columns' ranks out of
// the [2][m] checkpoint record -- 4 useful bytes per 64-byte line, for every workgroup of every launch (profiles/r05_c3: 7.25 GB
// fetched per launch against 2.3 GB compulsory).  Gathered ONCE per selection into slot order, the start of a workgroup is a
// contiguous read of T x 4 bytes.
__global__ void gather_start_ranks_kernel(const int32_t *__restrict__ rank, const int32_t *__restrict__ slot_col, int32_t *__restrict__ out,
                                          int m, int n_slots, int64_t total, int32_t pad)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t rp = i / n_slots;                                     // (record, plane)
    const int col = slot_col[(int)(i - rp * n_slots)];
    out[i] = col >= 0 ? rank[rp * m + col] : pad;
}
hipError_t launch_gather_start_ranks(const int32_t *rank, const int32_t *slot_col, int32_t *out, int m, int n_slots, int64_t n_rec, int32_t pad,
                                     hipStream_t s)
{
    const int64_t total = n_rec * 2 * n_slots;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_start_ranks_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, rank, slot_col, out, m, n_slots, total, pad);
    return hipGetLastError();
}

hipError_t launch_plane1_by_plane0(const int32_t *rank, int32_t *out, int m, int64_t n_rec, hipStream_t s)
{
    const int64_t total = n_rec * m;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(plane1_by_plane0_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, rank, out, m, total);
    return hipGetLastError();
}

__global__ void verify_inverse_kernel(const int32_t *perm, const int32_t *inv, int m, int64_t total, int *bad)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= total) return;
    const int64_t base = j / m * m;
    const int32_t i = inv[j];
    if (i < 0 || i >= m || perm[base + i] != (int32_t)(j - base)) atomicOr(bad, 2);
}

hipError_t launch_invert(const int32_t *perm, int32_t *inv, int m, int64_t n_perm, hipStream_t s, int *bad)
{
    const int64_t total = n_perm * m;
    if (total <= 0) return hipSuccess;
    if (bad) {
        hipError_t e = hipMemsetAsync(inv, 0xff, (size_t)total * 4, s);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(invert_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       perm, inv, m, total, bad);
    if (bad)
        hipLaunchKernelGGL(verify_inverse_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                           perm, inv, m, total, bad);
    return hipGetLastError();
}

// Composition of rank maps.  A row moves whatever sits at position R to LF(R) (reference pbwt.c:76-88): the ranks after a
// stretch of rows are a map of POSITIONS, so the ranks a stretch produces from the identity order (`table`: virtual column
// v started at position v) turn into the ranks it produces from any start order `via` by one gather per plane.
__global__ void compose_kernel(const int32_t *table, int64_t table_stride, const int32_t *via, int64_t via_stride, int32_t *out,
                               int64_t out_stride, int m, int64_t total)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t rec = i / (2 * m);
    const int k = (int)(i - rec * 2 * m), plane = k >= m;
    const int32_t pos = via[rec * via_stride + k];
    out[rec * out_stride + k] = table[rec * table_stride + (int64_t)plane * m + pos];
}

hipError_t launch_compose(const int32_t *table, int64_t table_stride, const int32_t *via, int64_t via_stride, int32_t *out,
                          int64_t out_stride, int m, int64_t n, hipStream_t s)
{
    const int64_t total = n * 2 * m;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(compose_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       table, table_stride, via, via_stride, out, out_stride, m, total);
    return hipGetLastError();
}

// 2-bit codes a1<<1|a0 of output column i at bits 2*(i&3) of byte i>>2 (feeds bgt.c:306-311)
__global__ void pack2_kernel(const uint64_t *h0, const uint64_t *h1, const int32_t *slot_of_out,
                             uint8_t *gt, int64_t n_rows, int n_chunks, int width)
{
    const int nb = (width + 3) / 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * nb) return;
    const int64_t row = i / nb;
    const int b = (int)(i - row * nb);
    const uint64_t *p0 = h0 + row * n_chunks, *p1 = h1 + row * n_chunks;
    uint32_t v = 0;
    for (int k = 0; k < 4; ++k) {
        const int col = 4 * b + k;
        if (col < width) {
            const int s = slot_of_out[col];
            const uint32_t a0 = (uint32_t)(p0[s >> 6] >> (s & 63)) & 1u;
            const uint32_t a1 = (uint32_t)(p1[s >> 6] >> (s & 63)) & 1u;
            v |= (a1 << 1 | a0) << (2 * k);
        }
    }
    gt[i] = (uint8_t)v;
}

hipError_t launch_pack2(const uint64_t *h0, const uint64_t *h1, const int32_t *slot_of_out, uint8_t *gt,
                        int64_t n_rows, int n_chunks, int width, hipStream_t s)
{
    const int64_t total = n_rows * ((width + 3) / 4);
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(pack2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       h0, h1, slot_of_out, gt, n_rows, n_chunks, width);
    return hipGetLastError();
}

__global__ void unpack_bytes_kernel(const uint64_t *h0, const uint64_t *h1, const int32_t *slot_of_out,
                                    uint8_t *a0, uint8_t *a1, int64_t n_rows, int n_chunks, int width)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * width) return;
    const int64_t row = i / width;
    const int col = (int)(i - row * width);
    const int s = slot_of_out[col];
    a0[i] = (uint8_t)((h0[row * n_chunks + (s >> 6)] >> (s & 63)) & 1u);
    a1[i] = (uint8_t)((h1[row * n_chunks + (s >> 6)] >> (s & 63)) & 1u);
}

hipError_t launch_unpack_bytes(const uint64_t *h0, const uint64_t *h1, const int32_t *slot_of_out,
                               uint8_t *a0, uint8_t *a1, int64_t n_rows, int n_chunks, int width,
                               hipStream_t s)
{
    const int64_t total = n_rows * width;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_bytes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       h0, h1, slot_of_out, a0, a1, n_rows, n_chunks, width);
    return hipGetLastError();
}

// The genotype vector bgt_gen_gt builds (reference bgt.c:290-313, table bgt_bits2gt bgt.c:250) and its VCF text
// (reference vcf.c:940-969 for a GT-only FORMAT), straight from the bit planes: per haplotype i of the output
//   gt8[i]  = (allele + 1) << 1 = {2, 4, 0, 6}[code]           code = a1 << 1 | a0; 0 REF, 1 ALT, 2 missing, 3 <M>
//   text    = '\t' | '/' (first | second haplotype of the sample) then '0' '1' '.' '2'
// One thread per sample (two haplotypes): 2 bytes of gt8, 4 characters of text.
__global__ void emit_gt_kernel(const uint64_t *h0, const uint64_t *h1, const int32_t *slot_of_out, uint8_t *gt8,
                               uint32_t *text, int64_t n_rows, int n_chunks, int n_samples)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * n_samples) return;
    const int64_t row = i / n_samples;
    const int smp = (int)(i - row * n_samples);
    uint32_t code[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int s = slot_of_out[2 * smp + k];
        const uint32_t a0 = (uint32_t)(h0[row * n_chunks + (s >> 6)] >> (s & 63)) & 1u;
        const uint32_t a1 = (uint32_t)(h1[row * n_chunks + (s >> 6)] >> (s & 63)) & 1u;
        code[k] = a1 << 1 | a0;
    }
    if (gt8) {
        const uint32_t tab = 0x06000402u;                      // bytes {2, 4, 0, 6}
        reinterpret_cast<uint16_t*>(gt8)[i] = (uint16_t)(((tab >> (8 * code[0])) & 255u) | (((tab >> (8 * code[1])) & 255u) << 8));
    }
    if (text) {
        const uint32_t chr = 0x322e3130u;                      // bytes "01.2"
        text[i] = (uint32_t)'\t' | ((chr >> (8 * code[0])) & 255u) << 8 | (uint32_t)'/' << 16 | ((chr >> (8 * code[1])) & 255u) << 24;
    }
}

hipError_t launch_emit_gt(const uint64_t *h0, const uint64_t *h1, const int32_t *slot_of_out, uint8_t *gt8,
                          uint32_t *text, int64_t n_rows, int n_chunks, int width, hipStream_t s)
{
    const int64_t total = n_rows * (width / 2);
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(emit_gt_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       h0, h1, slot_of_out, gt8, text, n_rows, n_chunks, width / 2);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------------------------
// Allele-set reductions (reference bgt.c:859-876, `bgt view -a ... -S / -H`): one decoded row folded into the two
// per-reader accumulators.  One thread per output sample; both arrays live in HBM across the rows of a query.
//   carriers[s] += 1 if either haplotype of sample s has code `code` (1 = the allele, 0 = a reference-allele query)
//   hap[2s+k]   |= 1 << bit  if haplotype k of sample s has code 1
// ----------------------------------------------------------------------------------------------------
__global__ void fold_alleles_kernel(const uint64_t *h0, const uint64_t *h1, const int32_t *slot_of_out, int32_t *carriers,
                                    uint64_t *hap, int n_samples, int code, int bit)
{
    const int smp = blockIdx.x * blockDim.x + threadIdx.x;
    if (smp >= n_samples) return;
    uint32_t c[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int s = slot_of_out[2 * smp + k];
        const uint32_t a0 = (uint32_t)(h0[s >> 6] >> (s & 63)) & 1u;
        const uint32_t a1 = (uint32_t)(h1[s >> 6] >> (s & 63)) & 1u;
        c[k] = a1 << 1 | a0;
    }
    if (carriers && code >= 0) carriers[smp] += (c[0] == (uint32_t)code || c[1] == (uint32_t)code);
    if (hap && bit >= 0) {
        uint4 *hp = reinterpret_cast<uint4*>(hap) + smp;                  // the sample's two 64-bit signatures in one access
        uint4 v = *hp;
        const uint32_t lo = bit < 32 ? 1u << bit : 0u, hi = bit < 32 ? 0u : 1u << (bit - 32);
        if (c[0] == 1u) { v.x |= lo; v.y |= hi; }
        if (c[1] == 1u) { v.z |= lo; v.w |= hi; }
        *hp = v;
    }
}

hipError_t launch_fold_alleles(const uint64_t *h0_row, const uint64_t *h1_row, const int32_t *slot_of_out, int32_t *carriers,
                               uint64_t *hap, int width, int code, int bit, hipStream_t s)
{
    const int n_samples = width / 2;
    if (n_samples <= 0) return hipSuccess;
    hipLaunchKernelGGL(fold_alleles_kernel, dim3((unsigned)((n_samples + 255) / 256)), dim3(256), 0, s,
                       h0_row, h1_row, slot_of_out, carriers, hap, n_samples, code, bit);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------------------------
// Site filter on the device: the reverse-Polish program of a `-f` expression (exported by the host parser,
// filter_expr.c ke_export) evaluated per site on the counts the scan produced.  Same value model as the host
// evaluator (reference kexpr.c:105-153): every slot carries an int64 and a double view plus a type;
// `/` yields a real, `//` `%` `<<` `>>` `&` `|` `^` integers, comparisons use the reals if either side is
// real, an unbound variable fails the site.
// ----------------------------------------------------------------------------------------------------
struct FilterSlot { long long i; double r; long long real; };   // (no padding bytes: a bool here left 7 of them per slot in memory, and the slots with them)

// one unary (+ - ~ !) or binary operator on the top of the stack: p = p OP q.  Values in, value out (no references): the
// four named stack slots of the SMALL kernel live in registers.  (With a `bool` in the slot its 7 padding bytes stayed in
// memory and the compiler promoted them to LDS, 28 KB per workgroup -- and a launch that leaves the LDS allocator in that
// state keeps the plane-split kernels, which need two 75 KB workgroups per CU, at one per CU for half their run: C3 took
// 23.5 instead of 19.9 ms whenever the device filter ran between two scans; `make resource-usage` shows the LDS size.)
__device__ __forceinline__ FilterSlot filter_unop(int o, FilterSlot p)
{
    if (o == 2) { p.i = -p.i; p.r = -p.r; }
    else if (o == 3) { p.i = ~p.i; p.r = (double)p.i; p.real = false; }
    else if (o == 4) { p.i = !p.i; p.r = (double)p.i; p.real = false; }
    return p;
}

__device__ __forceinline__ FilterSlot filter_binop(int o, FilterSlot p, const FilterSlot q, bool &err)
{
    const bool anyreal = p.real || q.real;
    bool cmp = false, iscmp = false;
    switch (o) {
    case 5:  p.r = pow(p.r, q.r); p.i = (long long)(p.r + .5); p.real = anyreal; break;
    case 6:  p.i *= q.i; p.r *= q.r; p.real = anyreal; break;
    case 7:  p.r /= q.r; p.i = (long long)(p.r + .5); p.real = true; break;
    case 8:  if (q.i == 0) { err = true; p.i = 0; } else p.i /= q.i; p.r = (double)p.i; p.real = false; break;
    case 9:  if (q.i == 0) { err = true; p.i = 0; } else p.i %= q.i; p.r = (double)p.i; p.real = false; break;
    case 10: p.i += q.i; p.r += q.r; p.real = anyreal; break;
    case 11: p.i -= q.i; p.r -= q.r; p.real = anyreal; break;
    case 12: p.i <<= q.i; p.r = (double)p.i; p.real = false; break;
    case 13: p.i >>= q.i; p.r = (double)p.i; p.real = false; break;
    case 14: iscmp = true; cmp = anyreal ? p.r <  q.r : p.i <  q.i; break;
    case 15: iscmp = true; cmp = anyreal ? p.r <= q.r : p.i <= q.i; break;
    case 16: iscmp = true; cmp = anyreal ? p.r >  q.r : p.i >  q.i; break;
    case 17: iscmp = true; cmp = anyreal ? p.r >= q.r : p.i >= q.i; break;
    case 18: iscmp = true; cmp = anyreal ? p.r == q.r : p.i == q.i; break;
    case 19: iscmp = true; cmp = anyreal ? p.r != q.r : p.i != q.i; break;
    case 20: p.i &= q.i; p.r = (double)p.i; p.real = false; break;
    case 21: p.i ^= q.i; p.r = (double)p.i; p.real = false; break;
    case 22: p.i |= q.i; p.r = (double)p.i; p.real = false; break;
    case 23: iscmp = true; cmp = p.i && q.i; break;
    case 24: iscmp = true; cmp = p.i || q.i; break;
    default: err = true; break;
    }
    if (iscmp) { p.i = cmp; p.r = (double)cmp; p.real = false; }
    return p;
}

// SMALL: the program never holds more than four values and leaves exactly one (checked by launch_filter): the stack
// is four named slots shifted on push / pop, i.e. registers instead of a dynamically indexed array in scratch memory
template <bool SMALL>
__global__ void filter_kernel(const FilterProgram prog, const int32_t *counts, int64_t n_rows, int ints_per_row,
                              uint8_t *flags, unsigned long long *n_pass)
{
    unsigned long long passed = 0;                               // (wave-uniform)
    for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row - threadIdx.x < n_rows; row += (int64_t)gridDim.x * blockDim.x) {
    bool pass = false;
    if (row < n_rows) {
        const int32_t *c = counts + row * ints_per_row;
        bool err = false;
        if (SMALL) {
            FilterSlot s0 = {0, 0., false}, s1 = s0, s2 = s0, s3 = s0;        // s0 = top of the stack
            for (int k = 0; k < prog.n; ++k) {
                const int op = prog.op[k];
                if (op <= 2) {
                    FilterSlot x;
                    if (op == 2) {
                        const int slot = prog.slot[k];
                        if (slot < 0 || slot >= ints_per_row) { err = true; x.i = 0; x.r = 0.; x.real = true; }
                        else { x.i = c[slot]; x.r = (double)c[slot]; x.real = false; }
                    } else { x.i = prog.ival[k]; x.r = prog.rval[k]; x.real = op == 1; }   // a literal keeps BOTH views ("010" is 8 and 10.0)
                    s3 = s2; s2 = s1; s1 = s0; s0 = x;
                } else {
                    const int o = op - 16;
                    if (o >= 1 && o <= 4) s0 = filter_unop(o, s0);
                    else { s0 = filter_binop(o, s1, s0, err); s1 = s2; s2 = s3; }
                }
            }
            pass = !err && s0.i != 0;
        } else {
            FilterSlot st[kFilterMaxItems];
            int top = 0;
            for (int k = 0; k < prog.n; ++k) {
                const int op = prog.op[k];
                if (op == 0 || op == 1) { st[top].i = prog.ival[k]; st[top].r = prog.rval[k]; st[top].real = op == 1; ++top; }
                else if (op == 2) {
                    const int slot = prog.slot[k];
                    if (slot < 0 || slot >= ints_per_row) { err = true; st[top].i = 0; st[top].r = 0.; st[top].real = true; }
                    else { st[top].i = c[slot]; st[top].r = (double)c[slot]; st[top].real = false; }
                    ++top;
                } else {
                    const int o = op - 16;
                    if (o >= 1 && o <= 4) st[top - 1] = filter_unop(o, st[top - 1]);
                    else { --top; st[top - 1] = filter_binop(o, st[top - 1], st[top], err); }
                }
            }
            pass = !err && top >= 1 && st[0].i != 0;
        }
        flags[row] = pass ? 1 : 0;
    }
    passed += (unsigned long long)__popcll(__ballot(pass));
    }
    if ((threadIdx.x & 63) == 0 && passed) atomicAdd(n_pass, passed);
}

hipError_t launch_filter(const FilterProgram &prog, const int32_t *counts, int64_t n_rows, int ints_per_row,
                         uint8_t *flags, unsigned long long *n_pass, hipStream_t s)
{
    if (n_rows <= 0) return hipSuccess;
    int depth = 0, deepest = 0;                                 // stack profile of the program
    bool regular = true;
    for (int k = 0; k < prog.n; ++k) {
        const int op = prog.op[k];
        if (op <= 2) { if (++depth > deepest) deepest = depth; }
        else if (op - 16 >= 1 && op - 16 <= 4) { if (depth < 1) regular = false; }
        else { if (depth < 2) regular = false; --depth; }
    }
    // a bounded grid walked in strides: a million rows as 4 k workgroups of one row per thread kept the dispatcher busy into
    // the start of the next scan
    const dim3 grid((unsigned)std::min<int64_t>((n_rows + 255) / 256, 1024)), block(256);
    if (regular && depth == 1 && deepest <= 4)
        hipLaunchKernelGGL(filter_kernel<true>, grid, block, 0, s, prog, counts, n_rows, ints_per_row, flags, n_pass);
    else
        hipLaunchKernelGGL(filter_kernel<false>, grid, block, 0, s, prog, counts, n_rows, ints_per_row, flags, n_pass);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------------------------
// PMC calibration aid: stream a buffer of known size with the access width of the scan kernel (one
// dword per lane) or with 16-byte loads, so that rocprofv3's FETCH_SIZE can be scaled to real bytes
// (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern").
// ----------------------------------------------------------------------------------------------------
template <typename T>
__global__ void stream_read_kernel(const T *src, size_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = src[i];
        const uint32_t *w = reinterpret_cast<const uint32_t*>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; ++k) acc ^= w[k];
    }
    if (acc == 0x9e3779b9u) *sink = acc;          // keeps the loads alive
}

hipError_t launch_stream_read(const void *src, size_t bytes, int width, uint32_t *sink, hipStream_t s)
{
    if (width == 16) hipLaunchKernelGGL(stream_read_kernel<uint4>, dim3(2048), dim3(256), 0, s, (const uint4*)src, bytes / 16, sink);
    else hipLaunchKernelGGL(stream_read_kernel<uint32_t>, dim3(2048), dim3(256), 0, s, (const uint32_t*)src, bytes / 4, sink);
    return hipGetLastError();
}

}  // namespace bgth
