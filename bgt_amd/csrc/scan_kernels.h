// Internal interface between the C-ABI host code (bgt_hip.cpp) and the gfx950 kernels (scan_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bgth {

// Row descriptor: byte offset of the RLE string in the packed stream (low 40 bits) | length (high 24).
constexpr int      kDescLenShift = 40;
constexpr uint64_t kDescOffMask  = (1ull << kDescLenShift) - 1;

// Row index (wide cohorts, built once per file by rowindex_kernel):
//   chunkinfo[slot]  one record per 256-byte chunk c of string i, slot = ((offset_i + 256 c) >> 8) + i
//                    (injective because strings do not overlap):  start position of the chunk in the row
//                    (bits 0..29) | kChunkDead if the chunk lies behind a terminating zero byte | bit of the
//                    last byte before the chunk << 31
//   segc[i][s]       s = 0..S8-1, S8 = ceil(m / 8192): ones before position 8192 s (bits 0..30) | bit at
//                    position 8192 s - 1 << 31 -- the carries a directory trip of 256 words starts from;
//   segc[i][S8]      number of ones in the row
constexpr uint32_t kChunkPosMask = 0x3fffffffu;
constexpr uint32_t kChunkDead    = 0x40000000u;

// Chunk descriptor (one per 64 tracked slots): group id (bits 0..7, 0-based) | valid slots (bits 8..14).
// Slots are laid out group by group, each group padded to a multiple of 64, so a 64-slot chunk never
// mixes groups and the per-chunk ballot/popcount reduction needs no per-lane group lookup.

struct ScanArgs {
    const uint8_t  *rle;         // packed RLE strings of the whole file
    const uint64_t *rowdesc;     // [2*n_rows]
    const int32_t  *rank0;       // initial ranks by column: [blk][2][m] (blk_stride = 2*m) or one [2][m]
    int64_t         rank0_blk_stride;
    const int32_t  *slot_col;    // [n_chunks*64] column of each tracked slot, -1 = padding
    const int32_t  *order0;      // whole-cohort counts scans (the default of every such scan; measured in profiles/r05_lds): != NULL = slot s
    int64_t         order_blk_stride;   //   of a sub-block tracks the column of plane-0 rank s at its checkpoint; order0[blk * stride + s] =
                                 //   that column's plane-1 rank (the image's d_order table, built at the first such scan)
    int32_t         pk16;        // profiling build only (round-6 A/B, profiles/r06_pk16): with whole_counts, m <= 65,504, a column's two ranks in one register
    const int32_t  *start_slots; // plane-split kernels, != NULL: the start ranks of the reader's OWN slots, compact -- [sub-block][plane][n_chunks * 64]
    int64_t         start_blk_stride;   //   (= 2 n_chunks 64), gathered once per selection (launch_gather_start_ranks); padding slots hold 32 nw
    const uint32_t *chunk_desc;  // [n_chunks]
    int32_t        *raw_counts;  // [(row1-row0)][G][3] += {n(code1), n(code2), n(code3)}
    uint64_t       *h0, *h1;     // optional [(row1-row0)][n_chunks] bit planes in slot order
    int64_t         h_row0;      // plane-split kernels only: the row h0 / h1 start at (row0, or the first emitted row of a pass)
    int32_t        *final_rank;  // optional [2][m]: ranks by column after the last row of the launch; with final_blk_stride != 0
    int64_t         final_blk_stride;   //   one record per block of the launch: [n_blk][2][m] at final_rank + bl * stride
    int32_t        *snap;        // optional: sub-checkpoints, ranks by column BEFORE every row that is a multiple of
    int32_t         snap_shift;  //   1 << snap_shift (block starts excepted), at snap[(row >> snap_shift) * 2m]
    const uint32_t *chunkinfo;   // row index (see above); only read by the team (wide-cohort) kernels
    const uint32_t *segc;
    int32_t  m, nw, shift, n_chunks, G, K, wpp, nbuf, S8;
    int32_t  zp;                 // use the kernels with the all-zero-plane-1 shortcut (the image has such rows)
    int32_t  whole_counts;       // whole cohort, one group, no bit planes: kernels may count n(code 3) only and take the planes' ones from the strings
    int32_t  walk_prio;          // team kernels: progress-based wave priorities in the walk (the profiling build can switch them off)
    int32_t  tog_off;            // team mode: byte offset in LDS of the separate toggle array [2K][(nw+4)&~3], 0 = toggles in place
    int32_t  blk0, n_blk, n_slices;
    int64_t  row0, row1;         // rows whose results are emitted; decoding starts at blk0<<shift
    // Directory path (scan_dir.hip; wide cohorts whose columns span several workgroups): dirbuild_kernel writes the
    // {bits, ones before} entries of every plane-row ONCE into an HBM arena, walk_kernel pulls them into LDS with LDS-DMA
    // and only walks.  Plane-row (row, plane) occupies dir_nwp entries at dir[(2 (row - dir_row0) + plane) dir_nwp]:
    // nw entries, an all-zero sentinel, padding to 16 bytes; dir_n0[2 (row - dir_row0) + plane] = its number of zeros.
    uint2    *dir;
    uint32_t *dir_n0;
    uint32_t *tog_mem;           // cohorts beyond 650,000 haplotypes (dirbuild_mem_kernel): toggle words in memory, [workgroups][2][nwt]
    int64_t   dir_row0;
    int32_t   dir_nwp;
    int32_t   dir_stage;         // bit 2: FOUR plane buffers (rows double-buffered, one barrier per row); bit 0: three plane buffers in LDS (plane 0 of the next row lands while this row is walked; else
                                 // two); bit 1: touch the next row's plane 1 during the walk so that its DMA finds it in the L2
    // Profiling builds only (make ABLATE=1 -> libbgt_hip_ablate.so, used by scripts/profile.sh): the shipped library
    // compiles every one of these switches out (BGTH_SKIP / BGTH_TIMES below are constant 0) and never reads the
    // environment variables that set them -- an ablation switch makes the kernels return wrong numbers faster.
    unsigned long long *debug_times;   // [8] cycle sums per phase
    int32_t  debug_skip;         // ablation bits: 1 no walk, 2 no RLE read / toggles, 4 no directory build,
                                 //   8 every lookup reads the sentinel word (no LDS conflicts), 64 no toggle atomics, 0x100 timing
                                 //   of one wave only, 0x2000 no priority rotation in the walk, 0x4000 no raised priority for the build
};

#ifdef BGTH_ABLATE
#define BGTH_SKIP(a, bit) ((a).debug_skip & (bit))
#define BGTH_TIMES(a)     ((a).debug_times != nullptr)
#define BGTH_WALK_PRIO(a) ((a).walk_prio != 0)
#define BGTH_DIR_PRIO(a)  (!((a).dir_stage & 8))
#else
#define BGTH_SKIP(a, bit) 0
#define BGTH_TIMES(a)     false
#define BGTH_WALK_PRIO(a) true          // (only the profiling build can switch the progress-based wave priorities off: a run-time
#define BGTH_DIR_PRIO(a)  true          //  flag was re-tested through a VGPR several times per row)
#endif

// columns per thread instantiated for each workgroup size (keep in sync with kGeoms in scan_kernels.hip)
#define BGTH_CPT_256(X)  X(2) X(4) X(8) X(12) X(16) X(20)
#define BGTH_CPT_512(X)  X(4) X(8) X(10) X(12) X(16) X(20) X(24) X(32) X(40) X(48)
#define BGTH_CPT_1024(X) X(4) X(8) X(10) X(12) X(16) X(20) X(24) X(28) X(32) X(36) X(40)
// team kernels only (wide cohorts: as many columns per workgroup as the 256 VGPRs of a 512-thread
// workgroup hold, so that few column slices repeat the per-row bit-vector build)
#define BGTH_CPT_512_WIDE(X) X(64) X(80) X(98)

struct Geometry { int threads, cpt, slices, K, lds_bytes, workgroups, wpp, nbuf, tog_off; int dir_stage = -1; int low = 0; };   // dir_stage >= 0: directory path; low: the plane-split kernels at <= 80 VGPRs (six waves per SIMD)

// Picks threads/columns-per-thread/slices/K for a selection of n_chunks*64 slots over n_blk blocks.
// Returns false if the row bit-vectors of this m cannot fit in LDS.
bool choose_geometry(int m, int n_chunks, int G, int n_blk, int want_threads, int want_cpt, int want_K,
                     Geometry *g, bool allow_tog = true);
hipError_t launch_scan(const ScanArgs &a, const Geometry &g, hipStream_t s);
// directory path: geometry of the walk-only kernel (false: one plane-row pair does not fit the LDS), the producer over the
// plane-rows [2 row_lo, 2 row_hi) of the image into a.dir / a.dir_n0 (needs the row index), and the walk itself
bool choose_walk_geometry(int m, int n_chunks, int G, int n_blk, int want_threads, int want_cpt, Geometry *g);
hipError_t launch_dirbuild(const ScanArgs &a, int64_t row_lo, int64_t row_hi, hipStream_t s);
hipError_t launch_walk(const ScanArgs &a, const Geometry &g, hipStream_t s);
// row index of n_str strings (see above); chunkinfo must hold packed_bytes/256 + n_str + 1 records
hipError_t launch_rowindex(const uint64_t *rowdesc, const uint8_t *rle, int64_t n_str, int m, int S8,
                           uint32_t *chunkinfo, uint32_t *segc, hipStream_t s);

// raw {c1,c2,c3} per group -> {AN,AC,AC<M>} for total (+ per group when G>1)
hipError_t launch_finalize(const int32_t *raw, int32_t *out, const int32_t *group_haps, int64_t n_rows,
                           int G, hipStream_t s);
// inv[perm[j]] = j for n_perm permutations of m entries each.  bad != NULL (untrusted input): entries outside
// 0..m-1 are not stored and every record is checked to be a permutation; *bad (device int) becomes non-zero otherwise
hipError_t launch_plane1_by_plane0(const int32_t *rank, int32_t *out, int m, int64_t n_rec, hipStream_t s);
// out[rec][plane][slot] = rank[rec][plane][slot_col[slot]] (pad where slot_col < 0): a selection's start ranks at every (sub-)checkpoint
hipError_t launch_gather_start_ranks(const int32_t *rank, const int32_t *slot_col, int32_t *out, int m, int n_slots, int64_t n_rec, int32_t pad,
                                     hipStream_t s);
hipError_t launch_invert(const int32_t *perm, int32_t *inv, int m, int64_t n_perm, hipStream_t s, int *bad = nullptr);
// plane-split kernels (scan_plane.hip; sparse selections of wide cohorts): one workgroup per (sub-block, plane), two per CU;
// the planes meet in count_planes (raw[row][g][3] = the three popcounts per group from the bit planes h0 / h1)
bool choose_plane_geometry(int m, int n_chunks, int n_blk, Geometry *g);
int plane_slots_per_cu(int m);   // workgroups of the plane-split kernels a CU holds (2 or 3)
hipError_t launch_plane_scan(const ScanArgs &a, const Geometry &g, hipStream_t s);
hipError_t launch_count_planes(const uint64_t *h0, const uint64_t *h1, const uint32_t *chunk_desc, int32_t *raw, int64_t n_rows,
                               int n_chunks, int G, hipStream_t s);
// walk-only plane kernels over the directory arena (scan_plane.hip; cohorts whose two bit-vectors do not fit the LDS together:
// 327,000 < m <= 650,000): one workgroup per (sub-block, column slice, plane)
bool choose_walk_plane_geometry(int m, int n_chunks, int n_blk, Geometry *g);
hipError_t launch_walk_plane(const ScanArgs &a, const Geometry &g, hipStream_t s);
// ... and cohorts whose ONE bit-vector with its rank directory (m / 4 bytes) does not fit the LDS either (more than 650,000
// haplotypes): the producer keeps its toggle words in memory (a.tog_mem: dirbuild_mem_workgroups() x 2 x dirbuild_mem_words(m)
// words) and the walk gathers its entries from the arena in memory (L2) instead of the LDS.  Any m.
void choose_walk_mem_geometry(int m, int n_chunks, int n_blk, Geometry *g);
int64_t dirbuild_mem_workgroups(int64_t n_rows);
int64_t dirbuild_mem_words(int m);
hipError_t launch_dirbuild_mem(const ScanArgs &a, int64_t row_lo, int64_t row_hi, hipStream_t s);
hipError_t launch_walk_mem(const ScanArgs &a, const Geometry &g, hipStream_t s);
// out[i][p][c] = table[i * table_stride][p][ via[i * via_stride][p][c] ] for n records of [2][m] ranks: the composition of rank maps
// behind the parallel checkpoint derivation (bgt_hip.cpp: from_rle_impl, bgth_pbf_rebase)
hipError_t launch_compose(const int32_t *table, int64_t table_stride, const int32_t *via, int64_t via_stride, int32_t *out,
                          int64_t out_stride, int m, int64_t n, hipStream_t s);
// slot-ordered bit planes -> 2-bit codes in output-column order (4 per byte)
hipError_t launch_pack2(const uint64_t *h0, const uint64_t *h1, const int32_t *slot_of_out, uint8_t *gt,
                        int64_t n_rows, int n_chunks, int width, hipStream_t s);
// slot-ordered bit planes -> byte-per-column planes a0/a1 in output order (what pbf_read returns)
hipError_t launch_unpack_bytes(const uint64_t *h0, const uint64_t *h1, const int32_t *slot_of_out,
                               uint8_t *a0, uint8_t *a1, int64_t n_rows, int n_chunks, int width,
                               hipStream_t s);

// slot-ordered bit planes -> BCF GT bytes (1 per haplotype) and/or VCF GT text (2 characters per haplotype), output
// order; width must be even (haplotype pairs = samples); either output may be NULL
hipError_t launch_emit_gt(const uint64_t *h0, const uint64_t *h1, const int32_t *slot_of_out, uint8_t *gt8,
                          uint32_t *text, int64_t n_rows, int n_chunks, int width, hipStream_t s);
hipError_t launch_fold_alleles(const uint64_t *h0_row, const uint64_t *h1_row, const int32_t *slot_of_out, int32_t *carriers,
                               uint64_t *hap, int width, int code, int bit, hipStream_t s);

// compiled `-f` expression (reverse Polish): op 0 = int constant, 1 = real constant, 2 = variable read from
// counts[slot], 16 + k = operator k in the numbering of filter_expr.c
constexpr int kFilterMaxItems = 48;
struct FilterProgram { int32_t n; int32_t op[kFilterMaxItems]; int32_t slot[kFilterMaxItems]; long long ival[kFilterMaxItems]; double rval[kFilterMaxItems]; };
hipError_t launch_filter(const FilterProgram &prog, const int32_t *counts, int64_t n_rows, int ints_per_row,
                         uint8_t *flags, unsigned long long *n_pass, hipStream_t s);
hipError_t launch_stream_read(const void *src, size_t bytes, int width, uint32_t *sink, hipStream_t s);
// issue-rate calibration (microbench.hip, linked into libbgt_hip_bench.so only): out = {cycles of the slowest wave, ms, VALU wave-instr per wave, LDS wave-instr per wave}
hipError_t run_issue_rate(int mix, int waves_per_simd, int iters, double out[4]);
const char *issue_rate_mix_name(int mix);
hipError_t run_op_rate(int op, int waves_per_simd, int iters, double out[3]);
const char *op_rate_name(int op);

}  // namespace bgth
