// Directory path for wide cohorts (m = 200,000: a row's two bit-vectors with their rank directory are 100 KB).
//
// The team kernels of scan_wide.hip rebuild every row's {bits, ones before} directory in each of the column slices
// that share a sub-block (4 at m = 200,000): a third of their VALU instructions and most of their barrier time.
// Here the row is built ONCE:
//
//   dirbuild_kernel   RLE string -> toggles (LDS) -> directory entries, written to an HBM arena.  Every plane-row is an
//                     independent unit of work (its chunk positions and trip carries come from the row index), so the
//                     kernel runs with small workgroups at high occupancy instead of inside a 256-VGPR walker.
//   walk_kernel       workgroup = (sub-block, column slice) as before; the ranks of its columns stay in VGPRs, the row
//                     comes from the arena by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no VALU) into one of three
//                     50 KB plane buffers: plane 0 of row r+1 lands while row r is walked, plane 1 of row r+1 goes to
//                     plane 1 of row r's buffer after the barrier that ends the walk.  No toggle array, no build.
//
// Same arithmetic as the other kernels (reference pbwt.c:69-90, 129-170; bgt.c:735-757); the row step is the
// hand-scheduled statement of scan_device.inc.h.
#include "scan_device.inc.h"

namespace bgth {

static const int kLdsBytesDir = 160 * 1024;
static const int kSoloWords = 3200;                  // plane-rows of up to 102,400 positions are built by one wave each (dirbuild_solo_kernel)

// ----------------------------------------------------------------------------------------------------
// producer
// ----------------------------------------------------------------------------------------------------
// Workgroup = team of NT/64 waves over a contiguous range of plane-rows; two toggle arrays alternate so that one
// barrier per plane-row suffices:  toggles(i) | barrier | directory(i) (reads and clears the array) ; toggles(i+1) ...
template <int NT>
__global__ __launch_bounds__(NT) void dirbuild_kernel(const ScanArgs a, const uint64_t *__restrict__ rowdesc,
                                                      const uint8_t *__restrict__ rle, const uint32_t *__restrict__ chunkinfo,
                                                      const uint32_t *__restrict__ segc, int64_t str_lo, int64_t str_hi, int per_wg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WPP = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int tw = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = a.m, nw = a.nw, nwp = a.dir_nwp;
    const int nwt = (nw + 4) & ~3;
    uint32_t *TOG = reinterpret_cast<uint32_t*>(smem);                   // [2][nwt]
    for (int i = tid; i < 2 * nwt; i += NT) TOG[i] = 0u;
    lds_barrier();
    const uint32_t tail_mask = (m & 31) ? ((1u << (m & 31)) - 1u) : 0xffffffffu;
    const int ntrip = (nw + 255) >> 8;
    const int64_t s0 = str_lo + (int64_t)blockIdx.x * per_wg;
    int64_t s1 = s0 + per_wg;
    if (s1 > str_hi) s1 = str_hi;

    // What a plane-row's toggles read from memory -- this wave's first chunk of the string, that chunk's row-index record, the
    // carries of its directory trips and the row's ones (lane 63) -- is fetched a plane-row ahead, its descriptor two ahead:
    // the directory stores of the row before then cover the latency instead of a chain descriptor -> string -> decode.
    struct Ahead { uint32_t w, ci, cyl; };
    auto load_ahead = [&](int64_t sidx, uint64_t d) -> Ahead {
        Ahead p = {0u, 0u, 0u};
        if (sidx >= s1) return p;
        const uint32_t slen = (uint32_t)(d >> kDescLenShift);
        const uint64_t off = d & kDescOffMask;
        const uint32_t *sc = segc + (size_t)sidx * (size_t)(a.S8 + 1);
        const int t = tw + lane * WPP;                                  // (ntrip <= 40 < 63 WPP: lane 63 has no trip)
        if (t < ntrip || lane == 63) p.cyl = sc[lane == 63 ? a.S8 : t];
        const uint32_t k0 = (uint32_t)tw * 256u + 4u * (uint32_t)lane;
        if (k0 < slen) p.w = reinterpret_cast<const uint32_t*>(rle + off)[tw * 64 + lane];
        if ((uint32_t)tw * 256u < slen) p.ci = chunkinfo[((off + (uint64_t)tw * 256u) >> 8) + (uint64_t)sidx];
        return p;
    };
    uint64_t desc = s0 < s1 ? rowdesc[s0] : 0ull;
    uint64_t desc1 = s0 + 1 < s1 ? rowdesc[s0 + 1] : 0ull;
    Ahead cur = load_ahead(s0, desc);
    for (int64_t sidx = s0; sidx < s1; ++sidx) {
        uint32_t *trow = TOG + (size_t)((sidx - s0) & 1) * nwt;
        const uint64_t cd0 = desc;
        const Ahead here = cur;
        desc = desc1;
        cur = load_ahead(sidx + 1, desc);
        desc1 = sidx + 2 < s1 ? rowdesc[sidx + 2] : 0ull;
        const uint32_t slen = (uint32_t)(cd0 >> kDescLenShift);
        const uint64_t off = cd0 & kDescOffMask;
        const uint32_t cyl = lane == 63 ? 0u : here.cyl;
        const uint32_t tot1 = (uint32_t)__builtin_amdgcn_readlane((int)here.cyl, 63);
        if (!BGTH_SKIP(a, 0x200000))                                      // (profiling build: the producer without its toggles, timing only)
        for (int c = tw; (uint32_t)c * 256u < slen; c += WPP) {
            const uint32_t k0 = (uint32_t)c * 256u + 4u * (uint32_t)lane;
            uint32_t w, ci;
            if (c == tw) { w = here.w; ci = here.ci; }
            else {
                w = k0 < slen ? reinterpret_cast<const uint32_t*>(rle + off)[c * 64 + lane] : 0u;
                ci = chunkinfo[((off + (uint64_t)c * 256u) >> 8) + (uint64_t)sidx];
            }
            if (ci & kChunkDead) break;                                  // behind a terminating zero byte
            const ChunkDecode cd = decode_chunk(w, k0, slen, lane);
            chunk_toggles(a, trow, 1, cd, ci & kChunkPosMask, ci >> 31, lane);
        }
        lds_barrier();
        uint2 *dst = a.dir + (size_t)(sidx - 2 * a.dir_row0) * (size_t)nwp;
        directory_trips_tog<2>(trow, dst, tw, WPP, ntrip, nw, tail_mask, cyl, lane);      // (non-temporal stores here: 13.8 instead of 6.6 ms per 26 GB)
        if (tw == 0 && lane == 0) {
            for (int i = nw; i < nwp; ++i) dst[i] = make_uint2(0u, 0u);  // the sentinel entry padding slots read
            a.dir_n0[sidx - 2 * a.dir_row0] = (uint32_t)m - tot1;
        }
    }
}

// ... and for narrow plane-rows (a string of one or two 256-byte chunks, a handful of directory trips) every WAVE builds plane-rows of
// its own: no barrier, four plane-rows in flight per workgroup instead of one whose toggles three of the four waves wait for.
// HRC shape x 142,000 sites: producer 1.30 -> 1.22 ms (3.5 -> 3.8 TB/s of stores), the scan 5.10 -> 4.97; x 524,288: 18.25 -> 18.12.
template <int NT>
__global__ __launch_bounds__(NT) void dirbuild_solo_kernel(const ScanArgs a, const uint64_t *__restrict__ rowdesc,
                                                           const uint8_t *__restrict__ rle, const uint32_t *__restrict__ chunkinfo,
                                                           const uint32_t *__restrict__ segc, int64_t str_lo, int64_t str_hi, int per_wg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WPP = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int tw = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = a.m, nw = a.nw, nwp = a.dir_nwp;
    const int nwt = (nw + 4) & ~3;
    uint32_t *trow = reinterpret_cast<uint32_t*>(smem) + (size_t)tw * nwt;   // this wave's toggle words
    for (int i = lane; i < nwt; i += 64) trow[i] = 0u;
    const uint32_t tail_mask = (m & 31) ? ((1u << (m & 31)) - 1u) : 0xffffffffu;
    const int ntrip = (nw + 255) >> 8;                                   // (<= 63: checked at launch)
    const int64_t s0 = str_lo + (int64_t)blockIdx.x * per_wg;
    int64_t s1 = s0 + per_wg;
    if (s1 > str_hi) s1 = str_hi;
    struct Ahead { uint32_t w, ci, cyl; };
    auto load_ahead = [&](int64_t sidx, uint64_t d) -> Ahead {
        Ahead p = {0u, 0u, 0u};
        if (sidx >= s1) return p;
        const uint32_t slen = (uint32_t)(d >> kDescLenShift);
        const uint64_t off = d & kDescOffMask;
        const uint32_t *sc = segc + (size_t)sidx * (size_t)(a.S8 + 1);
        if (lane < ntrip || lane == 63) p.cyl = sc[lane == 63 ? a.S8 : lane];
        if (4u * (uint32_t)lane < slen) p.w = reinterpret_cast<const uint32_t*>(rle + off)[lane];
        if (slen) p.ci = chunkinfo[(off >> 8) + (uint64_t)sidx];
        return p;
    };
    int64_t sidx = s0 + tw;
    uint64_t desc = sidx < s1 ? rowdesc[sidx] : 0ull;
    uint64_t desc1 = sidx + WPP < s1 ? rowdesc[sidx + WPP] : 0ull;
    Ahead cur = load_ahead(sidx, desc);
    for (; sidx < s1; sidx += WPP) {
        const uint64_t cd0 = desc;
        const Ahead here = cur;
        desc = desc1;
        cur = load_ahead(sidx + WPP, desc);
        desc1 = sidx + 2 * WPP < s1 ? rowdesc[sidx + 2 * WPP] : 0ull;
        const uint32_t slen = (uint32_t)(cd0 >> kDescLenShift);
        const uint64_t off = cd0 & kDescOffMask;
        const uint32_t cyl = lane == 63 ? 0u : here.cyl;
        const uint32_t tot1 = (uint32_t)__builtin_amdgcn_readlane((int)here.cyl, 63);
        for (int c = 0; (uint32_t)c * 256u < slen; ++c) {
            const uint32_t k0 = (uint32_t)c * 256u + 4u * (uint32_t)lane;
            uint32_t w, ci;
            if (c == 0) { w = here.w; ci = here.ci; }
            else {
                w = k0 < slen ? reinterpret_cast<const uint32_t*>(rle + off)[c * 64 + lane] : 0u;
                ci = chunkinfo[((off + (uint64_t)c * 256u) >> 8) + (uint64_t)sidx];
            }
            if (ci & kChunkDead) break;
            const ChunkDecode cd = decode_chunk(w, k0, slen, lane);
            chunk_toggles(a, trow, 1, cd, ci & kChunkPosMask, ci >> 31, lane);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the wave's own toggles (LDS atomics) have landed
        uint2 *dst = a.dir + (size_t)(sidx - 2 * a.dir_row0) * (size_t)nwp;
        directory_trips_tog<2>(trow, dst, 0, 1, ntrip, nw, tail_mask, cyl, lane);
        if (lane == 0) {
            for (int i = nw; i < nwp; ++i) dst[i] = make_uint2(0u, 0u);
            a.dir_n0[sidx - 2 * a.dir_row0] = (uint32_t)m - tot1;
        }
    }
}

// ... and for plane-rows whose toggle words do not fit the LDS (more than 650,000 haplotypes: 2 x m / 8 bytes): the same team, the
// same two alternating toggle arrays, in MEMORY -- a region per workgroup of a.tog_mem.  The toggles are atomic XORs performed in
// the L2, which the CU's vector cache does not see: the barrier between a plane-row's toggles and its directory trips therefore
// also waits for memory and invalidates that cache (__threadfence: release + acquire at device scope), and so does the one that
// orders the trips' clearing stores before the next toggles.  chunk_toggles / directory_trips_tog are the LDS kernels' own.
// The reference opens any int32 m (pbwt.c:92-105, 221-262); speed is not this path's point.
__device__ __forceinline__ void mem_barrier() { __threadfence(); __syncthreads(); __threadfence(); }

template <int NT>
__global__ __launch_bounds__(NT) void dirbuild_mem_kernel(const ScanArgs a, const uint64_t *__restrict__ rowdesc,
                                                          const uint8_t *__restrict__ rle, const uint32_t *__restrict__ chunkinfo,
                                                          const uint32_t *__restrict__ segc, int64_t str_lo, int64_t str_hi, int per_wg)
{
    constexpr int WPP = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int tw = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = a.m, nw = a.nw, nwp = a.dir_nwp;
    const int nwt = (nw + 4) & ~3;
    uint32_t *TOG = a.tog_mem + (size_t)blockIdx.x * 2 * (size_t)nwt;    // [2][nwt], this workgroup's
    for (int i = tid; i < 2 * nwt; i += NT) TOG[i] = 0u;
    mem_barrier();
    const uint32_t tail_mask = (m & 31) ? ((1u << (m & 31)) - 1u) : 0xffffffffu;
    const int ntrip = (nw + 255) >> 8;                                   // (<= 63 WPP trips: checked at launch)
    const int64_t s0 = str_lo + (int64_t)blockIdx.x * per_wg;
    int64_t s1 = s0 + per_wg;
    if (s1 > str_hi) s1 = str_hi;
    for (int64_t sidx = s0; sidx < s1; ++sidx) {
        uint32_t *trow = TOG + (size_t)((sidx - s0) & 1) * nwt;
        const uint64_t cd0 = rowdesc[sidx];
        const uint32_t slen = (uint32_t)(cd0 >> kDescLenShift);
        const uint64_t off = cd0 & kDescOffMask;
        const uint32_t *sc = segc + (size_t)sidx * (size_t)(a.S8 + 1);
        const int t = tw + lane * WPP;                                   // lane l of wave tw carries trip tw + l WPP; lane 63: the row's ones
        uint32_t cyl = (t < ntrip && lane < 63) ? sc[t] : 0u;
        const uint32_t tot1 = sc[a.S8];
        for (int c = tw; (uint32_t)c * 256u < slen; c += WPP) {
            const uint32_t k0 = (uint32_t)c * 256u + 4u * (uint32_t)lane;
            const uint32_t w = k0 < slen ? reinterpret_cast<const uint32_t*>(rle + off)[c * 64 + lane] : 0u;
            const uint32_t ci = chunkinfo[((off + (uint64_t)c * 256u) >> 8) + (uint64_t)sidx];
            if (ci & kChunkDead) break;                                  // behind a terminating zero byte
            const ChunkDecode cd = decode_chunk(w, k0, slen, lane);
            chunk_toggles(a, trow, 1, cd, ci & kChunkPosMask, ci >> 31, lane);
        }
        mem_barrier();                                                   // this plane-row's toggles are all in memory, and visible
        uint2 *dst = a.dir + (size_t)(sidx - 2 * a.dir_row0) * (size_t)nwp;
        directory_trips_tog<1>(trow, dst, tw, WPP, ntrip, nw, tail_mask, cyl, lane);
        if (tw == 0 && lane == 0) {
            for (int i = nw; i < nwp; ++i) dst[i] = make_uint2(0u, 0u);  // the sentinel entry padding slots read
            a.dir_n0[sidx - 2 * a.dir_row0] = (uint32_t)m - tot1;
        }
        // (the trips clear the words they read; the array is next written two plane-rows on, behind the barrier above)
    }
}

int64_t dirbuild_mem_workgroups(int64_t n_rows) { const int64_t n_str = 2 * n_rows; return n_str < 1024 ? (n_str > 0 ? n_str : 1) : 1024; }
int64_t dirbuild_mem_words(int m) { return ((int64_t)((m + 31) / 32) + 4) & ~(int64_t)3; }

hipError_t launch_dirbuild_mem(const ScanArgs &a, int64_t row_lo, int64_t row_hi, hipStream_t s)
{
    if (row_hi <= row_lo) return hipSuccess;
    constexpr int NT = 1024;                                             // 16 waves: up to 63 x 16 directory trips = 8.2 M positions
    if (((a.nw + 255) >> 8) > 63 * (NT / 64) || !a.tog_mem) return hipErrorInvalidConfiguration;
    const int64_t n_str = 2 * (row_hi - row_lo);
    const int64_t grid = dirbuild_mem_workgroups(row_hi - row_lo);
    const int per_wg = (int)((n_str + grid - 1) / grid);
    hipLaunchKernelGGL(dirbuild_mem_kernel<NT>, dim3((unsigned)((n_str + per_wg - 1) / per_wg)), dim3(NT), 0, s, a, a.rowdesc, a.rle, a.chunkinfo, a.segc,
                       2 * row_lo, 2 * row_hi, per_wg);
    return hipGetLastError();
}

hipError_t launch_dirbuild(const ScanArgs &a, int64_t row_lo, int64_t row_hi, hipStream_t s)
{
    if (row_hi <= row_lo) return hipSuccess;
    const int nwt = (a.nw + 4) & ~3;
    const int lds = 2 * nwt * 4;
    if (lds > kLdsBytesDir) return hipErrorInvalidConfiguration;
    const int64_t n_str = 2 * (row_hi - row_lo);
    if (a.nw <= kSoloWords) {                                            // narrow plane-rows: a wave per plane-row
        auto fs = dirbuild_solo_kernel<256>;
        const int lds_solo = 4 * nwt * 4;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fs), hipFuncAttributeMaxDynamicSharedMemorySize, lds_solo);
        if (e != hipSuccess) return e;
        int per_wg = 32;
#ifdef BGTH_ABLATE
        if (const char *e = getenv("BGTH_SOLO_PER_WG")) per_wg = atoi(e);   // (profiling build: tuning knob)
#endif
        while (per_wg > 4 && (n_str + per_wg - 1) / per_wg < 3072) per_wg >>= 1;
        const int64_t grid = (n_str + per_wg - 1) / per_wg;
        hipLaunchKernelGGL(fs, dim3((unsigned)grid), dim3(256), lds_solo, s, a, a.rowdesc, a.rle, a.chunkinfo, a.segc,
                           2 * row_lo, 2 * row_hi, per_wg);
        return hipGetLastError();
    }
    auto fn = dirbuild_kernel<256>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    // plane-rows per workgroup: enough to amortise the start of a workgroup, few enough to spread short ranges over the chip
    int per_wg = 16;
    while (per_wg > 2 && (n_str + per_wg - 1) / per_wg < 3072) per_wg >>= 1;
    const int64_t grid = (n_str + per_wg - 1) / per_wg;
    hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(256), lds, s, a, a.rowdesc, a.rle, a.chunkinfo, a.segc,
                       2 * row_lo, 2 * row_hi, per_wg);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------------------------
// consumer
// ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// NB: plane buffers known at compile time (4 or 3; the counts-only one-group kernels, where a row's bookkeeping is a visible share
// of its time -- as run-time flags the compiler kept them as lane masks and re-tested them through VGPRs every row), 0 = a.dir_stage decides
// WC: whole cohort, one group, counts only -- n(code 3) alone is counted, the planes' ones come from the rows' zero counts (BGTH_COUNT3)
template <int NT, int CPT, bool MULTI, bool GT, bool S4, int NB = 0, bool WC = false>
__global__ __launch_bounds__(NT) void walk_kernel(const ScanArgs a, const uint32_t *__restrict__ n0tab)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NWAVE = NT / 64;
    static_assert(CPT % 2 == 0, "columns per thread are stepped in pairs");
    static_assert(CPT * 64 < 65536, "per-wave counts of a row are kept in 16 bits");
    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int S   = a.n_slices;                                          // (sub-block, slice): all slices of a sub-block on one XCD
    const int wg  = blockIdx.x;
    const int sup = wg / (8 * S), rem = wg % (8 * S);
    const int slice = rem >> 3;
    const int bl    = sup * 8 + (rem & 7);
    if (bl >= a.n_blk) return;

    const int m = a.m, nw = a.nw, nwp = a.dir_nwp, G = a.G;
    const uint32_t plane_bytes = (uint32_t)nwp * 8u;                     // multiple of 16
    // NB = 8, 16: ROWS PER BARRIER -- narrow plane-rows (<= 19 KB: m <= 78,000; <= 9 KB: m <= 38,000) leave the LDS room for two
    // sets of 2 or 4 rows, and the workgroup meets once per 2 or 4 rows instead of once per row (the wait at the barrier -- the SIMD
    // arbiter's oldest waves are through a row ~15 % ahead of its youngest -- was 13 % of a row at the HRC shape)
    constexpr int RPB = NB >= 8 ? NB / 4 : 1;
    const bool four = NB >= 4 || (NB == 0 && (a.dir_stage & 4));         // both planes of the next row land during this row's walk
    const bool staged = NB == 3 || (NB == 0 && !four && (a.dir_stage & 1)), warm = !four && (a.dir_stage & 2);
    const int nplane = NB >= 8 ? NB : four ? 4 : staged ? 3 : 2;
    int32_t *lcnt = reinterpret_cast<int32_t*>(smem + (size_t)nplane * plane_bytes);   // [2][cnt_stride]: rows alternate
    const int cnt_stride = MULTI ? G * 3 : NWAVE * 2;
    const uint32_t pad_rank = 32u * (uint32_t)nw;
    const uint32_t lds0 = __builtin_amdgcn_groupstaticsize();

    const int64_t blk      = (int64_t)a.blk0 + bl;
    const int64_t blk_beg  = blk << a.shift;
    int64_t       blk_end  = (blk + 1) << a.shift;
    if (blk_end > a.row1) blk_end = a.row1;

    const int chunk0 = (slice * NWAVE + wave) * CPT;
    uint32_t r0[CPT], r1[CPT];
    {
        const int32_t *rk = a.rank0 + blk * a.rank0_blk_stride;
        load_start_ranks<CPT>(a, blk, rk, chunk0, lane, false, pad_rank, r0, r1);
    }
    if (MULTI) for (int i = tid; i < 2 * cnt_stride; i += NT) lcnt[i] = 0;
    // MULTI: the group of every row-step statement of this wave (see scan_kernel): counts accumulate on the scalar unit and
    // leave for the LDS where the group changes; 255 = the statement straddles two groups
    constexpr int STEP_ = S4 ? 4 : 2;
    constexpr int NSTMT = (CPT + STEP_ - 1) / STEP_;
    // (four group numbers to a word, kept in SGPRs: as a VGPR per statement the table cost the 50-column walk-only kernel 25 registers
    //  it did not have -- it spilled, and a two-group scan of a C4-width cohort took 2.7 x the ungrouped time)
    uint32_t stmt_groups[MULTI ? (NSTMT + 3) / 4 : 1] = {};
    auto stmt_group = [&](int q) -> uint32_t { return (stmt_groups[q >> 2] >> (8 * (q & 3))) & 255u; };
    if constexpr (MULTI) {
        uint32_t last = 0;
#pragma unroll
        for (int q = 0; q < NSTMT; ++q) {
            uint32_t g = 254u;
#pragma unroll
            for (int u = 0; u < STEP_; ++u) {
                const int c = chunk0 + STEP_ * q + u;
                if (STEP_ * q + u < CPT && c < a.n_chunks) {
                    const uint32_t x = a.chunk_desc[c] & 255u;
                    g = g == 254u ? x : (g == x ? g : 255u);
                }
            }
            if (g == 254u) g = last;
            stmt_groups[q >> 2] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(stmt_groups[q >> 2] | g << (8 * (q & 3))));
            if (g != 255u) last = g;
        }
    }

    // LDS-DMA of one plane-row: pieces of 1 KiB (64 lanes x 16 bytes) dealt round-robin over the waves.  A wave's lane offset, its
    // number of pieces and the lanes of its last piece are constants of the kernel; the source is a running pointer (the next row's
    // plane 0, advanced by two plane-rows per row) -- as `dirbase + (2 (row - dir_row0) + plane) * plane_bytes` per call, with a
    // 64-bit multiply, a loop and a compare per piece, the two or three calls of a row were ~60 scalar instructions per wave: 16
    // waves x 60 on the CU's one scalar unit = the ~1.1 k cycles every row spent ISSUING its DMA (profiles/r05_walk).
    const unsigned char *dirbase = reinterpret_cast<const unsigned char*>(a.dir);
    const int npiece = (int)((plane_bytes + 1023u) >> 10);
    const uint32_t voff = (uint32_t)wave * 1024u + (uint32_t)lane * 16u;
    const int npc = __builtin_amdgcn_readfirstlane(wave < npiece ? (npiece - wave + NWAVE - 1) / NWAVE : 0);
    // (every lane of the workgroup is active where the DMA is issued: the last piece's lanes are an exec mask kept in SGPRs)
    const uint64_t last_lanes = __ballot(npc > 0 && voff + (uint32_t)(npc - 1) * (uint32_t)(NWAVE * 1024) < plane_bytes);
    const uint32_t lds_wave = lds0 + (uint32_t)wave * 1024u;
    auto dma_plane = [&](int buf, const unsigned char *src) {
        if (BGTH_SKIP(a, 0x100000) || npc == 0) return;                  // (profiling build, timing only: 0x10000 no walk, 0x80000 no barriers, 0x100000 no DMA)
        uint32_t m0v = lds_wave + (uint32_t)buf * plane_bytes;
#pragma unroll 1
        for (int k = 0; k + 1 < npc; ++k) {
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(src), "s"(m0v) : "memory");
            src += NWAVE * 1024;
            m0v += NWAVE * 1024;
        }
        asm volatile("s_mov_b64 exec, %3\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1"
                     :: "v"(voff), "s"(src), "s"(m0v), "s"(last_lanes) : "memory");
    };
    // One piece (the wave's k-th) of a plane-row, for the DMA that is dealt out over the walk's statements: a wave whose DMA
    // instructions wait for the vector-memory path at the top of the row (49 KB-sized instructions per plane and CU, 16 cycles each)
    // cannot issue its walk behind them -- in-order issue -- and the SIMDs idle; one piece between two statements is taken at once.
    auto dma_piece = [&](int buf, const unsigned char *src, int k) {
        if (BGTH_SKIP(a, 0x100000)) return;
        const uint32_t m0v = lds_wave + (uint32_t)buf * plane_bytes + (uint32_t)k * (uint32_t)(NWAVE * 1024);
        const uint64_t lanes = k + 1 == npc ? last_lanes : ~0ull;
        asm volatile("s_mov_b64 exec, %3\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1"
                     :: "v"(voff), "s"(src + (size_t)k * (NWAVE * 1024)), "s"(m0v), "s"(lanes) : "memory");
    };
    // Where a wave issues them: its k-th piece ahead of statement k (plane 1's, with four buffers, from statement MAXP on) -- or,
    // with three buffers of ~50 KB (TURNS: 3-4 pieces per wave), the waves take turns: bit q of `issue_mask` = this wave issues
    // its next piece ahead of statement q, the turns spread over the first 5/8 of the statements so that the last piece has the
    // rest of the walk to land.  Sixteen waves at the same statement queue for the one address path all the same (~360 cycles per
    // piece and wave).  One C4 shard / HRC shape x 142,000 / x 524,288 sites, ms: at the top of the row 156.5 / 5.31 / 19.15,
    // k-th piece at statement k 156.3 / 5.21 / 18.72, turns 152.2 / 5.33 / 19.08 (the test and the out-of-line branch per statement
    // cost the 16-statement kernel what the turns give it).
    constexpr bool TURNS = NB == 3;
    constexpr int NSTMT_W = (CPT + (S4 ? 4 : 2) - 1) / (S4 ? 4 : 2);
    constexpr int MAXP = (54 + NWAVE - 1) / NWAVE;                      // pieces per wave and plane: three <= 54 KB planes fit the LDS (checked at launch)
    static_assert(NSTMT_W >= 2 * MAXP && NSTMT_W <= 32, "a statement per piece");
    uint32_t issue_mask = 0;
    {
        const int total = npc;                                           // (three buffers: plane 0 of the next row only)
        const int W = total > NSTMT_W * 5 / 8 ? total : NSTMT_W * 5 / 8;
        for (int i = 0; TURNS && i < total; ++i) {
            int q = (wave * W / NWAVE + i * W / total) % W;
            while (issue_mask >> q & 1u) q = (q + 1) % W;
            issue_mask |= 1u << q;
        }
        issue_mask = (uint32_t)__builtin_amdgcn_readfirstlane((int)issue_mask);
    }
    auto row_src = [&](int64_t row) { return dirbase + (size_t)(2 * (row - a.dir_row0)) * plane_bytes; };   // plane 0 of `row`

    // (profiling build only: cycles per phase -- 0 stage DMA issue, 1 walk, 2 wait + barrier, 3 counts, 4 plane-1 DMA issue,
    //  5 its wait + barrier)
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = BGTH_TIMES(a) ? __builtin_amdgcn_s_memtime() : 0ull;
    int c0 = 0, c1 = 1, st = 2;                                          // plane buffers: current row's planes, staging
    for (int r = 0; r < RPB; ++r)                                        // (buffer 2 r + p: plane p of the batch's r-th row)
        if (blk_beg + r < blk_end) { dma_plane(2 * r, row_src(blk_beg + r)); dma_plane(2 * r + 1, row_src(blk_beg + r) + plane_bytes); }
    wait_vm0();
    lds_barrier();
    const unsigned char *nsrc = row_src(blk_beg + RPB);                  // plane 0 of the row (batch) after the one being walked

    // (rows counted from the sub-block's first, 32 bits, and the rows' zero counts behind a running pointer: the loop's own
    //  bookkeeping stays on the scalar unit without 64-bit compares in VGPRs)
    const int nrows = __builtin_amdgcn_readfirstlane((int)(blk_end - blk_beg));
    const int emit_from = __builtin_amdgcn_readfirstlane(a.row0 > blk_beg ? (int)(a.row0 - blk_beg) : 0);
    const uint32_t *n0p = n0tab + 2 * (blk_beg - a.dir_row0);
    uint32_t z0 = nrows > 0 ? n0p[0] : 0u, z1 = nrows > 0 ? n0p[1] : 0u;   // the rows' zero counts, fetched a row ahead

    // (Measured and not kept, round 5: a row's planes walked IN TURN at three buffers -- plane 0 for all columns, a barrier that frees
    //  its buffer for the next row's plane 1, then plane 1, n(code 3) read off the column's new plane-0 rank (bit = rank >= n0) in
    //  the statements whose plane-1 ballots are not all empty.  Every fetch then has a whole row to land; one C4 shard 150.1 ->
    //  150.6 ms: the barrier between the planes costs what the exposed fetch did.  profiles/r05_walk has the per-wave times.)
    if constexpr (RPB > 1) {
        static_assert(!MULTI && !GT, "counts of one group");
        constexpr int STEP = S4 ? 4 : 2;
        int set = 0;                                                     // buffers [set * 2 RPB, (set + 1) * 2 RPB): this batch's rows
        for (int ri = 0; ri < nrows; ri += RPB, n0p += 2 * RPB) {
            const int nb = nrows - ri < RPB ? nrows - ri : RPB, nnext = nrows - ri - nb < RPB ? nrows - ri - nb : RPB;
            uint32_t zz[2 * RPB];
#pragma unroll
            for (int r = 0; r < RPB; ++r) { zz[2 * r] = r < nb ? n0p[2 * r] : 0u; zz[2 * r + 1] = r < nb ? n0p[2 * r + 1] : 0u; }
#pragma unroll
            for (int r = 0; r < RPB; ++r) {
                if (r >= nb) break;
                const uint32_t base0 = lds0 + (uint32_t)(set * 2 * RPB + 2 * r) * plane_bytes - 8u, base1 = base0 + plane_bytes;
                const uint32_t n00 = 0u - zz[2 * r], n01 = 0u - zz[2 * r + 1];
                const bool inwalk = r < nnext;                           // the next batch's r-th row lands during this row's walk
                const int nb0 = (set ^ 1) * 2 * RPB + 2 * r;
                const unsigned char *src = nsrc + (size_t)(2 * r) * plane_bytes;
                uint32_t ca = 0, cb = 0, cc = 0;
#pragma unroll
                for (int j = 0; j < CPT; j += STEP) {
                    uint64_t m0[4] = {0, 0, 0, 0}, m1[4] = {0, 0, 0, 0};
                    const int q = j / STEP;
                    // (a wave's first piece of a plane in line, further ones -- narrow plane-rows have one per wave -- out of line: the
                    //  common path falls through instead of branching around six absent pieces per row)
                    if (q < MAXP) { if (__builtin_expect(inwalk && q < npc, q == 0)) dma_piece(nb0, src, q); }
                    else if (q < 2 * MAXP) { if (__builtin_expect(inwalk && q - MAXP < npc, q == MAXP)) dma_piece(nb0 + 1, src + plane_bytes, q - MAXP); }
                    {   // a wave's priority falls as it gets through the BATCH, the steps close to its end (see the one-row loop below)
                        constexpr int NS = (CPT + STEP - 1) / STEP, TOT = RPB * NS;
                        const int g = r * NS + q;
                        if (g == 0) __builtin_amdgcn_s_setprio(3);
                        else if (g == TOT * 5 / 8) __builtin_amdgcn_s_setprio(2);
                        else if (g == TOT * 13 / 16) __builtin_amdgcn_s_setprio(1);
                        else if (g == TOT * 15 / 16) __builtin_amdgcn_s_setprio(0);
                    }
                    if ((CPT - j) >= STEP && STEP == 4) {
                        uint32_t q0[4] = {r0[j], r0[j + 1], r0[j + 2], r0[j + 3]};
                        uint32_t q1[4] = {r1[j], r1[j + 1], r1[j + 2], r1[j + 3]};
                        step4<false, WC>(q0, q1, m0, m1, ca, cb, cc, base0, base1, n00, n01);
#pragma unroll
                        for (int u = 0; u < 4; ++u) { r0[j + u] = q0[u]; r1[j + u] = q1[u]; }
                    } else {
                        step2<false, WC>(r0[j], r1[j], r0[j + 1], r1[j + 1], m0[0], m1[0], m0[1], m1[1], ca, cb, cc, base0, base1, n00, n01);
                    }
                }
                if (lane == 0) {
                    uint2 *slot = reinterpret_cast<uint2*>(lcnt + (set * RPB + r) * cnt_stride) + wave;
                    if constexpr (WC) *slot = make_uint2(cc, slice == 0 && wave < 2 ? (uint32_t)m + (wave == 0 ? n00 : n01) : 0u);
                    else *slot = make_uint2((ca - cc) | (cb - cc) << 16, cc);
                }
            }
            wait_vm0();                                                  // this wave's pieces of the next batch have landed
            lds_barrier();                                               // ... everybody's; every wave is through this batch
            nsrc += (size_t)(2 * RPB) * plane_bytes;
            if (tid < 3 * nb) {
                const int r = tid / 3, comp = tid - 3 * r;
                if (ri + r >= emit_from) {
                    const int32_t *lcb = lcnt + (set * RPB + r) * cnt_stride;
                    int32_t v = 0;
                    if constexpr (WC) {
                        int32_t c3 = 0;
#pragma unroll
                        for (int w = 0; w < NWAVE; ++w) c3 += lcb[w * 2];
                        v = comp == 2 ? c3 : lcb[comp * 2 + 1] - c3;
                    } else
#pragma unroll
                    for (int w = 0; w < NWAVE; ++w) {
                        const uint32_t x = (uint32_t)lcb[w * 2 + (comp >> 1)];
                        v += (int32_t)(comp == 0 ? x & 0xffffu : comp == 1 ? x >> 16 : x);
                    }
                    int32_t *dst = a.raw_counts + (size_t)(blk_beg + ri + r - a.row0) * 3 + comp;
                    if (a.n_slices == 1) *dst = v;
                    else if (v) atomicAdd(dst, v);
                }
            }
            set ^= 1;
        }
    } else
    for (int ri = 0; ri < nrows; ++ri, n0p += 2) {
        const int64_t row = blk_beg + ri;
        const bool more = ri + 1 < nrows;
        const uint32_t n00 = 0u - z0, n01 = 0u - z1;
        if (more) { z0 = n0p[2]; z1 = n0p[3]; }
        // the next row's plane 0 (four buffers: and its plane 1; {0, 1} and {2, 3} alternate) lands during the walk: issued
        // piece by piece between the walk's statements (MAXP pieces per wave and plane at most: the LDS holds <= 52 KB planes)
        const bool inwalk = (four || staged) && more && !BGTH_SKIP(a, 0x10000);     // (npc <= MAXP: three planes fit the LDS, checked at launch)
        if ((four || staged) && more && !inwalk) { dma_plane(four ? c0 ^ 2 : st, nsrc); if (four) dma_plane(c1 ^ 2, nsrc + plane_bytes); }
        const int nb0 = four ? c0 ^ 2 : st, nb1 = c1 ^ 2;
        const uint32_t turns = inwalk ? issue_mask : 0u;
        int kk = 0;                                                      // pieces issued so far: plane 0's, then plane 1's
        // Plane 1 of the next row can only be fetched when this row's buffer is free, i.e. behind the barrier that ends
        // the walk; one dword per 128-byte line now (8 KB per wave-instruction) brings it into this XCD's L2 meanwhile.
        uint32_t touched = 0;
        if (warm && more) {
            const uint32_t off = ((uint32_t)wave * 64u + (uint32_t)lane) * 128u;
            if (off < plane_bytes)
                touched = *reinterpret_cast<const uint32_t*>(nsrc + plane_bytes + off);
        }
        BGTH_TICK(0);
        // ---- walk the row: ranks stay in registers
        if (!(BGTH_SKIP(a, 0x10000))) {
            const uint32_t base0 = lds0 + (uint32_t)c0 * plane_bytes - 8u;
            const uint32_t base1 = lds0 + (uint32_t)c1 * plane_bytes - 8u;
            int32_t *lcb = lcnt + (ri & 1) * cnt_stride;
            const bool emit = ri >= emit_from;
            uint32_t ca = 0, cb = 0, cc = 0;
            constexpr int NKEEP = (CPT + 63) / 64;
            uint64_t keep0[NKEEP] = {}, keep1[NKEEP] = {};
            constexpr int STEP = S4 ? 4 : 2;                              // lookups in flight per statement: 8 or 4
            uint32_t pa = 0, pb = 0, pc = 0;                          // MULTI: ca / cb / cc at the start of the current run
            uint32_t run_g = MULTI ? (uint32_t)__builtin_amdgcn_readfirstlane((int)stmt_group(0)) : 0u;
#define BGTH_FLUSH_GROUP(GRP)                                                                              \
            do {                                                                                           \
                const uint32_t a_ = ca - pa, b_ = cb - pb, n3_ = cc - pc, n1_ = a_ - n3_, n2_ = b_ - n3_, g_ = (GRP);                         \
                if (emit && g_ < 254u && lane == 0) {                                                      \
                    int32_t *dst = lcb + g_ * 3;                                                           \
                    if (n1_) atomicAdd(dst + 0, (int32_t)n1_);                                             \
                    if (n2_) atomicAdd(dst + 1, (int32_t)n2_);                                             \
                    if (n3_) atomicAdd(dst + 2, (int32_t)n3_);                                             \
                }                                                                                          \
            } while (0)
#pragma unroll
            for (int j = 0; j < CPT; j += STEP) {
                uint64_t m0[4] = {0, 0, 0, 0}, m1[4] = {0, 0, 0, 0};
                const int NC = (CPT - j) >= STEP ? STEP : 2;              // CPT is even: the tail is one pair
                const uint32_t sg = MULTI ? (uint32_t)__builtin_amdgcn_readfirstlane((int)stmt_group(j / STEP)) : 0u;   // (uniform, and said so)
                if (MULTI && __builtin_expect(sg != run_g, 0)) {             // (out of line: the common path falls through)
                    BGTH_FLUSH_GROUP(run_g);
                    pa = ca; pb = cb; pc = cc;                            // (the scalar sums only ever grow: a run is a difference)
                    run_g = sg;
                }
                {
                    const int q = j / STEP;
                    if constexpr (TURNS) {                                // this wave's turn: its next piece of the next row
                        if (__builtin_expect(turns >> q & 1u, 0)) {       // (out of line: a taken branch per statement costs a row ~4 %)
                            dma_piece(nb0, nsrc, kk);
                            ++kk;
                        }
                    } else if (q < MAXP) { if (__builtin_expect(inwalk && q < npc, q == 0)) dma_piece(nb0, nsrc, q); }
                    else if (q < 2 * MAXP) { if (__builtin_expect(inwalk && four && q - MAXP < npc, q == MAXP)) dma_piece(nb1, nsrc + plane_bytes, q - MAXP); }
                }
                // The SIMD arbiter prefers its oldest wave: left alone, the four waves of a SIMD finish a row one after the
                // other and the early ones idle at the barrier while the last walks nearly alone (at 7.6 instead of 4.0 cycles
                // per instruction).  A wave's priority falls as it gets through its columns, so the laggards catch up -- in steps
                // close to the END of the walk: among waves of one priority the oldest still goes first, and only the lead it gains
                // in the last stretch is waited for at the barrier (per-wave times in profiles/r05_walk: the oldest four were through
                // a row 15 % ahead of the youngest).  Steps at 1/4, 1/2, 3/4 of the statements (until round 5) / 1/2, 3/4, 7/8 /
                // 5/8, 13/16, 15/16 / 3/4, 7/8, 31/32: one C4 shard 150.4 / 147.6 / 145.6 / 148.6 ms, HRC shape x 524,288 sites
                // 18.50 / 18.50 / 18.22 / 18.94.
                if (BGTH_DIR_PRIO(a)) {
                    constexpr int NS = (CPT + STEP - 1) / STEP;
                    if (j == 0) __builtin_amdgcn_s_setprio(3);
                    else if (j / STEP == NS * 5 / 8) __builtin_amdgcn_s_setprio(2);
                    else if (j / STEP == NS * 13 / 16) __builtin_amdgcn_s_setprio(1);
                    else if (j / STEP == NS * 15 / 16) __builtin_amdgcn_s_setprio(0);
                }
                if (NC == 4) {
                    uint32_t q0[4] = {r0[j], r0[j + 1], r0[j + 2], r0[j + 3]};
                    uint32_t q1[4] = {r1[j], r1[j + 1], r1[j + 2], r1[j + 3]};
                    step4<false, WC>(q0, q1, m0, m1, ca, cb, cc, base0, base1, n00, n01);
#pragma unroll
                    for (int u = 0; u < 4; ++u) { r0[j + u] = q0[u]; r1[j + u] = q1[u]; }
                } else {
                    step2<false, WC>(r0[j], r1[j], r0[j + 1], r1[j + 1], m0[0], m1[0], m0[1], m1[1], ca, cb, cc, base0, base1, n00, n01);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (u >= NC) break;
                    if (GT && lane == ((j + u) & 63)) { keep0[(j + u) >> 6] = m0[u]; keep1[(j + u) >> 6] = m1[u]; }
                    if (MULTI && __builtin_expect(run_g == 255u, 0)) {
                        const int c = chunk0 + j + u;
                        if (emit && lane == 0 && c < a.n_chunks) {
                            int32_t *dst = lcb + (a.chunk_desc[c] & 255u) * 3;
                            atomicAdd(dst + 0, __builtin_amdgcn_readfirstlane(__popcll(m0[u] & ~m1[u])));
                            atomicAdd(dst + 1, __builtin_amdgcn_readfirstlane(__popcll(~m0[u] & m1[u])));
                            atomicAdd(dst + 2, __builtin_amdgcn_readfirstlane(__popcll(m0[u] & m1[u])));
                        }
                    }
                }
                if (MULTI && __builtin_expect(run_g == 255u, 0)) { pa = ca; pb = cb; pc = cc; }
            }
            if (MULTI) BGTH_FLUSH_GROUP(run_g);
#undef BGTH_FLUSH_GROUP
            if (!MULTI && lane == 0) {
                if constexpr (WC)                                        // {n(code 3) of this wave, the row's ones of plane 0 / 1 (waves 0 / 1 of slice 0)}
                    reinterpret_cast<uint2*>(lcb)[wave] = make_uint2(cc, slice == 0 && wave < 2 ? (uint32_t)m + (wave == 0 ? n00 : n01) : 0u);
                else
                    reinterpret_cast<uint2*>(lcb)[wave] = make_uint2((ca - cc) | (cb - cc) << 16, cc);
            }
            if (GT && emit) {
#pragma unroll
                for (int q = 0; q < NKEEP; ++q) {
                    const int c = q * 64 + lane;
                    if (c < CPT && chunk0 + c < a.n_chunks) {
                        const size_t at = (size_t)(row - a.row0) * a.n_chunks + chunk0 + c;
                        a.h0[at] = keep0[q];
                        a.h1[at] = keep1[q];
                    }
                }
            }
        }
        BGTH_TICK(1);
        asm volatile("" :: "v"(touched));                                // (the touch is waited for here, not before the walk)
        wait_vm0();                                                      // this wave's pieces of the staged plane have landed
        if (!(BGTH_SKIP(a, 0x80000))) lds_barrier();                    // every wave is past its walk: both planes are free
        BGTH_TICK(2);
        // three buffers: the next row's plane 1 (and with two buffers its plane 0) can only start now -- ahead of the counts, so
        // that they run while it travels (round 5; before: counts, then the DMA, its latency behind both)
        if (!four && more) {
            if (staged) { const int nc0 = st; dma_plane(c1, nsrc + plane_bytes); st = c0; c0 = nc0; }
            else { dma_plane(c0, nsrc); dma_plane(c1, nsrc + plane_bytes); }
        }
        nsrc += 2 * (size_t)plane_bytes;
        BGTH_TICK(4);
        // ---- counts of this row and slice -> HBM
        {
            int32_t *lcb = lcnt + (ri & 1) * cnt_stride;
            if (ri >= emit_from) {
                if (MULTI) {
                    for (int i = tid; i < G * 3; i += NT) {
                        const int32_t v = lcb[i];
                        if (v) { atomicAdd(a.raw_counts + (size_t)(row - a.row0) * G * 3 + i, v); lcb[i] = 0; }
                    }
                } else if (tid < 3) {
                    const int comp = tid;
                    int32_t v = 0;
                    if constexpr (WC) {
                        int32_t c3 = 0;
#pragma unroll
                        for (int w = 0; w < NWAVE; ++w) c3 += lcb[w * 2];
                        v = comp == 2 ? c3 : lcb[comp * 2 + 1] - c3;
                    } else
#pragma unroll
                    for (int w = 0; w < NWAVE; ++w) {
                        const uint32_t x = (uint32_t)lcb[w * 2 + (comp >> 1)];
                        v += (int32_t)(comp == 0 ? x & 0xffffu : comp == 1 ? x >> 16 : x);
                    }
                    int32_t *dst = a.raw_counts + (size_t)(row - a.row0) * 3 + comp;
                    if (a.n_slices == 1) *dst = v;
                    else if (v) atomicAdd(dst, v);
                }
            }
        }
        BGTH_TICK(3);
        if (four) { c0 ^= 2; c1 ^= 2; }                                  // (one barrier per row: the next row is already there)
        else if (more) {
            wait_vm0();
            if (!(BGTH_SKIP(a, 0x80000))) lds_barrier();
            BGTH_TICK(5);
        }
    }
#ifdef BGTH_ABLATE
    if (BGTH_TIMES(a) && lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(a.debug_times + i, tsum[i]);
#endif

    if (a.final_rank) {
        int32_t *fin = a.final_rank + (int64_t)bl * a.final_blk_stride;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = chunk0 + j;
            if (c < a.n_chunks) {
                const int col = a.slot_col[c * 64 + lane];
                if (col >= 0) { fin[col] = (int32_t)~r0[j]; fin[m + col] = (int32_t)~r1[j]; }
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------------
// geometry and launch
// ----------------------------------------------------------------------------------------------------
#define BGTH_WALK_GEOMS(X) X(512, 64) X(512, 80) X(512, 98) X(1024, 50) X(1024, 44) X(1024, 40) X(1024, 36) X(1024, 32) X(1024, 26)

static int walk_lds_need(int nw, int G, int threads, int nplane)
{
    const int nwp = (nw + 2) & ~1;
    const int cnt = G > 1 ? G * 3 * 4 : (threads / 64) * 8;
    return nplane * nwp * 8 + 2 * cnt * (nplane >= 8 ? nplane / 4 : 1) + 64;
}

bool choose_walk_geometry(int m, int n_chunks, int G, int n_blk, int want_threads, int want_cpt, Geometry *g)
{
    const int nw = (m + 31) / 32;
    static const int geoms[][2] = {
#define X(nt, cpt) {nt, cpt},
        BGTH_WALK_GEOMS(X)
#undef X
    };
    int best = -1;
    long best_key = 0;
    for (int i = 0; i < (int)(sizeof(geoms) / sizeof(geoms[0])); ++i) {
        const int nt = geoms[i][0], cpt = geoms[i][1];
        if (want_threads && nt != want_threads) continue;
        if (want_cpt && cpt != want_cpt) continue;
        if (walk_lds_need(nw, G, nt, 2) > kLdsBytesDir) continue;
        const int cap = nt / 64 * cpt;
        const long slices = (n_chunks + cap - 1) / cap;
        const long waste = slices * cap - n_chunks;
        // Time of the launch ~ rounds of workgroups over the 256 CUs (one walk-only workgroup per CU) x the time of one: a
        // row costs a workgroup its columns' lookups plus a fixed part (stage DMA, two barriers: about 9 columns' worth at four waves per SIMD,
        // profiles/r03_c4shard).  A long scan thus wants the least slices x (columns + 9) -- idle slots of a half-empty last
        // slice are lookups like any other (m = 65,000: 2 x 32 columns instead of 2 x 50) --, a short one (fewer workgroups
        // than CUs) the fewest columns that still fit the chip in one round.  512 threads: two waves per SIMD issue the row
        // step ~15 % slower (28.1 ms with 1024 x 50 against 30.7 ms with 512 x 98 at m = 200,000).
        const long wgs = slices * (long)n_blk;
        const long rounds256 = wgs <= 256 ? 256 : wgs;                       // (in 1/256 rounds)
        const long per_simd = (long)cpt * (nt / 256);                        // lookups (x 2 planes) a SIMD issues per row
        const long key = rounds256 * (per_simd + 36) * (nt == 1024 ? 100 : 115) * 64 + waste;
        if (best < 0 || key < best_key) best = i, best_key = key;
    }
    if (best < 0) return false;
    g->threads = geoms[best][0];
    g->cpt = geoms[best][1];
    const int cap = g->threads / 64 * g->cpt;
    g->slices = (n_chunks + cap - 1) / cap;
    g->K = 1; g->wpp = g->threads / 64; g->nbuf = 1; g->tog_off = 0;
    // plane buffers in LDS: four where they fit (m <= 160,000: both planes of the next row land during the walk, one barrier
    // per row), else three (plane 0 of the next row lands during the walk, plane 1 behind it: two barriers), else two
#ifdef BGTH_ABLATE
    static const int four_knob = [] { const char *v = getenv("BGTH_WALK_FOUR"); return v ? atoi(v) : 1; }();   // (0: A/B runs)
#else
    constexpr int four_knob = 1;
#endif
    // ... and sixteen or eight (one group): two sets of 4 or 2 rows, one barrier per set
    const int nplane = four_knob && G == 1 && walk_lds_need(nw, G, g->threads, 16) <= kLdsBytesDir ? 16
                     : four_knob && G == 1 && walk_lds_need(nw, G, g->threads, 8) <= kLdsBytesDir ? 8
                     : four_knob && walk_lds_need(nw, G, g->threads, 4) <= kLdsBytesDir ? 4 : walk_lds_need(nw, G, g->threads, 3) <= kLdsBytesDir ? 3 : 2;
    g->dir_stage = nplane == 16 ? 4 | 32 : nplane == 8 ? 4 | 16 : nplane == 4 ? 4 : nplane == 3 ? 1 : 0;   // (bits 4 / 5: 2 / 4 rows per barrier)
    g->lds_bytes = (walk_lds_need(nw, G, g->threads, nplane) + 15) & ~15;
    g->workgroups = ((n_blk + 7) / 8) * 8 * g->slices;
    return true;
}

template <int NT, int CPT, bool MULTI, bool GT, int NB = 0, bool WC = false>
static hipError_t launch_walk_one(const ScanArgs &a, const Geometry &g, hipStream_t s)
{
    auto fn = walk_kernel<NT, CPT, MULTI, GT, (NT <= 512), NB, WC>;       // 1024 threads: 128 VGPRs, lookups in pairs (8 scratch registers)
    // (the next row's planes are fetched a piece per statement: at most ceil(54 / waves) pieces of 1 KiB per wave and plane)
    if ((a.dir_stage & 5) && (a.dir_nwp * 8 + 1023) / 1024 > (54 + NT / 64 - 1) / (NT / 64) * (NT / 64)) return hipErrorInvalidConfiguration;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, g.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fn, dim3(g.workgroups), dim3(NT), g.lds_bytes, s, a, a.dir_n0);
    return hipGetLastError();
}

hipError_t launch_walk(const ScanArgs &a, const Geometry &g, hipStream_t s)
{
    const int v = (a.G > 1 ? 2 : 0) | (a.h0 ? 1 : 0);
#define X(NT_, CPT_)                                                                \
    if (g.threads == NT_ && g.cpt == CPT_) {                                        \
        switch (v) {                                                                \
        case 0: if (a.whole_counts && (a.dir_stage & 5))                                                             \
                    return (a.dir_stage & 32) ? launch_walk_one<NT_, CPT_, false, false, 16, true>(a, g, s)           \
                         : (a.dir_stage & 16) ? launch_walk_one<NT_, CPT_, false, false, 8, true>(a, g, s)            \
                         : (a.dir_stage & 4) ? launch_walk_one<NT_, CPT_, false, false, 4, true>(a, g, s)             \
                                             : launch_walk_one<NT_, CPT_, false, false, 3, true>(a, g, s);            \
                return (a.dir_stage & 4) ? launch_walk_one<NT_, CPT_, false, false, 4>(a, g, s)                      \
                     : (a.dir_stage & 1) ? launch_walk_one<NT_, CPT_, false, false, 3>(a, g, s)                      \
                                         : launch_walk_one<NT_, CPT_, false, false>(a, g, s);                        \
        case 1: return launch_walk_one<NT_, CPT_, false, true>(a, g, s);            \
        case 2: return launch_walk_one<NT_, CPT_, true, false>(a, g, s);            \
        default: return launch_walk_one<NT_, CPT_, true, true>(a, g, s);            \
        }                                                                           \
    }
    BGTH_WALK_GEOMS(X)
#undef X
    return hipErrorInvalidConfiguration;
}

}  // namespace bgth
