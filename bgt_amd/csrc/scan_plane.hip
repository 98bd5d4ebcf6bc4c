// Plane-split kernels for SPARSE selections of a wide cohort (configs[2]: 5,000 of 100,000 samples).
//
// With few tracked columns the walk is a small part of a row (20,000 lookups against two 6,250-word directories to
// build at m = 200,000) and the team kernel of scan_device.inc.h spends its time in the latency chain of the build and
// at its barriers: one workgroup per CU (100 KB of directory + 50 KB of toggles), two barriers per row, nothing to
// overlap them with.  The two planes never meet in the walk (plane p's ranks move by plane p's row alone, reference
// pbwt.c:129-170), only in the counts.  So here a workgroup owns ONE plane:
//
//   plane_kernel    grid = (sub-block, plane).  512 threads track the selection's ranks of that plane (CPT chunks per
//                   wave), build one 50 KB plane-row per row from the row index (team of 8 waves: chunks -> toggles ->
//                   directory trips, as the team kernel does) and walk it; 75 KB of LDS, <= 128 VGPRs: TWO workgroups
//                   share a CU, and while one sits in its build or at a barrier the other walks.  Output: the plane's
//                   64-bit ballot of every chunk and row (the bit planes H0 / H1 the genotype path uses anyway).
//   count_kernel    counts[row][group] = popcounts of H0 & ~H1, ~H0 & H1, H0 & H1 per chunk (reference bgt.c:735-757).
//
// Same arithmetic, same row step (the plane-0 branch of the hand-scheduled statement).
#include "scan_device.inc.h"

namespace bgth {

static const int kLdsBytesPlane = 160 * 1024;

// LOW: the <= 80-VGPR statement, six waves per SIMD: THREE workgroups of 512 threads share a CU where their LDS allows
// (m <= 139,000), else TWO of 768 (twelve waves each).  Measured against two of 512 at 120 VGPRs, every 20th sample,
// 1.5 M sites: m = 100,000 16.0 -> 12.4 ms, m = 131,072 19.7 -> 16.0 ms; HRC shape, every 13th: 1.58 -> 1.15 ms.
// (Tried and dropped: a third workgroup of 512 threads at m = 140,000 ... 212,000 by keeping the toggles in the .x of the
// entries themselves -- four barriers per row, nothing of the build beside the workgroup's own walk: 55.5 against 68.0 M sites/s at C3.)
// (Eight waves per SIMD -- four workgroups of 512 or two of 1024 threads at 64 VGPRs -- gain nothing over six: m = 64,976
// 1.27 against 1.15 ms, m = 100,000 equal, C3 with 2 x 1024 16.8 against 14.7 ms with 2 x 768.)
template <int CPT, int LOW = 0, int NT = 512>                        // LOW 1: <= 80 VGPRs (six waves per SIMD)
__global__ __launch_bounds__(NT, LOW == 1 ? 6 : 4) void plane_kernel(const ScanArgs a, const uint64_t *__restrict__ rowdesc,
                                                       const uint8_t *__restrict__ rle, const uint32_t *__restrict__ chunkinfo,
                                                       const uint32_t *__restrict__ segc)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WPP = NT / 64;                                        // all waves: one team on the plane-row
    static_assert(CPT % 4 == 0, "columns are stepped four at a time");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroup -> (sub-block, plane).  Workgroup ids go round-robin over the 8 XCDs; k = the index within the XCD.  Plane 0
    // has the longer strings, so a CU should hold one workgroup of each plane, whichever way the XCD fills its 32 CUs x 2
    // slots: neighbours in k (depth first) and ids 32 apart in k (breadth first) both get different planes.  (With plain
    // alternation the same launch took 19.6 or 24.9 ms depending on what ran before it.)
    const int k = blockIdx.x >> 3;
    const int plane = __builtin_amdgcn_readfirstlane((k ^ (k >> 5)) & 1);
    const int bl = __builtin_amdgcn_readfirstlane((k >> 1) * 8 + (int)(blockIdx.x & 7));
    if (bl >= a.n_blk) return;

    const int m = a.m, nw = a.nw, nwp = (nw + 2) & ~1, nwt = (nw + 4) & ~3;
    uint2 *BD = reinterpret_cast<uint2*>(smem);                         // [nwp] {bits, ones before} + sentinel
    uint32_t *TOG = reinterpret_cast<uint32_t*>(smem + (size_t)nwp * 8); // [nwt] toggles of the NEXT row
    uint32_t *n0s = TOG + nwt;                                           // [2]: alternates by row
    const uint32_t pad_rank = 32u * (uint32_t)nw;
    const uint32_t lds0 = __builtin_amdgcn_groupstaticsize();
    const uint32_t tail_mask = (m & 31) ? ((1u << (m & 31)) - 1u) : 0xffffffffu;
    const int ntrip = (nw + 255) >> 8;

    const int64_t blk = (int64_t)a.blk0 + bl;
    const int64_t blk_beg = blk << a.shift;
    int64_t blk_end = (blk + 1) << a.shift;
    if (blk_end > a.row1) blk_end = a.row1;

    const int chunk0 = wave * CPT;
    uint32_t rk_[CPT];
    if (a.start_slots) {                                                 // the selection's own start ranks, compact: one coalesced read
        const int32_t *__restrict__ st = a.start_slots + blk * a.start_blk_stride + (int64_t)plane * (a.n_chunks * 64);
#pragma unroll
        for (int j = 0; j < CPT; ++j) rk_[j] = ~(chunk0 + j < a.n_chunks ? (uint32_t)st[(chunk0 + j) * 64 + lane] : pad_rank);
    } else {
        // (unconditional loads from clamped indices, the gathers in a second loop: see load_start_ranks)
        const int32_t *__restrict__ rk = a.rank0 + blk * a.rank0_blk_stride + (int64_t)plane * m;
        const int last = a.n_chunks * 64 - 1;
        int col_[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int slot = (chunk0 + j) * 64 + lane;
            col_[j] = a.slot_col[slot <= last ? slot : 0];
        }
        uint32_t t_[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) t_[j] = (uint32_t)rk[col_[j] >= 0 ? col_[j] : 0];
#pragma unroll
        for (int j = 0; j < CPT; ++j) rk_[j] = ~(chunk0 + j < a.n_chunks && col_[j] >= 0 ? t_[j] : pad_rank);
    }
    for (int i = tid; i < nwt; i += NT) TOG[i] = 0u;
    if (tid == 0) BD[nw] = make_uint2(0u, 0u);
    lds_barrier();

    uint64_t *hout = plane ? a.h1 : a.h0;
    const int tw = wave;

    // (Round 5, measured and dropped -- profiles/r05_c3/README.md: the toggles' chunks dealt to the waves that have no column to walk,
    // two chunks each: C3 13.9 -> 17.9 ms, a wave's SECOND chunk is not prefetched and its loads sit in the row's critical path; the
    // same with three chunks prefetched in the registers a walker keeps ranks in: 14.6 ms, and +9 % on a shape WITHOUT an idle wave --
    // the branches around the loads cost the compiler its vmcnt bookkeeping, the toggles then wait for the loads just issued; with
    // a row loop of its own per role (walker / toggle wave, same barriers) that is cured and C3 runs 13.3 -> 13.15 ms: the toggles
    // beside the walk instead of behind it buy 1 %, not the 1.7 ms the ablation prices them at -- not worth three loop bodies;
    // the walk's all-padding statements skipped by a wave-uniform branch: 13.9 -> 13.9 ms, they only ever filled empty issue slots.)
    // What the toggles of a row need from memory -- its descriptor, this wave's first chunk of the string with that chunk's
    // row-index record, the carries of this wave's directory trips and the row's number of ones -- is fetched a row AHEAD,
    // behind the walk (the descriptor two rows ahead, so that nothing waits for an address either).  Everything goes through
    // vector loads, uniform addresses included: scalar loads would count in lgkmcnt, and the walk's every statement opens with
    // s_waitcnt lgkmcnt(0).  Addresses: a wave-uniform 64-bit base computed on the scalar unit + a 32-bit lane offset (the
    // loads' saddr form) -- as 64-bit per-lane arithmetic the prefetch cost every wave ~34 VALU instructions per row, 12 % of
    // the kernel's.  Rows are counted from the sub-block's first (32 bits).
    const int nrows = __builtin_amdgcn_readfirstlane((int)(blk_end - blk_beg));
    const uint64_t str0 = (uint64_t)(2 * blk_beg) + (uint64_t)plane;     // string index of relative row 0: 64 bits, wave-uniform (SGPRs)
    auto vzero = []() { uint32_t z = 0; asm volatile("" : "+v"(z)); return z; };
    const char *desc0 = reinterpret_cast<const char*>(rowdesc + str0);   // (the uniform part of the address stays 64-bit on the scalar
    auto load_desc = [&](int i) -> uint64_t {                          //  unit; only the row's delta inside the sub-block is 32-bit)
        if (i >= nrows) return 0ull;
        const uint32_t bo = 16u * (uint32_t)i + vzero();
        return *reinterpret_cast<const uint64_t*>(desc0 + bo);
    };
    const char *sc0 = reinterpret_cast<const char*>(segc + (uint64_t)str0 * (uint64_t)(a.S8 + 1));
    const uint32_t sc_step = 2u * (uint32_t)(a.S8 + 1) * 4u;
    struct Ahead { uint32_t w, ci, cyl; };
    auto load_ahead = [&](int i, uint64_t d) -> Ahead {                // d = the row's descriptor (already here, wave-uniform)
        Ahead p = {0u, 0u, 0u};
        if (i >= nrows) return p;
        const uint64_t sidx = (uint64_t)str0 + 2u * (uint32_t)i;
        const uint32_t slen = (uint32_t)(d >> kDescLenShift);
        const uint64_t off = d & kDescOffMask;
        const char *sc = sc0 + (size_t)(uint32_t)i * sc_step;           // (uniform; the row's carries: 2 strings x (S8 + 1) words per row)
        const int t = tw + lane * WPP;                                  // (ntrip <= 40: lane 63 has no trip and carries the row's ones)
        if (t < ntrip || lane == 63) p.cyl = *reinterpret_cast<const uint32_t*>(sc + 4u * (uint32_t)(lane == 63 ? a.S8 : t));
        const uint32_t c0 = (uint32_t)tw * 256u;                         // this wave's first chunk of the string
        if (c0 + 4u * (uint32_t)lane < slen) p.w = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(rle + off + c0) + 4u * (uint32_t)lane);
        if (c0 < slen) p.ci = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(chunkinfo + (((off + c0) >> 8) + sidx)) + vzero());
        return p;
    };

    // toggles of relative row i into TOG (the array is clean: the directory pass clears what it reads)
    uint32_t keep_cyl = 0, keep_tot = 0;
    auto toggles = [&](int i, uint64_t d, const Ahead &p) {
        const uint64_t sidx = (uint64_t)str0 + 2u * (uint32_t)i;
        const uint32_t slen = (uint32_t)(d >> kDescLenShift);
        const uint64_t off = d & kDescOffMask;
        keep_cyl = lane == 63 ? 0u : p.cyl;
        keep_tot = (uint32_t)__builtin_amdgcn_readlane((int)p.cyl, 63);
        for (int c = tw; (uint32_t)c * 256u < slen; c += WPP) {
            const uint32_t k0 = (uint32_t)c * 256u + 4u * (uint32_t)lane;
            uint32_t w, ci;
            if (c == tw) { w = p.w; ci = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.ci); }
            else {
                w = k0 < slen ? reinterpret_cast<const uint32_t*>(rle + off)[c * 64 + lane] : 0u;
                ci = chunkinfo[((off + (uint64_t)c * 256u) >> 8) + sidx];
            }
            if (ci & kChunkDead) break;
            const ChunkDecode cd = decode_chunk(w, k0, slen, lane);
            chunk_toggles(a, TOG, 1, cd, ci & kChunkPosMask, ci >> 31, lane);
        }
    };
    // cur: relative row i + 1 of the loop below (fetched an iteration ago); d_next: descriptor of row i + 2
    uint64_t d_cur, d_next;
    Ahead cur;
    {
        const uint64_t d0 = load_desc(0);
        d_next = load_desc(1);
        d_cur = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(d0 >> 32)) << 32 | (uint32_t)__builtin_amdgcn_readfirstlane((int)d0);
        cur = load_ahead(0, d_cur);
    }
    const int emit_from = __builtin_amdgcn_readfirstlane(a.row0 > blk_beg ? (int)(a.row0 - blk_beg) : 0);   // first relative row whose ballots leave

    uint64_t *hnext = hout + ((int64_t)blk_beg - a.h_row0) * a.n_chunks + chunk0;   // this wave's ballots of the row to be walked next
    // (profiling build only: cycles per phase -- 0 walk, 1 toggles, 2 barrier, 3 directory, 4 barrier)
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = BGTH_TIMES(a) ? __builtin_amdgcn_s_memtime() : 0ull;
    // Loop (iteration -1 only prepares row blk_beg):   walk(row) + toggles(row+1) | directory(row+1) |     two barriers per row
    for (int i = -1; i < nrows; ++i) {
        const bool walking = i >= 0, more = i + 1 < nrows;
        // row + 2: its descriptor came an iteration ago; its data and the descriptor of row + 3 travel behind this walk
        const uint64_t d_ahead = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(d_next >> 32)) << 32 | (uint32_t)__builtin_amdgcn_readfirstlane((int)d_next);
        const Ahead ahead = load_ahead(i + 2, d_ahead);
        d_next = load_desc(i + 3);
        if (walking && !(BGTH_SKIP(a, 0x10000))) {                       // (profiling build: 0x10000 no walk, 0x20000 no toggles,
            const uint32_t base = lds0 - 8u;                             //  0x40000 no directory trips, 0x80000 no barriers -- timing only)
            const uint32_t n0 = 0u - n0s[i & 1];
            // chunks of this wave whose ballots leave in this row, as ONE scalar integer: as a boolean per column the test was kept
            // in 64-bit lane masks, spilled to VGPR lanes and read back with two v_readlane per column and row
            const int n_emit = __builtin_amdgcn_readfirstlane(i >= emit_from ? a.n_chunks - chunk0 : 0);
            [[maybe_unused]] uint32_t ca = 0, cb = 0, cc = 0;
            // The ballots leave straight from their SGPRs: one scalar store per column (s_store_dwordx2, written back by the
            // s_dcache_wb at the end of the kernel) instead of moving them into lanes first (two v_cndmask per column: a
            // quarter of this walk's VALU instructions).
            // A scalar store reads its data registers when it EXECUTES, not when it issues: the ballots of a group stay in
            // their SGPRs (pm) until the next statement's opening s_waitcnt lgkmcnt(0) has retired the stores.
            uint64_t *hrow = hnext;                                      // (a running pointer: no 64-bit multiply per row on the scalar unit)
            hnext += a.n_chunks;
            uint64_t pm[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < CPT; j += 4) {
                // a wave's priority falls as it gets through its columns (2 -> 1 -> 0 at 40 % and 80 %), so that the waves of a
                // SIMD reach the barrier together instead of one after the other -- below the build's 3 throughout
                // (thresholds 4/12, 8/16, 4/8 of 20 columns measured: 16.75 / 16.59 / 16.92 ms against 17.36 without)
                if (BGTH_WALK_PRIO(a)) {
                    constexpr int T1 = (2 * CPT / 5 / 4) * 4, T2 = (4 * CPT / 5 / 4) * 4;
                    if (j == 0) __builtin_amdgcn_s_setprio(2);
                    else if (T1 > 0 && j == T1) __builtin_amdgcn_s_setprio(1);
                    else if (T2 > T1 && j == T2) __builtin_amdgcn_s_setprio(0);
                }
                uint32_t q0[4] = {rk_[j], rk_[j + 1], rk_[j + 2], rk_[j + 3]};
                uint32_t q1[4] = {0u, 0u, 0u, 0u};                       // (plane-0 branch of the statement: untouched)
                uint64_t m0[4] = {0, 0, 0, 0}, m1[4] = {0, 0, 0, 0};
                if constexpr (LOW == 1) step4_plane_low(q0, m0, base, n0);
                else step4<true>(q0, q1, m0, m1, ca, cb, cc, base, 0u, n0, 0u);
                asm volatile("" :: "s"(pm[0]), "s"(pm[1]), "s"(pm[2]), "s"(pm[3]));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    rk_[j + u] = q0[u];
                    if (j + u < n_emit)                                    // wave-uniform
                        asm volatile("s_store_dwordx2 %0, %1, %2" :: "s"(m0[u]), "s"(hrow), "n"((j + u) * 8) : "memory");
                    pm[u] = m0[u];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" :: "s"(pm[0]), "s"(pm[1]), "s"(pm[2]), "s"(pm[3]) : "memory");
        }
        BGTH_TICK(0);
        // The build of the next row is a latency chain of few instructions (decode, LDS atomics, prefix scans, two barriers):
        // it goes ahead of the walk of the CU's other workgroup, which fills the issue slots it leaves.
        if (BGTH_WALK_PRIO(a)) __builtin_amdgcn_s_setprio(3);
        if (more && !(BGTH_SKIP(a, 0x20000))) toggles(i + 1, d_cur, cur);
        d_cur = d_ahead; cur = ahead;
        BGTH_TICK(1);
        if (!(BGTH_SKIP(a, 0x80000))) lds_barrier();                    // every wave is past its walk; the toggles are complete
        BGTH_TICK(2);
        if (more && !(BGTH_SKIP(a, 0x40000))) {
            directory_trips_tog<2>(TOG, BD, tw, WPP, ntrip, nw, tail_mask, keep_cyl, lane);
            if (tid == 0) n0s[(i + 1) & 1] = (uint32_t)m - keep_tot;
        }
        BGTH_TICK(3);
        if (!(BGTH_SKIP(a, 0x80000))) lds_barrier();
        if (BGTH_WALK_PRIO(a)) __builtin_amdgcn_s_setprio(0);
        BGTH_TICK(4);
    }
    asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");     // the scalar stores of the ballots reach memory
#ifdef BGTH_ABLATE
    if (BGTH_TIMES(a) && lane == 0) for (int i = 0; i < 8; ++i) atomicAdd(a.debug_times + i, tsum[i]);
#endif
}

// counts[row][g][3] += {n(code 1), n(code 2), n(code 3)} from the bit planes, one wave per row
__global__ __launch_bounds__(256) void count_planes_kernel(const uint64_t *__restrict__ h0, const uint64_t *__restrict__ h1,
                                                           const uint32_t *__restrict__ chunk_desc, int32_t *raw, int64_t n_rows,
                                                           int n_chunks, int G)
{
    __shared__ int32_t acc[4][32 * 3];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + w;
    for (int i = lane; i < G * 3; i += 64) acc[w][i] = 0;
    __syncthreads();
    if (row < n_rows) {
        for (int c = lane; c < n_chunks; c += 64) {
            const uint64_t x = h0[row * n_chunks + c], y = h1[row * n_chunks + c];
            const int g = (int)(chunk_desc[c] & 255u);
            const int c1 = __popcll(x & ~y), c2 = __popcll(~x & y), c3 = __popcll(x & y);
            if (c1) atomicAdd(&acc[w][3 * g], c1);
            if (c2) atomicAdd(&acc[w][3 * g + 1], c2);
            if (c3) atomicAdd(&acc[w][3 * g + 2], c3);
        }
    }
    __syncthreads();
    if (row < n_rows) for (int i = lane; i < G * 3; i += 64) raw[row * G * 3 + i] = acc[w][i];
}

#define BGTH_PLANE_CPTS(X) X(4) X(8) X(12) X(20) X(32)
#define BGTH_PLANE_CPTS_768(X) X(4) X(8) X(12) X(16) X(20)

// g->low: 1 = the one-plane statement and six waves per SIMD asked of the compiler (80 VGPRs); 0 = the round-3 shape (two
// workgroups of 512 threads, the two-plane statement's plane-0 branch; BGTH_PLANE_LOW=0, for A/B runs).
#ifdef BGTH_ABLATE
static int plane_knob(const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; }
#else
static int plane_knob(const char *, int dflt) { return dflt; }       // (the shipped library reads no tuning knob from the environment)
#endif
static int plane_lds(int m) { const int nw = (m + 31) / 32, nwp = (nw + 2) & ~1, nwt = (nw + 4) & ~3; return ((nwp * 8 + nwt * 4 + 16) + 15) & ~15; }
// workgroups of the plane-split kernels a CU holds: three of 512 threads where their LDS allows (m <= 139,000), else two
int plane_slots_per_cu(int m)
{
    static const int low_knob = plane_knob("BGTH_PLANE_LOW", 1);
    return low_knob && 3 * plane_lds(m) <= kLdsBytesPlane ? 3 : 2;
}

bool choose_plane_geometry(int m, int n_chunks, int n_blk, Geometry *g)
{
    const int lds = plane_lds(m), nw = (m + 31) / 32, nwp = (nw + 2) & ~1;
    if (2 * lds > kLdsBytesPlane) return false;                         // two workgroups must share a CU: that is the point
    static const int low_knob = plane_knob("BGTH_PLANE_LOW", 1);
    const int slots = plane_slots_per_cu(m);
    // six waves per SIMD: three workgroups of 512 threads or two of 768 (chunks permitting: <= 20 per wave); else two of 512
    int threads = 512, low = 0, cpt = 0;
    if (low_knob && slots == 3) { threads = 512; low = 1; }
    // (round 5, measured and dropped: two workgroups of 896 threads at 72 VGPRs = seven waves per SIMD, 12 columns per wave, no wave
    // with two directory units: C3 19.6 ms against 13.9 -- profiles/r05_c3)
    else if (low_knob && 12 * 20 >= n_chunks) { threads = 768; low = 1; }
#ifdef BGTH_ABLATE      // (profiling build, round 6: ten waves instead of twelve where they hold the selection -- C3: 160 chunk slots for 157 chunks)
    if (threads == 768 && plane_knob("BGTH_PLANE_THREADS", 768) == 640 && 10 * 20 >= n_chunks && n_chunks > 10 * 12) threads = 640;
#endif
    const int waves = threads / 64;
    if (threads == 768 || threads == 640) {
#define X(C) if (!cpt && waves * C >= n_chunks) cpt = C;
        BGTH_PLANE_CPTS_768(X)
#undef X
    } else {
#define X(C) if (!cpt && 8 * C >= n_chunks) cpt = C;
        BGTH_PLANE_CPTS(X)
#undef X
    }
    if (!cpt) return false;
    g->threads = threads; g->cpt = cpt; g->slices = 1; g->K = 1; g->wpp = threads / 64; g->nbuf = 1; g->tog_off = nwp * 8;
    g->low = low;
    g->lds_bytes = lds;
    g->workgroups = ((n_blk + 7) / 8) * 16;                              // 8 sub-blocks x 2 planes per group of 16 ids
    g->dir_stage = -1;
    return true;
}

hipError_t launch_plane_scan(const ScanArgs &a, const Geometry &g, hipStream_t s)
{
    const unsigned grid = (unsigned)g.workgroups;
#define LAUNCH(FN)                                                                                                  \
    {                                                                                                               \
        auto fn = FN;                                                                                               \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, g.lds_bytes); \
        if (e != hipSuccess) return e;                                                                              \
        hipLaunchKernelGGL(fn, dim3(grid), dim3(g.threads), g.lds_bytes, s, a, a.rowdesc, a.rle, a.chunkinfo, a.segc); \
        return hipGetLastError();                                                                                   \
    }
#define X(C) if (g.cpt == C && g.threads == 512 && g.low == 1) LAUNCH((plane_kernel<C, 1, 512>))
    BGTH_PLANE_CPTS(X)
#undef X
#define X(C) if (g.cpt == C && g.threads == 768 && g.low == 1) LAUNCH((plane_kernel<C, 1, 768>))
    BGTH_PLANE_CPTS_768(X)
#undef X
#ifdef BGTH_ABLATE
    if (g.cpt == 16 && g.threads == 640 && g.low == 1) LAUNCH((plane_kernel<16, 1, 640>))
    if (g.cpt == 20 && g.threads == 640 && g.low == 1) LAUNCH((plane_kernel<20, 1, 640>))
#endif
#define X(C) if (g.cpt == C && g.threads == 512 && g.low == 0) LAUNCH((plane_kernel<C, 0, 512>))
    BGTH_PLANE_CPTS(X)
#undef X
#undef LAUNCH
    return hipErrorInvalidConfiguration;
}

hipError_t launch_count_planes(const uint64_t *h0, const uint64_t *h1, const uint32_t *chunk_desc, int32_t *raw, int64_t n_rows,
                               int n_chunks, int G, hipStream_t s)
{
    if (n_rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(count_planes_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, s, h0, h1, chunk_desc, raw, n_rows, n_chunks, G);
    return hipGetLastError();
}


// ----------------------------------------------------------------------------------------------------
// Cohorts too wide for both bit-vectors of a row in one LDS (327,000 < m <= 650,000 haplotypes): the walk-only kernel of
// the directory path (scan_dir.hip), one PLANE per workgroup.  A plane-row (m / 4 bytes) comes from the arena by LDS-DMA
// into one of two buffers when two fit (the next row lands while this one is walked: one barrier per row), else into
// the only one; the ballots go out as bit planes and count_planes_kernel joins them.
// ----------------------------------------------------------------------------------------------------
template <int NT, int CPT>
__global__ __launch_bounds__(NT) void walk_plane_kernel(const ScanArgs a, const uint32_t *__restrict__ n0tab)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NWAVE = NT / 64;
    static_assert(CPT % 4 == 0 && CPT <= 64, "four columns per statement; lane l keeps the ballot of column l");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = a.n_slices;
    const int wg = blockIdx.x;
    const int sup = wg / (16 * S), rem = wg % (16 * S);
    const int plane = (rem >> 3) & 1, slice = rem >> 4;
    const int bl = sup * 8 + (rem & 7);
    if (bl >= a.n_blk) return;

    const int m = a.m, nw = a.nw, nwp = a.dir_nwp;
    const uint32_t plane_bytes = (uint32_t)nwp * 8u;
    const bool two = a.dir_stage & 1;
    const uint32_t pad_rank = 32u * (uint32_t)nw;
    const uint32_t lds0 = __builtin_amdgcn_groupstaticsize();
    const int64_t blk = (int64_t)a.blk0 + bl;
    const int64_t blk_beg = blk << a.shift;
    int64_t blk_end = (blk + 1) << a.shift;
    if (blk_end > a.row1) blk_end = a.row1;

    const int chunk0 = (slice * NWAVE + wave) * CPT;
    uint32_t rk_[CPT];
    {
        // (unconditional loads from clamped indices, the gathers in a second loop: see load_start_ranks)
        const int32_t *__restrict__ rk = a.rank0 + blk * a.rank0_blk_stride + (int64_t)plane * m;
        const int last = a.n_chunks * 64 - 1;
        int col_[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int slot = (chunk0 + j) * 64 + lane;
            col_[j] = a.slot_col[slot <= last ? slot : 0];
        }
        uint32_t t_[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) t_[j] = (uint32_t)rk[col_[j] >= 0 ? col_[j] : 0];
#pragma unroll
        for (int j = 0; j < CPT; ++j) rk_[j] = ~(chunk0 + j < a.n_chunks && col_[j] >= 0 ? t_[j] : pad_rank);
    }
    const unsigned char *dirbase = reinterpret_cast<const unsigned char*>(a.dir);
    const int npiece = (int)((plane_bytes + 1023u) >> 10);
    auto dma_row = [&](int buf, int64_t row) {
        const unsigned char *src = dirbase + (size_t)(2 * (row - a.dir_row0) + plane) * plane_bytes;
        for (int pc = wave; pc < npiece; pc += NWAVE) {
            const uint32_t off = (uint32_t)pc * 1024u + (uint32_t)lane * 16u;
            if (off < plane_bytes)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off),
                                                 (__attribute__((address_space(3))) void*)(smem + (size_t)buf * plane_bytes + (size_t)pc * 1024u),
                                                 16, 0, 0);
        }
    };
    uint64_t *hout = plane ? a.h1 : a.h0;
    int cur = 0;
    if (blk_beg < blk_end) dma_row(0, blk_beg);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    for (int64_t row = blk_beg; row < blk_end; ++row) {
        const bool more = row + 1 < blk_end;
        if (two && more) dma_row(cur ^ 1, row + 1);
        {
            const uint32_t base = lds0 + (uint32_t)cur * plane_bytes - 8u;
            const uint32_t n0 = 0u - n0tab[2 * (row - a.dir_row0) + plane];
            uint32_t ca = 0, cb = 0, cc = 0;
            uint64_t keep = 0;
#pragma unroll
            for (int j = 0; j < CPT; j += 4) {
                uint32_t q0[4] = {rk_[j], rk_[j + 1], rk_[j + 2], rk_[j + 3]};
                uint32_t q1[4] = {0u, 0u, 0u, 0u};
                uint64_t m0[4] = {0, 0, 0, 0}, m1[4] = {0, 0, 0, 0};
                step4<true>(q0, q1, m0, m1, ca, cb, cc, base, 0u, n0, 0u);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    rk_[j + u] = q0[u];
                    if (lane == j + u) keep = m0[u];
                }
            }
            if (row >= a.row0 && hout && lane < CPT && chunk0 + lane < a.n_chunks)
                hout[(size_t)(row - a.h_row0) * a.n_chunks + chunk0 + lane] = keep;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        if (two) cur ^= 1;
        else if (more) {
            dma_row(0, row + 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
        }
    }
    if (a.final_rank) {
        int32_t *fin = a.final_rank + (int64_t)bl * a.final_blk_stride + (int64_t)plane * m;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = chunk0 + j;
            if (c < a.n_chunks) {
                const int col = a.slot_col[c * 64 + lane];
                if (col >= 0) fin[col] = (int32_t)~rk_[j];
            }
        }
    }
}

#define BGTH_WALK_PLANE_GEOMS(X) X(1024, 16) X(1024, 32) X(1024, 64)

bool choose_walk_plane_geometry(int m, int n_chunks, int n_blk, Geometry *g)
{
    const int nw = (m + 31) / 32, nwp = (nw + 2) & ~1;
    if (nwp * 8 + 64 > kLdsBytesPlane || 2 * ((nw + 4) & ~3) * 4 > kLdsBytesPlane) return false;   // one plane-row / the producer's toggles
    int best = -1; long best_key = 0;
    static const int geoms[][2] = {
#define X(nt, cpt) {nt, cpt},
        BGTH_WALK_PLANE_GEOMS(X)
#undef X
    };
    for (int i = 0; i < (int)(sizeof(geoms) / sizeof(geoms[0])); ++i) {
        const int cap = geoms[i][0] / 64 * geoms[i][1];
        const long slices = (n_chunks + cap - 1) / cap, waste = slices * cap - n_chunks;
        const long key = slices * 1000000 + waste;
        if (best < 0 || key < best_key) best = i, best_key = key;
    }
    g->threads = geoms[best][0]; g->cpt = geoms[best][1];
    const int cap = g->threads / 64 * g->cpt;
    g->slices = (n_chunks + cap - 1) / cap;
    g->K = 1; g->wpp = g->threads / 64; g->nbuf = 1; g->tog_off = 0;
    g->dir_stage = 2 * nwp * 8 + 64 <= kLdsBytesPlane ? 1 : 0;          // two plane buffers when they fit
    g->lds_bytes = ((g->dir_stage ? 2 : 1) * nwp * 8 + 64 + 15) & ~15;
    g->workgroups = ((n_blk + 7) / 8) * 16 * g->slices;
    return true;
}

hipError_t launch_walk_plane(const ScanArgs &a, const Geometry &g, hipStream_t s)
{
#define X(NT_, C)                                                                                                   \
    if (g.threads == NT_ && g.cpt == C) {                                                                           \
        auto fn = walk_plane_kernel<NT_, C>;                                                                        \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, g.lds_bytes); \
        if (e != hipSuccess) return e;                                                                              \
        hipLaunchKernelGGL(fn, dim3(g.workgroups), dim3(NT_), g.lds_bytes, s, a, a.dir_n0);                          \
        return hipGetLastError();                                                                                   \
    }
    BGTH_WALK_PLANE_GEOMS(X)
#undef X
    return hipErrorInvalidConfiguration;
}

// ----------------------------------------------------------------------------------------------------
// Cohorts of more than 650,000 haplotypes: not even one plane-row fits the LDS.  The same workgroup = (sub-block, column slice,
// plane), the same ranks in registers, but every lookup reads its {bits, ones before} entry from the arena IN MEMORY (a plane-row
// of a million haplotypes is 250 KB: the L2 holds it).  No LDS, no barrier: the waves of a workgroup never meet.  Plain C++ --
// the reference decodes any int32 m (pbwt.c:92-105), this path is what makes the reader do so too; its speed is the L2's gather
// rate, not the row step's issue rate.
// ----------------------------------------------------------------------------------------------------
template <int NT, int CPT>
__global__ __launch_bounds__(NT) void walk_mem_kernel(const ScanArgs a, const uint32_t *__restrict__ n0tab)
{
    constexpr int NWAVE = NT / 64;
    static_assert(CPT <= 64, "lane l keeps the ballot of column l");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = a.n_slices;
    const int wg = blockIdx.x;
    const int sup = wg / (16 * S), rem = wg % (16 * S);
    const int plane = (rem >> 3) & 1, slice = rem >> 4;
    const int bl = sup * 8 + (rem & 7);
    if (bl >= a.n_blk) return;
    const int m = a.m, nw = a.nw, nwp = a.dir_nwp;
    const uint32_t pad_rank = 32u * (uint32_t)nw;                        // entry nw is the all-zero sentinel: a padding slot never moves
    const int64_t blk = (int64_t)a.blk0 + bl;
    const int64_t blk_beg = blk << a.shift;
    int64_t blk_end = (blk + 1) << a.shift;
    if (blk_end > a.row1) blk_end = a.row1;
    const int chunk0 = (slice * NWAVE + wave) * CPT;
    if (chunk0 >= a.n_chunks) return;
    uint32_t rk_[CPT];
    {
        const int32_t *__restrict__ rk = a.rank0 + blk * a.rank0_blk_stride + (int64_t)plane * m;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int col = chunk0 + j < a.n_chunks ? a.slot_col[(chunk0 + j) * 64 + lane] : -1;
            rk_[j] = col >= 0 ? (uint32_t)rk[col] : pad_rank;
        }
    }
    uint64_t *hout = plane ? a.h1 : a.h0;
    for (int64_t row = blk_beg; row < blk_end; ++row) {
        const uint2 *__restrict__ drow = a.dir + (size_t)(2 * (row - a.dir_row0) + plane) * (size_t)nwp;
        const uint32_t n0 = n0tab[2 * (row - a.dir_row0) + plane];
        uint64_t keep = 0;
#pragma unroll 8
        for (int j = 0; j < CPT; ++j) {
            const uint32_t r = rk_[j];
            const uint2 e = drow[r >> 5];
            const uint32_t sh = r & 31u;
            const uint32_t bit = (e.x >> sh) & 1u;
            const uint32_t ob = e.y + (uint32_t)__popc(e.x & ((1u << sh) - 1u));
            rk_[j] = bit ? n0 + ob : r - ob;                             // the LF-mapping step (pbwt.c:148-153)
            const uint64_t bal = __ballot(bit != 0u);
            if (lane == j) keep = bal;
        }
        if (row >= a.row0 && hout && lane < CPT && chunk0 + lane < a.n_chunks)
            hout[(size_t)(row - a.h_row0) * a.n_chunks + chunk0 + lane] = keep;
    }
    if (a.final_rank) {
        int32_t *fin = a.final_rank + (int64_t)bl * a.final_blk_stride + (int64_t)plane * m;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = chunk0 + j;
            if (c < a.n_chunks) {
                const int col = a.slot_col[c * 64 + lane];
                if (col >= 0) fin[col] = (int32_t)rk_[j];
            }
        }
    }
}

void choose_walk_mem_geometry(int m, int n_chunks, int n_blk, Geometry *g)
{
    (void)m;
    g->threads = 256;                                                    // four waves: no LDS, nothing shared -- small workgroups fill the chip
    g->cpt = n_chunks >= 4 * 64 ? 64 : n_chunks >= 4 * 32 ? 32 : 16;
    const int cap = g->threads / 64 * g->cpt;
    g->slices = (n_chunks + cap - 1) / cap;
    g->K = 1; g->wpp = g->threads / 64; g->nbuf = 1; g->tog_off = 0;
    g->dir_stage = 0;
    g->lds_bytes = 0;
    g->workgroups = ((n_blk + 7) / 8) * 16 * g->slices;
}

hipError_t launch_walk_mem(const ScanArgs &a, const Geometry &g, hipStream_t s)
{
#define X(C)                                                                                                        \
    if (g.threads == 256 && g.cpt == C) {                                                                           \
        hipLaunchKernelGGL((walk_mem_kernel<256, C>), dim3(g.workgroups), dim3(256), 0, s, a, a.dir_n0);            \
        return hipGetLastError();                                                                                   \
    }
    X(16) X(32) X(64)
#undef X
    return hipErrorInvalidConfiguration;
}

}  // namespace bgth
