// gfx950 (MI355X, CDNA4) kernels of the BGT genotype-matrix read path.
//
// What the reference does per site and plane (pbwt.c:69-90 full decode, :129-170 subset decode,
// bgt.c:735-757 histogram) is restated here in the inverse / rank-tracking form of SURVEY.md App. B:
//
//     every tracked column i keeps R[i] = its current PBWT rank (thread-private, in VGPRs)
//     per row and plane:   bit  = B[R[i]]                         (B = the row in PBWT order)
//                          R[i] = bit ? n0 + rank1(R[i]) : R[i] - rank1(R[i])
//
// so the permutation state never touches HBM or even LDS.  B is rebuilt per row in LDS from the RLE
// string as a bit-vector with a rank directory: {32 bits, number of ones before them} per 8-byte entry,
// so one ds_read_b64 answers both B[r] and rank1(r).  Work decomposition:
//
//   grid   = (8192-row checkpoint block) x (column slice); all slices of a block are placed on one XCD
//            (workgroup id mod 8) so the RLE bytes they share are served by one L2.
//   wave   = owns CPT consecutive 64-slot chunks; a chunk never mixes sample groups, so the allele
//            counts of a chunk are popcounts of the two 64-bit ballots (the v_cmp that selects the
//            new rank already is the ballot).
//   batch  = K rows: phase A builds the 2K bit-vectors (one wave per plane-row: byte -> run length,
//            wave prefix sum -> run starts, xor-toggle at every change of bit, prefix-xor -> bits,
//            popcount prefix sum -> rank directory), phase B walks the K rows with no barrier.
//
// No MFMA: this is integer/bit work bound by LDS issue and VALU, not by HBM.
#include "scan_kernels.h"

namespace bgth {

// ----------------------------------------------------------------------------------------------------
// wave-level helpers (64 lanes)
// ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lanes_below(uint64_t mask)
{   // number of set bits of mask in lanes below the caller
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ uint32_t wave_incl_add(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}

__device__ __forceinline__ uint32_t rle_len(uint32_t byte)
{   // reference pbwt.c:12-21 as arithmetic: code = byte>>1, len = (code&15) << 4*(code>>4)
    uint32_t code = byte >> 1;
    return (code & 15u) << ((code >> 4) << 2);
}

// ----------------------------------------------------------------------------------------------------
// the scan kernel
// ----------------------------------------------------------------------------------------------------
// ----------------------------------------------------------------------------------------------------
// The row step.  For one tracked column and one plane:   e = B[r>>5] = {32 bits, ones before them}
//     bit = e.bits[r&31] ;  ob = e.before + popc(e.bits & low(r&31)) = rank1(r)
//     r   = bit ? n0 + ob : r - ob                                  (LF-mapping)
// Ten VALU instructions and one ds_read_b64 per lookup; the v_cmp that steers the select is also the
// wave ballot of the decoded bit.  Hand-scheduled: left to hipcc the unrolled row body keeps one
// 64-bit SGPR condition per lookup alive to the end of the row and spills (measured: 151 VGPRs and
// 333 v_writelane/v_readlane at 16 columns per thread; this form needs 2 VGPRs per column + 12).
// Two columns x two planes per statement = 4 LDS reads in flight per wave.  Scratch registers are
// named (v112..v123) and declared as clobbers; every ds_read is waited for inside the statement.
// ----------------------------------------------------------------------------------------------------
#define BGTH_TAIL(R, ELO, EHI, T, MASK, N0)            \
    "v_and_b32 " T ", " T ", " ELO "\n\t"              \
    "v_bcnt_u32_b32 " T ", " T ", " EHI "\n\t"         \
    "v_bfe_u32 " ELO ", " ELO ", " R ", 1\n\t"         \
    "v_cmp_ne_u32_e64 " MASK ", 0, " ELO "\n\t"        \
    "v_sub_u32 " EHI ", " R ", " T "\n\t"              \
    "v_add_u32 " T ", " N0 ", " T "\n\t"               \
    "v_cndmask_b32_e64 " R ", " EHI ", " T ", " MASK "\n\t"

__device__ __forceinline__ void step2(uint32_t &ra0, uint32_t &ra1, uint32_t &rb0, uint32_t &rb1,
                                      uint64_t &ma0, uint64_t &ma1, uint64_t &mb0, uint64_t &mb1,
                                      uint32_t base0, uint32_t base1, uint32_t n00, uint32_t n01)
{
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_lshrrev_b32 v120, 5, %0\n\t"
        "v_lshrrev_b32 v121, 5, %1\n\t"
        "v_lshrrev_b32 v122, 5, %2\n\t"
        "v_lshrrev_b32 v123, 5, %3\n\t"
        "v_lshl_add_u32 v120, v120, 3, %8\n\t"
        "v_lshl_add_u32 v121, v121, 3, %9\n\t"
        "v_lshl_add_u32 v122, v122, 3, %8\n\t"
        "v_lshl_add_u32 v123, v123, 3, %9\n\t"
        "ds_read_b64 v[112:113], v120\n\t"
        "ds_read_b64 v[114:115], v121\n\t"
        "ds_read_b64 v[116:117], v122\n\t"
        "ds_read_b64 v[118:119], v123\n\t"
        "v_bfm_b32 v120, %0, 0\n\t"
        "v_bfm_b32 v121, %1, 0\n\t"
        "v_bfm_b32 v122, %2, 0\n\t"
        "v_bfm_b32 v123, %3, 0\n\t"
        "s_waitcnt lgkmcnt(3)\n\t"
        BGTH_TAIL("%0", "v112", "v113", "v120", "%4", "%10")
        "s_waitcnt lgkmcnt(2)\n\t"
        BGTH_TAIL("%1", "v114", "v115", "v121", "%5", "%11")
        "s_waitcnt lgkmcnt(1)\n\t"
        BGTH_TAIL("%2", "v116", "v117", "v122", "%6", "%10")
        "s_waitcnt lgkmcnt(0)\n\t"
        BGTH_TAIL("%3", "v118", "v119", "v123", "%7", "%11")
        : "+v"(ra0), "+v"(ra1), "+v"(rb0), "+v"(rb1), "=&s"(ma0), "=&s"(ma1), "=&s"(mb0), "=&s"(mb1)
        : "s"(base0), "s"(base1), "s"(n00), "s"(n01)
        : "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123",
          "memory");
}

// Template switches:  MULTI = more than one sample group (per-chunk LDS atomics instead of per-wave
// scalars);  GT = also emit the two bit planes of every row (slot order) for genotype output.
template <int NT, int CPT, bool MULTI, bool GT>
__global__ __launch_bounds__(NT) void scan_kernel(const ScanArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NWAVE = NT / 64;
    static_assert(CPT % 2 == 0, "columns per thread are stepped in pairs");
    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // workgroup -> (block, slice); consecutive workgroup ids go round-robin over the 8 XCDs, so
    // keep  id mod 8  == block mod 8 for every slice of a block.
    const int S   = a.n_slices;
    const int wg  = blockIdx.x;
    const int sup = wg / (8 * S), rem = wg % (8 * S);
    const int slice = rem >> 3;
    const int bl    = sup * 8 + (rem & 7);
    if (bl >= a.n_blk) return;

    // LDS: per plane-row nw entries {bits, ones before} + ONE all-zero sentinel entry.  Padding slots
    // carry the rank 32*nw: they read the sentinel, see bit 0 and "zero ones before", and map to
    // themselves -- so no validity mask is needed anywhere in the row loop.
    const int m = a.m, nw = a.nw, nwp = nw + 1, K = a.K, G = a.G;
    uint2    *BD   = reinterpret_cast<uint2*>(smem);                    // [2K][nwp]
    int32_t  *lcnt = reinterpret_cast<int32_t*>(smem + (((size_t)16 * K * nwp + 15) & ~(size_t)15));
    //   !MULTI: int4 [K][NWAVE] one private slot per wave and row   MULTI: int32 [K][G][3] (LDS atomics)
    uint32_t *n0s  = reinterpret_cast<uint32_t*>(lcnt + (MULTI ? K * G * 3 : K * NWAVE * 4)); // [2K]
    const uint32_t pad_rank = 32u * (uint32_t)nw;
    const uint32_t lds0 = __builtin_amdgcn_groupstaticsize();            // LDS byte address of smem[0]

    const int64_t blk      = (int64_t)a.blk0 + bl;
    const int64_t blk_beg  = blk << a.shift;
    int64_t       blk_end  = (blk + 1) << a.shift;
    if (blk_end > a.row1) blk_end = a.row1;

    // ---- tracked slots of this thread: chunk c = chunk0 + j, slot = 64c + lane
    const int chunk0 = (slice * NWAVE + wave) * CPT;
    uint32_t r0[CPT], r1[CPT];
    {
        const int32_t *rk = a.rank0 + blk * a.rank0_blk_stride;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = chunk0 + j;
            const int col = c < a.n_chunks ? a.slot_col[c * 64 + lane] : -1;
            r0[j] = col >= 0 ? (uint32_t)rk[col] : pad_rank;
            r1[j] = col >= 0 ? (uint32_t)rk[m + col] : pad_rank;
        }
    }
    if (MULTI) for (int i = tid; i < K * G * 3; i += NT) lcnt[i] = 0;
    for (int i = tid; i < 2 * K; i += NT) BD[(size_t)i * nwp + nw] = make_uint2(0u, 0u);

    const uint32_t tail_mask = (m & 31) ? ((1u << (m & 31)) - 1u) : 0xffffffffu;

    for (int64_t rb = blk_beg; rb < blk_end; rb += K) {
        const int Kc = (int)((blk_end - rb) < K ? (blk_end - rb) : K);

        // ================= phase A: build the bit-vectors of Kc rows x 2 planes =================
        for (int p = wave; p < 2 * Kc; p += NWAVE) {
            uint2 *bd = BD + (size_t)p * nwp;
            for (int i = lane; i < nw; i += 64) bd[i] = make_uint2(0u, 0u);
        }
        __syncthreads();
        for (int p = wave; p < 2 * Kc; p += NWAVE) {
            uint2 *bd = BD + (size_t)p * nwp;
            const uint64_t d   = a.rowdesc[2 * rb + p];
            const uint8_t *q   = a.rle + (d & kDescOffMask);
            const uint32_t len = (uint32_t)(d >> kDescLenShift);
            uint32_t pos = 0, prevbit = 0, ones = 0;
            bool stop = false;
            for (uint32_t base = 0; base < len && !stop; base += 64) {
                const uint32_t k = base + lane;
                bool valid = k < len;
                const uint32_t byte = valid ? (uint32_t)q[k] : 0xffu;
                const uint64_t z = __ballot(valid && byte == 0u);   // a zero byte ends the row (pbwt.c:73)
                if (z) { valid = valid && lane < (__ffsll((unsigned long long)z) - 1); stop = true; }
                const uint32_t l = valid ? rle_len(byte) : 0u;
                const uint32_t b = byte & 1u;
                const uint32_t incl  = wave_incl_add(l, lane);
                const uint32_t start = pos + incl - l;
                uint32_t pb = __shfl_up(b, 1);
                if (lane == 0) pb = prevbit;
                if (valid && b != pb && start < (uint32_t)m)
                    atomicXor(&bd[start >> 5].x, 1u << (start & 31));
                const uint32_t incl1 = wave_incl_add(b ? l : 0u, lane);
                ones += __shfl(incl1, 63);
                const int nvalid = __popcll(__ballot(valid));
                if (nvalid) prevbit = __shfl(b, nvalid - 1);
                pos += __shfl(incl, 63);
            }
            if (lane == 0) n0s[p] = (uint32_t)m - ones;
        }
        __syncthreads();
        for (int p = wave; p < 2 * Kc; p += NWAVE) {
            uint2 *bd = BD + (size_t)p * nwp;
            uint32_t carry_x = 0, carry_c = 0;
            for (int base = 0; base < nw; base += 64) {
                const int i = base + lane;
                const bool valid = i < nw;
                const uint32_t t = valid ? bd[i].x : 0u;
                uint32_t x = t;                       // prefix xor inside the word
                x ^= x << 1; x ^= x << 2; x ^= x << 4; x ^= x << 8; x ^= x << 16;
                const uint64_t par = __ballot(__popc(t) & 1);
                const uint32_t cin = (lanes_below(par) ^ carry_x) & 1u;   // parity of toggles before word
                uint32_t w = cin ? ~x : x;
                if (i == nw - 1) w &= tail_mask;
                if (!valid) w = 0u;
                const uint32_t incl = wave_incl_add((uint32_t)__popc(w), lane);
                if (valid) bd[i] = make_uint2(w, carry_c + incl - (uint32_t)__popc(w));
                carry_x ^= (uint32_t)__popcll(par) & 1u;
                carry_c += __shfl(incl, 63);
            }
        }
        __syncthreads();

        // ================= phase B: walk the rows, ranks stay in registers =================
        for (int k = 0; k < Kc; ++k) {
            const uint32_t base0 = lds0 + (uint32_t)(2 * k) * (uint32_t)nwp * 8u;    // LDS byte addresses
            const uint32_t base1 = base0 + (uint32_t)nwp * 8u;
            const uint32_t n00 = __builtin_amdgcn_readfirstlane(n0s[2 * k]);
            const uint32_t n01 = __builtin_amdgcn_readfirstlane(n0s[2 * k + 1]);
            const bool emit = (rb + k) >= a.row0;
            // ones of plane 0, ones of plane 1, ones in both:  n(code1) = ca - cc, n(code2) = cb - cc
            int32_t ca = 0, cb = 0, cc = 0;
            uint64_t keep0 = 0, keep1 = 0;
#pragma unroll
            for (int j = 0; j < CPT; j += 2) {
                uint64_t mA0, mA1, mB0, mB1;
                step2(r0[j], r1[j], r0[j + 1], r1[j + 1], mA0, mA1, mB0, mB1, base0, base1, n00, n01);
                if (GT) {
                    if (lane == j) { keep0 = mA0; keep1 = mA1; }
                    if (lane == j + 1) { keep0 = mB0; keep1 = mB1; }
                }
                if (MULTI) {
                    const int c = chunk0 + j;                                // wave-uniform
                    if (emit && lane == 0) {
                        if (c < a.n_chunks) {
                            int32_t *dst = lcnt + ((size_t)k * G + (a.chunk_desc[c] & 255u)) * 3;
                            atomicAdd(dst + 0, __builtin_amdgcn_readfirstlane(__popcll(mA0 & ~mA1)));
                            atomicAdd(dst + 1, __builtin_amdgcn_readfirstlane(__popcll(~mA0 & mA1)));
                            atomicAdd(dst + 2, __builtin_amdgcn_readfirstlane(__popcll(mA0 & mA1)));
                        }
                        if (c + 1 < a.n_chunks) {
                            int32_t *dst = lcnt + ((size_t)k * G + (a.chunk_desc[c + 1] & 255u)) * 3;
                            atomicAdd(dst + 0, __builtin_amdgcn_readfirstlane(__popcll(mB0 & ~mB1)));
                            atomicAdd(dst + 1, __builtin_amdgcn_readfirstlane(__popcll(~mB0 & mB1)));
                            atomicAdd(dst + 2, __builtin_amdgcn_readfirstlane(__popcll(mB0 & mB1)));
                        }
                    }
                } else {
                    ca += __popcll(mA0) + __popcll(mB0);
                    cb += __popcll(mA1) + __popcll(mB1);
                    cc += __popcll(mA0 & mA1) + __popcll(mB0 & mB1);
                }
            }
            if (!MULTI && lane == 0)
                reinterpret_cast<int4*>(lcnt)[k * NWAVE + wave] = make_int4(ca - cc, cb - cc, cc, 0);
            if (GT && emit && lane < CPT && chunk0 + lane < a.n_chunks) {
                const size_t at = (size_t)(rb + k - a.row0) * a.n_chunks + chunk0 + lane;
                a.h0[at] = keep0;
                a.h1[at] = keep1;
            }
        }
        __syncthreads();

        // ================= phase C: per-row counts of this slice -> HBM =================
        if (MULTI) {
            for (int i = tid; i < Kc * G * 3; i += NT) {
                const int32_t v = lcnt[i];
                if (v) {
                    atomicAdd(a.raw_counts + (size_t)(rb - a.row0) * G * 3 + i, v);
                    lcnt[i] = 0;
                }
            }
        } else {
            for (int i = tid; i < Kc * 3; i += NT) {
                const int k = i / 3, comp = i - 3 * k;
                if (rb + k >= a.row0) {
                    int32_t v = 0;
#pragma unroll
                    for (int w = 0; w < NWAVE; ++w) v += lcnt[(k * NWAVE + w) * 4 + comp];
                    if (v) atomicAdd(a.raw_counts + (size_t)(rb + k - a.row0) * 3 + comp, v);
                }
            }
        }
        // (the barriers of the next phase A order these reads before the next writes to lcnt)
    }

    if (a.final_rank) {
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = chunk0 + j;
            if (c < a.n_chunks) {
                const int col = a.slot_col[c * 64 + lane];
                if (col >= 0) { a.final_rank[col] = (int32_t)r0[j]; a.final_rank[m + col] = (int32_t)r1[j]; }
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------------
// launch geometry
// ----------------------------------------------------------------------------------------------------
static const int kLdsBytes = 160 * 1024;

template <int NT, int CPT, bool MULTI, bool GT>
static hipError_t launch_one(const ScanArgs &a, const Geometry &g, hipStream_t s)
{
    auto fn = scan_kernel<NT, CPT, MULTI, GT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, g.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fn, dim3(g.workgroups), dim3(NT), g.lds_bytes, s, a);
    return hipGetLastError();
}

#define BGTH_GEOMS(X) \
    X(256, 2) X(256, 4) X(256, 8) X(256, 16) \
    X(512, 8) X(512, 16) \
    X(1024, 8) X(1024, 16) X(1024, 24)

struct GeomEntry { int nt, cpt; };
static const GeomEntry kGeoms[] = {
#define X(nt, cpt) {nt, cpt},
    BGTH_GEOMS(X)
#undef X
};

static int lds_need(int nw, int K, int G, int threads)
{
    const int cnt = G > 1 ? K * G * 3 * 4 : K * (threads / 64) * 16;
    return ((16 * K * (nw + 1) + 15) & ~15) + cnt + 2 * K * 4;
}

bool choose_geometry(int m, int n_chunks, int G, int n_blk, int want_threads, int want_cpt, int want_K,
                     Geometry *g)
{
    const int nw = (m + 31) / 32;
    if (lds_need(nw, 1, G, 1024) > kLdsBytes) return false;
    int best = -1;
    long best_cost = 0;
    // enough workgroups to cover the 256 CUs when the file has few blocks
    int want_slices = n_blk >= 256 ? 1 : (256 + n_blk - 1) / n_blk;
    if (want_slices > 8) want_slices = 8;
    for (int i = 0; i < (int)(sizeof(kGeoms) / sizeof(kGeoms[0])); ++i) {
        const int nt = kGeoms[i].nt, cpt = kGeoms[i].cpt;
        if (want_threads && nt != want_threads) continue;
        if (want_cpt && cpt != want_cpt) continue;
        const int cap = nt / 64 * cpt;                       // chunks per workgroup
        const int slices = (n_chunks + cap - 1) / cap;
        const long waste = (long)slices * cap - n_chunks;    // idle chunk slots
        // cost: wasted lanes + distance from the wanted slice count + a bias to mid-size cpt
        long cost = waste * 4 + labs((long)slices - want_slices) * (long)n_chunks / 2;
        if (cpt > 24) cost += n_chunks / 8;
        if (best < 0 || cost < best_cost) best = i, best_cost = cost;
    }
    if (best < 0) return false;
    g->threads = kGeoms[best].nt;
    g->cpt     = kGeoms[best].cpt;
    const int cap = g->threads / 64 * g->cpt;
    g->slices  = (n_chunks + cap - 1) / cap;
    // LDS budget: leave room for 1024/threads workgroups per CU, but at least one row
    int budget = (int)((long)kLdsBytes * g->threads / 1024);
    int K = want_K > 0 ? want_K : 16;
    while (K > 1 && lds_need(nw, K, G, g->threads) > budget) --K;
    while (K > 1 && lds_need(nw, K, G, g->threads) > kLdsBytes) --K;
    g->K = K;
    g->lds_bytes = (lds_need(nw, K, G, g->threads) + 15) & ~15;
    g->workgroups = ((n_blk + 7) / 8) * 8 * g->slices;
    return true;
}

hipError_t launch_scan(const ScanArgs &a, const Geometry &g, hipStream_t s)
{
#define X(NT_, CPT_) if (g.threads == NT_ && g.cpt == CPT_) \
        return a.G > 1 ? (a.h0 ? launch_one<NT_, CPT_, true, true>(a, g, s) : launch_one<NT_, CPT_, true, false>(a, g, s)) \
                       : (a.h0 ? launch_one<NT_, CPT_, false, true>(a, g, s) : launch_one<NT_, CPT_, false, false>(a, g, s));
    BGTH_GEOMS(X)
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

// reference pbwt.c:343: invS[S[i]] = i
__global__ void invert_kernel(const int32_t *perm, int32_t *inv, int m, int64_t total)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t base = i / m * m;
    inv[base + perm[i]] = (int32_t)(i - base);
}

hipError_t launch_invert(const int32_t *perm, int32_t *inv, int m, int64_t n_perm, hipStream_t s)
{
    const int64_t total = n_perm * m;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(invert_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       perm, inv, m, total);
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

}  // namespace bgth
