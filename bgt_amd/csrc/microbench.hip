// Issue-rate calibration for the roofline of the scan kernel (bgth_debug_issue_rate, include/bgt_hip.h).
//
// The scan kernel moves almost nothing through HBM (the permutation lives in registers), so its bound is the
// issue side: VALU wave-instructions per cycle and SIMD, and LDS gathers.  The two guides disagree with the
// first round's assumption about the former (SIMD-32 / 2 cycles per wave64 instruction for v_fma_f32 vs the
// 4 cycles the PMC counters of the scan suggested for its integer mix), so this file MEASURES it: long
// dependency-free streams of one instruction, or of exactly the row step's eight-instruction mix (BGTH_ADDR +
// BGTH_TAIL of scan_device.inc.h, with and without its ds_read_b64), at a chosen number of waves per SIMD.
// Every wave brackets its loop with s_memtime (shader cycles); the host also times the launch with HIP events.
#include "scan_device.inc.h"     // step4<>: the product's own row step, and its instruction macros

namespace bgth {

// one workgroup per CU (the launch asks for more than half the LDS), NT / 256 waves per SIMD
enum { MIX_FMA = 0, MIX_ADD = 1, MIX_STEP = 2, MIX_BCNT = 3, MIX_CMPSEL = 4, MIX_MAD24 = 5, MIX_STEP_LDS_FLAT = 6,
       MIX_STEP_LDS_RANDOM = 7, MIX_LSHL = 8, MIX_LDS_ONLY_FLAT = 9, MIX_LDS_ONLY_RANDOM = 10, MIX_N = 11 };

// eight independent instances of one instruction, registers v40..v47 (+ v48..v55 as second operands)
#define R8(OP) OP(40) OP(41) OP(42) OP(43) OP(44) OP(45) OP(46) OP(47)
#define I_FMA(n)   "v_fma_f32 v" #n ", v" #n ", v48, v49\n\t"
#define I_ADD(n)   "v_add_u32 v" #n ", v" #n ", v48\n\t"
#define I_BCNT(n)  "v_bcnt_u32_b32 v" #n ", v48, v" #n "\n\t"
#define I_MAD24(n) "v_mad_i32_i24 v" #n ", v" #n ", -8, v48\n\t"
#define I_LSHL(n)  "v_lshlrev_b32 v" #n ", v48, v" #n "\n\t"
#define CLOB8 "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49"

template <int MIX>
__global__ void issue_rate_kernel(int iters, unsigned long long *cycles, uint32_t *sink, uint32_t seed)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint2 *tab = reinterpret_cast<uint2*>(smem);
    constexpr int NE = 4096;                                  // 32 KB of {bits, before} entries: a row of m = 131072 columns
    constexpr bool RANDOM = MIX == MIX_STEP_LDS_RANDOM || MIX == MIX_LDS_ONLY_RANDOM;
    if (MIX == MIX_STEP_LDS_FLAT || MIX == MIX_STEP_LDS_RANDOM || MIX == MIX_LDS_ONLY_FLAT || MIX == MIX_LDS_ONLY_RANDOM) {
        // a VALID directory, so that the row step is a true LF-mapping and the ranks stay inside the row for any number
        // of steps: RANDOM = pseudo-random bits with their prefix popcounts; FLAT = the all-zero row (ranks never move)
        for (int i = threadIdx.x; i < NE; i += blockDim.x) tab[i] = make_uint2(RANDOM ? 0x9e3779b9u * (uint32_t)(i + 1) * (uint32_t)(i + 7) : 0u, 0u);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t run = 0;
            for (int i = 0; i < NE; ++i) { tab[i].y = run; run += (uint32_t)__popc(tab[i].x); }
            tab[NE] = make_uint2((uint32_t)NE * 32u - run, 0u);             // number of zeros
        }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    uint32_t q[8];
    for (int i = 0; i < 8; ++i) {
        // FLAT: lane l reads entry l (+ 64 i): conflict-free; RANDOM: a different pseudo-random entry per lane and step
        uint32_t r = RANDOM
                         ? (((uint32_t)(threadIdx.x * 8 + i) * 2654435761u) ^ seed) % (NE * 32u)
                         : (uint32_t)((lane + 64 * i) * 32 + (lane & 31));
        q[i] = ~r;
    }
    const uint32_t base = __builtin_amdgcn_groupstaticsize() - 8u;        // as the scan kernel: (row address) - 8
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if constexpr (MIX == MIX_FMA || MIX == MIX_ADD || MIX == MIX_BCNT || MIX == MIX_MAD24 || MIX == MIX_LSHL) {
        asm volatile("v_mov_b32 v48, 3\n\tv_mov_b32 v49, 1.0\n\t"
                     "v_mov_b32 v40, 0\n\tv_mov_b32 v41, 1\n\tv_mov_b32 v42, 2\n\tv_mov_b32 v43, 3\n\t"
                     "v_mov_b32 v44, 4\n\tv_mov_b32 v45, 5\n\tv_mov_b32 v46, 6\n\tv_mov_b32 v47, 7\n\t" ::: CLOB8);
        for (int it = 0; it < iters; ++it) {
            if constexpr (MIX == MIX_FMA)   asm volatile(R8(I_FMA) R8(I_FMA) R8(I_FMA) R8(I_FMA) R8(I_FMA) R8(I_FMA) R8(I_FMA) R8(I_FMA) ::: CLOB8);
            if constexpr (MIX == MIX_ADD)   asm volatile(R8(I_ADD) R8(I_ADD) R8(I_ADD) R8(I_ADD) R8(I_ADD) R8(I_ADD) R8(I_ADD) R8(I_ADD) ::: CLOB8);
            if constexpr (MIX == MIX_BCNT)  asm volatile(R8(I_BCNT) R8(I_BCNT) R8(I_BCNT) R8(I_BCNT) R8(I_BCNT) R8(I_BCNT) R8(I_BCNT) R8(I_BCNT) ::: CLOB8);
            if constexpr (MIX == MIX_MAD24) asm volatile(R8(I_MAD24) R8(I_MAD24) R8(I_MAD24) R8(I_MAD24) R8(I_MAD24) R8(I_MAD24) R8(I_MAD24) R8(I_MAD24) ::: CLOB8);
            if constexpr (MIX == MIX_LSHL)  asm volatile(R8(I_LSHL) R8(I_LSHL) R8(I_LSHL) R8(I_LSHL) R8(I_LSHL) R8(I_LSHL) R8(I_LSHL) R8(I_LSHL) ::: CLOB8);
        }
        asm volatile("v_xor_b32 %0, v40, v47" : "=v"(acc) :: CLOB8);
    } else if constexpr (MIX == MIX_CMPSEL) {
        uint32_t a = (uint32_t)lane, b = 5u, c = 9u;
        for (int it = 0; it < iters; ++it) {
#define CS2 "v_cmp_gt_i32_e64 s[20:21], 0, %0\n\tv_cndmask_b32_e64 %1, %1, %2, s[20:21]\n\t" \
            "v_cmp_gt_i32_e64 s[22:23], 0, %1\n\tv_cndmask_b32_e64 %2, %2, %0, s[22:23]\n\t"
            asm volatile(CS2 CS2 CS2 CS2 CS2 CS2 CS2 CS2 CS2 CS2 CS2 CS2 CS2 CS2 CS2 CS2
                         : "+v"(a), "+v"(b), "+v"(c) :: "s20", "s21", "s22", "s23");
#undef CS2
        }
        acc = a ^ b ^ c;
    } else {
        // the row step on 4 columns x 2 planes = 8 lookups per statement (step4 of scan_device.inc.h, the statement the
        // scan kernel's walk is made of, SALU count accumulation included), four statements per loop iteration
        constexpr bool LDS = MIX == MIX_STEP_LDS_FLAT || MIX == MIX_STEP_LDS_RANDOM;
        constexpr bool ONLY = MIX == MIX_LDS_ONLY_FLAT || MIX == MIX_LDS_ONLY_RANDOM;
        uint32_t r0[4] = {q[0], q[2], q[4], q[6]}, r1[4] = {q[1], q[3], q[5], q[7]};
        uint64_t m0[4], m1[4];
        uint32_t ca = 0, cb = 0, cc = 0;
        const uint32_t nn0 = (LDS || ONLY) ? 0u - (uint32_t)__builtin_amdgcn_readfirstlane((int)tab[NE].x) : 0u - 77u;
        const uint32_t base0 = base, base1 = base, n00 = nn0, n01 = nn0;     // operand names of BGTH_STEP4_OPERANDS
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep) {
                if constexpr (ONLY) {
                    // eight gathers and their wait, nothing else: what the LDS alone sustains at this address pattern
                    asm volatile(BGTH_ADDR("v104", "%0", "%8") BGTH_ADDR("v106", "%1", "%8") BGTH_ADDR("v108", "%2", "%8") BGTH_ADDR("v110", "%3", "%8")
                                 BGTH_ADDR("v112", "%4", "%8") BGTH_ADDR("v114", "%5", "%8") BGTH_ADDR("v116", "%6", "%8") BGTH_ADDR("v118", "%7", "%8")
                                 "ds_read_b64 v[104:105], v104\n\tds_read_b64 v[106:107], v106\n\tds_read_b64 v[108:109], v108\n\t"
                                 "ds_read_b64 v[110:111], v110\n\tds_read_b64 v[112:113], v112\n\tds_read_b64 v[114:115], v114\n\t"
                                 "ds_read_b64 v[116:117], v116\n\tds_read_b64 v[118:119], v118\n\ts_waitcnt lgkmcnt(0)\n\t"
                                 : "+v"(r0[0]), "+v"(r1[0]), "+v"(r0[1]), "+v"(r1[1]), "+v"(r0[2]), "+v"(r1[2]), "+v"(r0[3]), "+v"(r1[3])
                                 : "s"(base), "v"(0)
                                 : "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115",
                                   "v116", "v117", "v118", "v119", "memory");
                } else if constexpr (LDS) {
                    step4<false>(r0, r1, m0, m1, ca, cb, cc, base, base, nn0, nn0);
                } else {
                    // the same statement without its eight ds_read_b64 (and waits): the entries are whatever v104.. hold
                    asm volatile(BGTH_ADDR("v104", "%0", "%19") BGTH_ADDR("v106", "%1", "%20") BGTH_ADDR("v108", "%2", "%19") BGTH_ADDR("v110", "%3", "%20")
                                 BGTH_ADDR("v112", "%4", "%19") BGTH_ADDR("v114", "%5", "%20") BGTH_ADDR("v116", "%6", "%19") BGTH_ADDR("v118", "%7", "%20")
                                 BGTH_TAIL("%0", "v104", "v105", "v104", "%8", "%21") BGTH_TAIL("%1", "v106", "v107", "v106", "%9", "%22")
                                 BGTH_COUNT("%8", "%9", "%16", "%17", "%18")
                                 BGTH_TAIL("%2", "v108", "v109", "v108", "%10", "%21") BGTH_TAIL("%3", "v110", "v111", "v110", "%11", "%22")
                                 BGTH_COUNT("%10", "%11", "%16", "%17", "%18")
                                 BGTH_TAIL("%4", "v112", "v113", "v112", "%12", "%21") BGTH_TAIL("%5", "v114", "v115", "v114", "%13", "%22")
                                 BGTH_COUNT("%12", "%13", "%16", "%17", "%18")
                                 BGTH_TAIL("%6", "v116", "v117", "v116", "%14", "%21") BGTH_TAIL("%7", "v118", "v119", "v118", "%15", "%22")
                                 BGTH_COUNT("%14", "%15", "%16", "%17", "%18")
                                 BGTH_STEP4_OPERANDS);
                }
            }
        }
        for (int i = 0; i < 4; ++i) acc ^= r0[i] ^ r1[i];
        acc ^= ca ^ cb ^ cc;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) atomicMax(cycles, t1 - t0);
    if (acc == 0x9e3779b9u) *sink = acc;
}

struct MixInfo { const char *name; int valu_per_iter; int lds_per_iter; };
static const MixInfo kMix[MIX_N] = {
    {"v_fma_f32", 64, 0}, {"v_add_u32", 64, 0}, {"row step (8 VALU), no LDS", 4 * 8 * 8, 0}, {"v_bcnt_u32_b32", 64, 0},
    {"v_cmp_gt_i32_e64 + v_cndmask_b32_e64", 64, 0}, {"v_mad_i32_i24", 64, 0},
    {"row step + ds_read_b64, conflict-free", 4 * 8 * 8, 4 * 8}, {"row step + ds_read_b64, random entries", 4 * 8 * 8, 4 * 8},
    {"v_lshlrev_b32", 64, 0}, {"ds_read_b64 + its 2 address VALU, conflict-free", 4 * 16, 4 * 8}, {"ds_read_b64 + its 2 address VALU, random entries", 4 * 16, 4 * 8},
};

template <int MIX>
static hipError_t launch_mix(int threads, int iters, unsigned long long *cyc, uint32_t *sink, hipStream_t s)
{
    auto fn = issue_rate_kernel<MIX>;
    const int lds = 96 * 1024;                                  // more than half the LDS: one workgroup per CU
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fn, dim3(256), dim3(threads), lds, s, iters, cyc, sink, 0x5bd1e995u);
    return hipGetLastError();
}

// out[0] = cycles of the slowest wave (s_memtime), out[1] = milliseconds (HIP events), out[2] = VALU wave-instructions
// per wave, out[3] = LDS wave-instructions per wave
hipError_t run_issue_rate(int mix, int waves_per_simd, int iters, double out[4])
{
    if (mix < 0 || mix >= MIX_N || waves_per_simd < 1 || waves_per_simd > 4) return hipErrorInvalidValue;
    unsigned long long *cyc = nullptr; uint32_t *sink = nullptr;
    hipEvent_t e0, e1;
    hipError_t e;
    if ((e = hipMalloc((void**)&cyc, 8)) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&sink, 4)) != hipSuccess) return e;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int threads = 256 * waves_per_simd;
    float ms = 0;
    for (int pass = 0; pass < 2; ++pass) {                       // pass 0 warms up (code object load, clocks)
        hipMemset(cyc, 0, 8);
        hipEventRecord(e0, nullptr);
        switch (mix) {
#define X(M) case M: e = launch_mix<M>(threads, iters, cyc, sink, nullptr); break;
        X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10)
#undef X
        }
        hipEventRecord(e1, nullptr);
        if (e != hipSuccess) break;
        if ((e = hipEventSynchronize(e1)) != hipSuccess) break;
        hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long h = 0;
    if (e == hipSuccess) e = hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    out[0] = (double)h; out[1] = ms;
    out[2] = (double)kMix[mix].valu_per_iter * iters; out[3] = (double)kMix[mix].lds_per_iter * iters;
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(cyc); hipFree(sink);
    return e;
}

const char *issue_rate_mix_name(int mix) { return mix >= 0 && mix < MIX_N ? kMix[mix].name : nullptr; }

}  // namespace bgth
