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
#include "scan_step_variants.inc.h"

namespace bgth {

// one workgroup per CU (the launch asks for more than half the LDS), NT / 256 waves per SIMD
enum { MIX_FMA = 0, MIX_ADD = 1, MIX_STEP = 2, MIX_BCNT = 3, MIX_CMPSEL = 4, MIX_MAD24 = 5, MIX_STEP_LDS_FLAT = 6,
       MIX_STEP_LDS_RANDOM = 7, MIX_LSHL = 8, MIX_LDS_ONLY_FLAT = 9, MIX_LDS_ONLY_RANDOM = 10,
       MIX_IL2 = 11, MIX_IL4 = 12, MIX_IL8 = 13, MIX_STEP_LDS_CLUSTERED = 14, MIX_N = 15 };

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
    constexpr bool IL = MIX == MIX_IL2 || MIX == MIX_IL4 || MIX == MIX_IL8;
    constexpr bool RANDOM = MIX == MIX_STEP_LDS_RANDOM || MIX == MIX_LDS_ONLY_RANDOM || IL;
    constexpr bool CLUSTERED = MIX == MIX_STEP_LDS_CLUSTERED;   // long runs + 64 consecutive ranks per wave and lookup: what rank-ordered slots would see
    if (MIX == MIX_STEP_LDS_FLAT || MIX == MIX_STEP_LDS_RANDOM || MIX == MIX_LDS_ONLY_FLAT || MIX == MIX_LDS_ONLY_RANDOM || IL || CLUSTERED) {
        // a VALID directory, so that the row step is a true LF-mapping and the ranks stay inside the row for any number
        // of steps: RANDOM = pseudo-random bits with their prefix popcounts; FLAT = the all-zero row (ranks never move)
        // (CLUSTERED: four runs of 1024 words -- neighbours in rank stay neighbours under the LF-mapping unless a run boundary falls between them)
        for (int i = threadIdx.x; i < NE; i += blockDim.x) tab[i] = make_uint2(RANDOM ? 0x9e3779b9u * (uint32_t)(i + 1) * (uint32_t)(i + 7) : CLUSTERED ? 0u - (uint32_t)((i >> 10) & 1) : 0u, 0u);
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
                         : CLUSTERED ? ((((uint32_t)((threadIdx.x >> 6) * 8 + i) * 2654435761u) ^ seed) % (NE * 32u - 64u)) + (uint32_t)lane
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
        constexpr bool LDS = MIX == MIX_STEP_LDS_FLAT || MIX == MIX_STEP_LDS_RANDOM || MIX == MIX_STEP_LDS_CLUSTERED;
        constexpr bool ONLY = MIX == MIX_LDS_ONLY_FLAT || MIX == MIX_LDS_ONLY_RANDOM;
        uint32_t r0[4] = {q[0], q[2], q[4], q[6]}, r1[4] = {q[1], q[3], q[5], q[7]};
        uint64_t m0[4], m1[4];
        uint32_t ca = 0, cb = 0, cc = 0;
        const uint32_t nn0 = (LDS || ONLY || IL) ? 0u - (uint32_t)__builtin_amdgcn_readfirstlane((int)tab[NE].x) : 0u - 77u;
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
                } else if constexpr (MIX == MIX_IL2) {
                    asm volatile("s_waitcnt lgkmcnt(0)\n\t" BGTH_STEP4_IL2 BGTH_STEP4_OPERANDS);
                } else if constexpr (MIX == MIX_IL4) {
                    asm volatile("s_waitcnt lgkmcnt(0)\n\t" BGTH_STEP4_IL4 BGTH_STEP4_OPERANDS);
                } else if constexpr (MIX == MIX_IL8) {
                    asm volatile("s_waitcnt lgkmcnt(0)\n\t" BGTH_STEP4_IL8 BGTH_STEP4_OPERANDS);
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

// ---- instruction classes: which VALU opcodes issue in 2 cycles per wave64 and which in 4 (or more) ----
// ONE instruction, 8 independent instances (destinations v40..v47; 64-bit forms v[40:41]..v[54:55]) x 8 per loop
// iteration; second / third sources v48 / v49 (v[56:59] for the 64-bit forms), condition / carry operands vcc, s[20:23].
template <int OP>
__device__ __forceinline__ void op_block()
{
    if constexpr (OP == 0) asm volatile("v_mov_b32 v40, v48\n\tv_mov_b32 v41, v48\n\tv_mov_b32 v42, v48\n\tv_mov_b32 v43, v48\n\tv_mov_b32 v44, v48\n\tv_mov_b32 v45, v48\n\tv_mov_b32 v46, v48\n\tv_mov_b32 v47, v48\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 1) asm volatile("v_and_b32 v40, v48, v40\n\tv_and_b32 v41, v48, v41\n\tv_and_b32 v42, v48, v42\n\tv_and_b32 v43, v48, v43\n\tv_and_b32 v44, v48, v44\n\tv_and_b32 v45, v48, v45\n\tv_and_b32 v46, v48, v46\n\tv_and_b32 v47, v48, v47\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 2) asm volatile("v_or_b32 v40, v48, v40\n\tv_or_b32 v41, v48, v41\n\tv_or_b32 v42, v48, v42\n\tv_or_b32 v43, v48, v43\n\tv_or_b32 v44, v48, v44\n\tv_or_b32 v45, v48, v45\n\tv_or_b32 v46, v48, v46\n\tv_or_b32 v47, v48, v47\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 3) asm volatile("v_xor_b32 v40, v48, v40\n\tv_xor_b32 v41, v48, v41\n\tv_xor_b32 v42, v48, v42\n\tv_xor_b32 v43, v48, v43\n\tv_xor_b32 v44, v48, v44\n\tv_xor_b32 v45, v48, v45\n\tv_xor_b32 v46, v48, v46\n\tv_xor_b32 v47, v48, v47\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 4) asm volatile("v_sub_u32 v40, v48, v40\n\tv_sub_u32 v41, v48, v41\n\tv_sub_u32 v42, v48, v42\n\tv_sub_u32 v43, v48, v43\n\tv_sub_u32 v44, v48, v44\n\tv_sub_u32 v45, v48, v45\n\tv_sub_u32 v46, v48, v46\n\tv_sub_u32 v47, v48, v47\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 5) asm volatile("v_lshrrev_b32 v40, 5, v40\n\tv_lshrrev_b32 v41, 5, v41\n\tv_lshrrev_b32 v42, 5, v42\n\tv_lshrrev_b32 v43, 5, v43\n\tv_lshrrev_b32 v44, 5, v44\n\tv_lshrrev_b32 v45, 5, v45\n\tv_lshrrev_b32 v46, 5, v46\n\tv_lshrrev_b32 v47, 5, v47\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 6) asm volatile("v_ashrrev_i32 v40, 5, v40\n\tv_ashrrev_i32 v41, 5, v41\n\tv_ashrrev_i32 v42, 5, v42\n\tv_ashrrev_i32 v43, 5, v43\n\tv_ashrrev_i32 v44, 5, v44\n\tv_ashrrev_i32 v45, 5, v45\n\tv_ashrrev_i32 v46, 5, v46\n\tv_ashrrev_i32 v47, 5, v47\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 7) asm volatile("v_lshlrev_b32 v40, 3, v40\n\tv_lshlrev_b32 v41, 3, v41\n\tv_lshlrev_b32 v42, 3, v42\n\tv_lshlrev_b32 v43, 3, v43\n\tv_lshlrev_b32 v44, 3, v44\n\tv_lshlrev_b32 v45, 3, v45\n\tv_lshlrev_b32 v46, 3, v46\n\tv_lshlrev_b32 v47, 3, v47\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 8) asm volatile("v_lshlrev_b32 v40, v48, v40\n\tv_lshlrev_b32 v41, v48, v41\n\tv_lshlrev_b32 v42, v48, v42\n\tv_lshlrev_b32 v43, v48, v43\n\tv_lshlrev_b32 v44, v48, v44\n\tv_lshlrev_b32 v45, v48, v45\n\tv_lshlrev_b32 v46, v48, v46\n\tv_lshlrev_b32 v47, v48, v47\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 9) asm volatile("v_lshl_add_u32 v40, v40, 3, v48\n\tv_lshl_add_u32 v41, v41, 3, v48\n\tv_lshl_add_u32 v42, v42, 3, v48\n\tv_lshl_add_u32 v43, v43, 3, v48\n\tv_lshl_add_u32 v44, v44, 3, v48\n\tv_lshl_add_u32 v45, v45, 3, v48\n\tv_lshl_add_u32 v46, v46, 3, v48\n\tv_lshl_add_u32 v47, v47, 3, v48\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 10) asm volatile("v_add_lshl_u32 v40, v40, v48, 3\n\tv_add_lshl_u32 v41, v41, v48, 3\n\tv_add_lshl_u32 v42, v42, v48, 3\n\tv_add_lshl_u32 v43, v43, v48, 3\n\tv_add_lshl_u32 v44, v44, v48, 3\n\tv_add_lshl_u32 v45, v45, v48, 3\n\tv_add_lshl_u32 v46, v46, v48, 3\n\tv_add_lshl_u32 v47, v47, v48, 3\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 11) asm volatile("v_and_or_b32 v40, v40, v48, v49\n\tv_and_or_b32 v41, v41, v48, v49\n\tv_and_or_b32 v42, v42, v48, v49\n\tv_and_or_b32 v43, v43, v48, v49\n\tv_and_or_b32 v44, v44, v48, v49\n\tv_and_or_b32 v45, v45, v48, v49\n\tv_and_or_b32 v46, v46, v48, v49\n\tv_and_or_b32 v47, v47, v48, v49\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 12) asm volatile("v_lshl_or_b32 v40, v40, 3, v48\n\tv_lshl_or_b32 v41, v41, 3, v48\n\tv_lshl_or_b32 v42, v42, 3, v48\n\tv_lshl_or_b32 v43, v43, 3, v48\n\tv_lshl_or_b32 v44, v44, 3, v48\n\tv_lshl_or_b32 v45, v45, 3, v48\n\tv_lshl_or_b32 v46, v46, 3, v48\n\tv_lshl_or_b32 v47, v47, 3, v48\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 13) asm volatile("v_or3_b32 v40, v40, v48, v49\n\tv_or3_b32 v41, v41, v48, v49\n\tv_or3_b32 v42, v42, v48, v49\n\tv_or3_b32 v43, v43, v48, v49\n\tv_or3_b32 v44, v44, v48, v49\n\tv_or3_b32 v45, v45, v48, v49\n\tv_or3_b32 v46, v46, v48, v49\n\tv_or3_b32 v47, v47, v48, v49\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 14) asm volatile("v_add3_u32 v40, v40, v48, v49\n\tv_add3_u32 v41, v41, v48, v49\n\tv_add3_u32 v42, v42, v48, v49\n\tv_add3_u32 v43, v43, v48, v49\n\tv_add3_u32 v44, v44, v48, v49\n\tv_add3_u32 v45, v45, v48, v49\n\tv_add3_u32 v46, v46, v48, v49\n\tv_add3_u32 v47, v47, v48, v49\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 15) asm volatile("v_bfe_u32 v40, v40, 5, 20\n\tv_bfe_u32 v41, v41, 5, 20\n\tv_bfe_u32 v42, v42, 5, 20\n\tv_bfe_u32 v43, v43, 5, 20\n\tv_bfe_u32 v44, v44, 5, 20\n\tv_bfe_u32 v45, v45, 5, 20\n\tv_bfe_u32 v46, v46, 5, 20\n\tv_bfe_u32 v47, v47, 5, 20\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 16) asm volatile("v_bfe_i32 v40, v40, 5, 20\n\tv_bfe_i32 v41, v41, 5, 20\n\tv_bfe_i32 v42, v42, 5, 20\n\tv_bfe_i32 v43, v43, 5, 20\n\tv_bfe_i32 v44, v44, 5, 20\n\tv_bfe_i32 v45, v45, 5, 20\n\tv_bfe_i32 v46, v46, 5, 20\n\tv_bfe_i32 v47, v47, 5, 20\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 17) asm volatile("v_bfi_b32 v40, v48, v40, v49\n\tv_bfi_b32 v41, v48, v41, v49\n\tv_bfi_b32 v42, v48, v42, v49\n\tv_bfi_b32 v43, v48, v43, v49\n\tv_bfi_b32 v44, v48, v44, v49\n\tv_bfi_b32 v45, v48, v45, v49\n\tv_bfi_b32 v46, v48, v46, v49\n\tv_bfi_b32 v47, v48, v47, v49\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 18) asm volatile("v_alignbit_b32 v40, v48, v40, 5\n\tv_alignbit_b32 v41, v48, v41, 5\n\tv_alignbit_b32 v42, v48, v42, 5\n\tv_alignbit_b32 v43, v48, v43, 5\n\tv_alignbit_b32 v44, v48, v44, 5\n\tv_alignbit_b32 v45, v48, v45, 5\n\tv_alignbit_b32 v46, v48, v46, 5\n\tv_alignbit_b32 v47, v48, v47, 5\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 19) asm volatile("v_perm_b32 v40, v40, v48, v49\n\tv_perm_b32 v41, v41, v48, v49\n\tv_perm_b32 v42, v42, v48, v49\n\tv_perm_b32 v43, v43, v48, v49\n\tv_perm_b32 v44, v44, v48, v49\n\tv_perm_b32 v45, v45, v48, v49\n\tv_perm_b32 v46, v46, v48, v49\n\tv_perm_b32 v47, v47, v48, v49\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 20) asm volatile("v_bitop3_b32 v40, v40, v48, v49 bitop3:0x36\n\tv_bitop3_b32 v41, v41, v48, v49 bitop3:0x36\n\tv_bitop3_b32 v42, v42, v48, v49 bitop3:0x36\n\tv_bitop3_b32 v43, v43, v48, v49 bitop3:0x36\n\tv_bitop3_b32 v44, v44, v48, v49 bitop3:0x36\n\tv_bitop3_b32 v45, v45, v48, v49 bitop3:0x36\n\tv_bitop3_b32 v46, v46, v48, v49 bitop3:0x36\n\tv_bitop3_b32 v47, v47, v48, v49 bitop3:0x36\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 21) asm volatile("v_cndmask_b32 v40, v48, v40, vcc\n\tv_cndmask_b32 v41, v48, v41, vcc\n\tv_cndmask_b32 v42, v48, v42, vcc\n\tv_cndmask_b32 v43, v48, v43, vcc\n\tv_cndmask_b32 v44, v48, v44, vcc\n\tv_cndmask_b32 v45, v48, v45, vcc\n\tv_cndmask_b32 v46, v48, v46, vcc\n\tv_cndmask_b32 v47, v48, v47, vcc\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 22) asm volatile("v_cndmask_b32_e64 v40, v48, v40, s[20:21]\n\tv_cndmask_b32_e64 v41, v48, v41, s[20:21]\n\tv_cndmask_b32_e64 v42, v48, v42, s[20:21]\n\tv_cndmask_b32_e64 v43, v48, v43, s[20:21]\n\tv_cndmask_b32_e64 v44, v48, v44, s[20:21]\n\tv_cndmask_b32_e64 v45, v48, v45, s[20:21]\n\tv_cndmask_b32_e64 v46, v48, v46, s[20:21]\n\tv_cndmask_b32_e64 v47, v48, v47, s[20:21]\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 23) asm volatile("v_cmp_gt_i32 vcc, 0, v40\n\tv_cmp_gt_i32 vcc, 0, v41\n\tv_cmp_gt_i32 vcc, 0, v42\n\tv_cmp_gt_i32 vcc, 0, v43\n\tv_cmp_gt_i32 vcc, 0, v44\n\tv_cmp_gt_i32 vcc, 0, v45\n\tv_cmp_gt_i32 vcc, 0, v46\n\tv_cmp_gt_i32 vcc, 0, v47\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 24) asm volatile("v_cmp_gt_i32_e64 s[22:23], 0, v40\n\tv_cmp_gt_i32_e64 s[22:23], 0, v41\n\tv_cmp_gt_i32_e64 s[22:23], 0, v42\n\tv_cmp_gt_i32_e64 s[22:23], 0, v43\n\tv_cmp_gt_i32_e64 s[22:23], 0, v44\n\tv_cmp_gt_i32_e64 s[22:23], 0, v45\n\tv_cmp_gt_i32_e64 s[22:23], 0, v46\n\tv_cmp_gt_i32_e64 s[22:23], 0, v47\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 25) asm volatile("v_mul_u32_u24 v40, v48, v40\n\tv_mul_u32_u24 v41, v48, v41\n\tv_mul_u32_u24 v42, v48, v42\n\tv_mul_u32_u24 v43, v48, v43\n\tv_mul_u32_u24 v44, v48, v44\n\tv_mul_u32_u24 v45, v48, v45\n\tv_mul_u32_u24 v46, v48, v46\n\tv_mul_u32_u24 v47, v48, v47\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 26) asm volatile("v_mad_u32_u24 v40, v40, 8, v48\n\tv_mad_u32_u24 v41, v41, 8, v48\n\tv_mad_u32_u24 v42, v42, 8, v48\n\tv_mad_u32_u24 v43, v43, 8, v48\n\tv_mad_u32_u24 v44, v44, 8, v48\n\tv_mad_u32_u24 v45, v45, 8, v48\n\tv_mad_u32_u24 v46, v46, 8, v48\n\tv_mad_u32_u24 v47, v47, 8, v48\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 27) asm volatile("v_mad_i32_i24 v40, v40, -8, v48\n\tv_mad_i32_i24 v41, v41, -8, v48\n\tv_mad_i32_i24 v42, v42, -8, v48\n\tv_mad_i32_i24 v43, v43, -8, v48\n\tv_mad_i32_i24 v44, v44, -8, v48\n\tv_mad_i32_i24 v45, v45, -8, v48\n\tv_mad_i32_i24 v46, v46, -8, v48\n\tv_mad_i32_i24 v47, v47, -8, v48\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 28) asm volatile("v_mul_lo_u32 v40, v40, v48\n\tv_mul_lo_u32 v41, v41, v48\n\tv_mul_lo_u32 v42, v42, v48\n\tv_mul_lo_u32 v43, v43, v48\n\tv_mul_lo_u32 v44, v44, v48\n\tv_mul_lo_u32 v45, v45, v48\n\tv_mul_lo_u32 v46, v46, v48\n\tv_mul_lo_u32 v47, v47, v48\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 29) asm volatile("v_mul_hi_u32 v40, v40, v48\n\tv_mul_hi_u32 v41, v41, v48\n\tv_mul_hi_u32 v42, v42, v48\n\tv_mul_hi_u32 v43, v43, v48\n\tv_mul_hi_u32 v44, v44, v48\n\tv_mul_hi_u32 v45, v45, v48\n\tv_mul_hi_u32 v46, v46, v48\n\tv_mul_hi_u32 v47, v47, v48\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 30) asm volatile("v_min_u32 v40, v48, v40\n\tv_min_u32 v41, v48, v41\n\tv_min_u32 v42, v48, v42\n\tv_min_u32 v43, v48, v43\n\tv_min_u32 v44, v48, v44\n\tv_min_u32 v45, v48, v45\n\tv_min_u32 v46, v48, v46\n\tv_min_u32 v47, v48, v47\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 31) asm volatile("v_mbcnt_lo_u32_b32 v40, v48, v40\n\tv_mbcnt_lo_u32_b32 v41, v48, v41\n\tv_mbcnt_lo_u32_b32 v42, v48, v42\n\tv_mbcnt_lo_u32_b32 v43, v48, v43\n\tv_mbcnt_lo_u32_b32 v44, v48, v44\n\tv_mbcnt_lo_u32_b32 v45, v48, v45\n\tv_mbcnt_lo_u32_b32 v46, v48, v46\n\tv_mbcnt_lo_u32_b32 v47, v48, v47\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 32) asm volatile("v_ffbh_u32 v40, v40\n\tv_ffbh_u32 v41, v41\n\tv_ffbh_u32 v42, v42\n\tv_ffbh_u32 v43, v43\n\tv_ffbh_u32 v44, v44\n\tv_ffbh_u32 v45, v45\n\tv_ffbh_u32 v46, v46\n\tv_ffbh_u32 v47, v47\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 33) asm volatile("v_bcnt_u32_b32 v40, v48, v40\n\tv_bcnt_u32_b32 v41, v48, v41\n\tv_bcnt_u32_b32 v42, v48, v42\n\tv_bcnt_u32_b32 v43, v48, v43\n\tv_bcnt_u32_b32 v44, v48, v44\n\tv_bcnt_u32_b32 v45, v48, v45\n\tv_bcnt_u32_b32 v46, v48, v46\n\tv_bcnt_u32_b32 v47, v48, v47\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 34) asm volatile("v_add_co_u32 v40, vcc, v48, v40\n\tv_add_co_u32 v41, vcc, v48, v41\n\tv_add_co_u32 v42, vcc, v48, v42\n\tv_add_co_u32 v43, vcc, v48, v43\n\tv_add_co_u32 v44, vcc, v48, v44\n\tv_add_co_u32 v45, vcc, v48, v45\n\tv_add_co_u32 v46, vcc, v48, v46\n\tv_add_co_u32 v47, vcc, v48, v47\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 35) asm volatile("v_xad_u32 v40, v40, v48, v49\n\tv_xad_u32 v41, v41, v48, v49\n\tv_xad_u32 v42, v42, v48, v49\n\tv_xad_u32 v43, v43, v48, v49\n\tv_xad_u32 v44, v44, v48, v49\n\tv_xad_u32 v45, v45, v48, v49\n\tv_xad_u32 v46, v46, v48, v49\n\tv_xad_u32 v47, v47, v48, v49\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 36) asm volatile("v_sad_u32 v40, v40, v48, v49\n\tv_sad_u32 v41, v41, v48, v49\n\tv_sad_u32 v42, v42, v48, v49\n\tv_sad_u32 v43, v43, v48, v49\n\tv_sad_u32 v44, v44, v48, v49\n\tv_sad_u32 v45, v45, v48, v49\n\tv_sad_u32 v46, v46, v48, v49\n\tv_sad_u32 v47, v47, v48, v49\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 37) asm volatile("v_add_u32_dpp v40, v48, v40 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp v41, v48, v41 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp v42, v48, v42 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp v43, v48, v43 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp v44, v48, v44 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp v45, v48, v45 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp v46, v48, v46 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp v47, v48, v47 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 38) asm volatile("v_mov_b32_dpp v40, v48 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp v41, v48 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp v42, v48 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp v43, v48 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp v44, v48 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp v45, v48 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp v46, v48 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp v47, v48 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 39) asm volatile("v_pk_add_u16 v40, v40, v48\n\tv_pk_add_u16 v41, v41, v48\n\tv_pk_add_u16 v42, v42, v48\n\tv_pk_add_u16 v43, v43, v48\n\tv_pk_add_u16 v44, v44, v48\n\tv_pk_add_u16 v45, v45, v48\n\tv_pk_add_u16 v46, v46, v48\n\tv_pk_add_u16 v47, v47, v48\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 40) asm volatile("v_cvt_f32_u32 v40, v40\n\tv_cvt_f32_u32 v41, v41\n\tv_cvt_f32_u32 v42, v42\n\tv_cvt_f32_u32 v43, v43\n\tv_cvt_f32_u32 v44, v44\n\tv_cvt_f32_u32 v45, v45\n\tv_cvt_f32_u32 v46, v46\n\tv_cvt_f32_u32 v47, v47\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 41) asm volatile("v_subrev_u32 v40, s20, v40\n\tv_subrev_u32 v41, s20, v41\n\tv_subrev_u32 v42, s20, v42\n\tv_subrev_u32 v43, s20, v43\n\tv_subrev_u32 v44, s20, v44\n\tv_subrev_u32 v45, s20, v45\n\tv_subrev_u32 v46, s20, v46\n\tv_subrev_u32 v47, s20, v47\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 42) asm volatile("v_and_b32 v40, 0xffffffe0, v40\n\tv_and_b32 v41, 0xffffffe0, v41\n\tv_and_b32 v42, 0xffffffe0, v42\n\tv_and_b32 v43, 0xffffffe0, v43\n\tv_and_b32 v44, 0xffffffe0, v44\n\tv_and_b32 v45, 0xffffffe0, v45\n\tv_and_b32 v46, 0xffffffe0, v46\n\tv_and_b32 v47, 0xffffffe0, v47\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 43) asm volatile("v_add_u32 v40, v48, v40\n\tv_add_u32 v41, v48, v41\n\tv_add_u32 v42, v48, v42\n\tv_add_u32 v43, v48, v43\n\tv_add_u32 v44, v48, v44\n\tv_add_u32 v45, v48, v45\n\tv_add_u32 v46, v48, v46\n\tv_add_u32 v47, v48, v47\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 44) asm volatile("v_fma_f32 v40, v40, v48, v49\n\tv_fma_f32 v41, v41, v48, v49\n\tv_fma_f32 v42, v42, v48, v49\n\tv_fma_f32 v43, v43, v48, v49\n\tv_fma_f32 v44, v44, v48, v49\n\tv_fma_f32 v45, v45, v48, v49\n\tv_fma_f32 v46, v46, v48, v49\n\tv_fma_f32 v47, v47, v48, v49\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 45) asm volatile("v_pk_fma_f32 v[40:41], v[40:41], v[56:57], v[58:59]\n\tv_pk_fma_f32 v[42:43], v[42:43], v[56:57], v[58:59]\n\tv_pk_fma_f32 v[44:45], v[44:45], v[56:57], v[58:59]\n\tv_pk_fma_f32 v[46:47], v[46:47], v[56:57], v[58:59]\n\tv_pk_fma_f32 v[56:57], v[56:57], v[56:57], v[58:59]\n\tv_pk_fma_f32 v[58:59], v[58:59], v[56:57], v[58:59]\n\tv_pk_fma_f32 v[52:53], v[52:53], v[56:57], v[58:59]\n\tv_pk_fma_f32 v[54:55], v[54:55], v[56:57], v[58:59]\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 46) asm volatile("v_lshlrev_b64 v[40:41], 3, v[40:41]\n\tv_lshlrev_b64 v[42:43], 3, v[42:43]\n\tv_lshlrev_b64 v[44:45], 3, v[44:45]\n\tv_lshlrev_b64 v[46:47], 3, v[46:47]\n\tv_lshlrev_b64 v[56:57], 3, v[56:57]\n\tv_lshlrev_b64 v[58:59], 3, v[58:59]\n\tv_lshlrev_b64 v[52:53], 3, v[52:53]\n\tv_lshlrev_b64 v[54:55], 3, v[54:55]\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 47) asm volatile("v_readlane_b32 s24, v40, 3\n\tv_readlane_b32 s24, v41, 3\n\tv_readlane_b32 s24, v42, 3\n\tv_readlane_b32 s24, v43, 3\n\tv_readlane_b32 s24, v44, 3\n\tv_readlane_b32 s24, v45, 3\n\tv_readlane_b32 s24, v46, 3\n\tv_readlane_b32 s24, v47, 3\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
    if constexpr (OP == 48) asm volatile("v_mad_u64_u32 v[40:41], vcc, v40, v56, v[40:41]\n\tv_mad_u64_u32 v[42:43], vcc, v42, v56, v[42:43]\n\tv_mad_u64_u32 v[44:45], vcc, v44, v56, v[44:45]\n\tv_mad_u64_u32 v[46:47], vcc, v46, v56, v[46:47]\n\tv_mad_u64_u32 v[56:57], vcc, v56, v56, v[56:57]\n\tv_mad_u64_u32 v[58:59], vcc, v50, v56, v[58:59]\n\tv_mad_u64_u32 v[52:53], vcc, v52, v56, v[52:53]\n\tv_mad_u64_u32 v[54:55], vcc, v54, v56, v[54:55]\n\t" ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21","s22","s23","s24");
}
static const char *const kOpName[] = {"v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_sub_u32", "v_lshrrev_b32 imm", "v_ashrrev_i32 imm", "v_lshlrev_b32 imm", "v_lshlrev_b32 vgpr", "v_lshl_add_u32", "v_add_lshl_u32", "v_and_or_b32", "v_lshl_or_b32", "v_or3_b32", "v_add3_u32", "v_bfe_u32", "v_bfe_i32", "v_bfi_b32", "v_alignbit_b32", "v_perm_b32", "v_bitop3_b32", "v_cndmask_b32 vcc", "v_cndmask_b32_e64 sgpr", "v_cmp_gt_i32 vcc", "v_cmp_gt_i32_e64 sgpr", "v_mul_u32_u24", "v_mad_u32_u24", "v_mad_i32_i24", "v_mul_lo_u32", "v_mul_hi_u32", "v_min_u32", "v_mbcnt_lo_u32_b32", "v_ffbh_u32", "v_bcnt_u32_b32", "v_add_co_u32", "v_xad_u32", "v_sad_u32", "v_add_u32 dpp row_shr", "v_mov_b32 dpp row_shr", "v_pk_add_u16", "v_cvt_f32_u32", "v_subrev_u32 sgpr", "v_and_b32 literal", "v_add_u32", "v_fma_f32", "v_pk_fma_f32", "v_lshlrev_b64", "v_readlane_b32", "v_mad_u64_u32"};
constexpr int kNumOps = 49;

template <int OP>
__global__ void op_rate_kernel(int iters, unsigned long long *cycles, uint32_t *sink)
{
    uint32_t acc;
    asm volatile("v_mov_b32 v48, 3\n\tv_mov_b32 v49, 1.0\n\tv_mov_b32 v56, 3\n\tv_mov_b32 v57, 0\n\tv_mov_b32 v58, 0\n\tv_mov_b32 v59, 0\n\t"
                 "v_mov_b32 v40, 0\n\tv_mov_b32 v41, 1\n\tv_mov_b32 v42, 2\n\tv_mov_b32 v43, 3\n\tv_mov_b32 v44, 4\n\tv_mov_b32 v45, 5\n\t"
                 "v_mov_b32 v46, 6\n\tv_mov_b32 v47, 7\n\tv_mov_b32 v50, 0\n\tv_mov_b32 v51, 0\n\tv_mov_b32 v52, 0\n\tv_mov_b32 v53, 0\n\t"
                 "v_mov_b32 v54, 0\n\tv_mov_b32 v55, 0\n\ts_mov_b64 s[20:21], 0x55\n\ts_mov_b64 vcc, 0x33\n\t"
                 ::: "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc","s20","s21");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        op_block<OP>(); op_block<OP>(); op_block<OP>(); op_block<OP>(); op_block<OP>(); op_block<OP>(); op_block<OP>(); op_block<OP>();
    }
    asm volatile("v_xor_b32 %0, v40, v47" : "=v"(acc) :: "v40", "v47");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) atomicMax(cycles, t1 - t0);
    if (acc == 0x9e3779b9u) *sink = acc;
}

template <int OP>
static hipError_t launch_op(int threads, int iters, unsigned long long *cyc, uint32_t *sink)
{
    auto fn = op_rate_kernel<OP>;
    const int lds = 96 * 1024;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fn, dim3(256), dim3(threads), lds, nullptr, iters, cyc, sink);
    return hipGetLastError();
}

template <int OP>
static hipError_t dispatch_op(int op, int threads, int iters, unsigned long long *cyc, uint32_t *sink)
{
    if constexpr (OP >= kNumOps) return hipErrorInvalidValue;
    else {
        if (op == OP) return launch_op<OP>(threads, iters, cyc, sink);
        return dispatch_op<OP + 1>(op, threads, iters, cyc, sink);
    }
}

// out[0] = cycles of the slowest wave, out[1] = ms, out[2] = instructions per wave
hipError_t run_op_rate(int op, int waves_per_simd, int iters, double out[3])
{
    if (op < 0 || op >= kNumOps || waves_per_simd < 1 || waves_per_simd > 4) return hipErrorInvalidValue;
    unsigned long long *cyc = nullptr; uint32_t *sink = nullptr;
    hipEvent_t e0, e1;
    hipError_t e;
    if ((e = hipMalloc((void**)&cyc, 8)) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&sink, 4)) != hipSuccess) return e;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int pass = 0; pass < 2; ++pass) {
        hipMemset(cyc, 0, 8);
        hipEventRecord(e0, nullptr);
        e = dispatch_op<0>(op, 256 * waves_per_simd, iters, cyc, sink);
        hipEventRecord(e1, nullptr);
        if (e != hipSuccess) break;
        if ((e = hipEventSynchronize(e1)) != hipSuccess) break;
        hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long h = 0;
    if (e == hipSuccess) e = hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    out[0] = (double)h; out[1] = ms; out[2] = 64.0 * iters;
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(cyc); hipFree(sink);
    return e;
}
const char *op_rate_name(int op) { return op >= 0 && op < kNumOps ? kOpName[op] : nullptr; }

struct MixInfo { const char *name; int valu_per_iter; int lds_per_iter; };
static const MixInfo kMix[MIX_N] = {
    {"v_fma_f32", 64, 0}, {"v_add_u32", 64, 0}, {"row step (8 VALU), no LDS", 4 * 8 * 8, 0}, {"v_bcnt_u32_b32", 64, 0},
    {"v_cmp_gt_i32_e64 + v_cndmask_b32_e64", 64, 0}, {"v_mad_i32_i24", 64, 0},
    {"row step + ds_read_b64, conflict-free", 4 * 8 * 8, 4 * 8}, {"row step + ds_read_b64, random entries", 4 * 8 * 8, 4 * 8},
    {"v_lshlrev_b32", 64, 0}, {"ds_read_b64 + its 2 address VALU, conflict-free", 4 * 16, 4 * 8}, {"ds_read_b64 + its 2 address VALU, random entries", 4 * 16, 4 * 8},
    {"row step, tails of 2 lookups interleaved, random entries", 4 * 8 * 8, 4 * 8}, {"row step, tails of 4 lookups interleaved, random entries", 4 * 8 * 8, 4 * 8},
    {"row step, tails of 8 lookups interleaved, random entries", 4 * 8 * 8, 4 * 8},
    {"row step + ds_read_b64, 64 consecutive ranks per wave and lookup (long runs)", 4 * 8 * 8, 4 * 8},
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
        X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14)
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

// ---- C entry points of libbgt_hip_bench.so (include/bgt_hip_bench.h).  The calibration kernels are measurement
// tools, not product: they live in a library of their own that bench.py and scripts/valu_calibration.py load.
extern "C" int bgth_debug_issue_rate(int device, int mix, int waves_per_simd, int iters, double out[4])
{
    if (hipSetDevice(device) != hipSuccess) return -1;
    return bgth::run_issue_rate(mix, waves_per_simd, iters, out) == hipSuccess ? 0 : -1;
}
extern "C" const char *bgth_debug_issue_rate_name(int mix) { return bgth::issue_rate_mix_name(mix); }
extern "C" int bgth_debug_op_rate(int device, int op, int waves_per_simd, int iters, double out[3])
{
    if (hipSetDevice(device) != hipSuccess) return -1;
    return bgth::run_op_rate(op, waves_per_simd, iters, out) == hipSuccess ? 0 : -1;
}
extern "C" const char *bgth_debug_op_rate_name(int op) { return bgth::op_rate_name(op); }
