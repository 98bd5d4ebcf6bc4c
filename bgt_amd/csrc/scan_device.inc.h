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
//   grid   = (sub-block: the rows between two rank-form checkpoints, 2048 by default) x (column slice); all slices
//            of a sub-block are placed on one XCD (workgroup id mod 8) so the RLE bytes they share come from one L2.
//   wave   = owns CPT consecutive 64-slot chunks; a chunk never mixes sample groups, so the allele
//            counts of a chunk are popcounts of the two 64-bit ballots (the v_cmp that selects the
//            new rank already is the ballot).
//   batch  = K rows: phase A builds the 2K bit-vectors (byte -> run length, wave prefix sum -> run starts,
//            xor-toggle at every change of bit, prefix-xor -> bits, popcount prefix sum -> rank directory),
//            phase B walks the K rows with no barrier.  Narrow cohorts: one wave per plane-row, two batch buffers,
//            batches pipelined.  Wide cohorts (TEAM): a team of waves per plane-row, driven by the row index.
//
// No MFMA: this is integer/bit work bound by LDS issue and VALU, not by HBM.
#pragma once
#include "scan_kernels.h"
#include <stdlib.h>

namespace bgth {

// ----------------------------------------------------------------------------------------------------
// wave-level helpers (64 lanes)
// ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lanes_below(uint64_t mask)
{   // number of set bits of mask in lanes below the caller
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// inclusive prefix sum over the 64 lanes, all in the VALU (DPP row shifts + the two row broadcasts)
__device__ __forceinline__ uint32_t wave_incl_add(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
    return v;
}

// value of the lane below (lane 0 receives `first`)
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v, uint32_t first)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138, 0xf, 0xf, false);   // wave_shr:1
}

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() also drains vmcnt (its
// workgroup-scope release covers global memory), which would turn every software-prefetched global load
// into a stall at the next barrier; LDS traffic only needs lgkmcnt(0).
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// s_setprio takes an immediate
// ... and a priority known only at run time is a branch: as a two-level tree, one or two taken branches (hipcc's switch: up to four
// branch instructions per call; s_setreg on STATUS.USER_PRIO is ignored -- scripts/setreg_prio_probe.hip)
__device__ __forceinline__ void set_wave_priority_uniform(uint32_t p)
{
    asm volatile("s_bitcmp1_b32 %0, 1\n\ts_cbranch_scc1 2f\n\ts_bitcmp1_b32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_setprio 0\n\ts_branch 9f\n"
                 "1:\n\ts_setprio 1\n\ts_branch 9f\n"
                 "2:\n\ts_bitcmp1_b32 %0, 0\n\ts_cbranch_scc1 3f\n\ts_setprio 2\n\ts_branch 9f\n"
                 "3:\n\ts_setprio 3\n"
                 "9:" :: "s"(p) : "scc");
}
__device__ __forceinline__ void set_wave_priority(int p)
{
    switch (p) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
    }
}

__device__ __forceinline__ uint32_t lane63(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }

__device__ __forceinline__ uint32_t rle_len(uint32_t byte)
{   // reference pbwt.c:12-21 as arithmetic: code = byte>>1, len = (code&15) << 4*(code>>4)
    uint32_t code = byte >> 1;
    return (code & 15u) << ((code >> 4) << 2);
}

// ----------------------------------------------------------------------------------------------------
// the scan kernel
// ----------------------------------------------------------------------------------------------------
// ----------------------------------------------------------------------------------------------------
// The row step.  For one tracked column and one plane:   e = B[r>>5] = {32 row bits, ones before them}
//     bit = e.bits[r&31] ;  rank1(r) = e.before + popc(e.bits & low(r&31))
//     r   = bit ? n0 + rank1(r) : r - rank1(r)                      (LF-mapping)
// The registers hold the COMPLEMENT q = ~r, which saves an instruction: the shift amount q&31 = 31-(r&31)
// puts bit r&31 of the word into the sign position and drops the bits above it in one v_lshlrev, so
//     t   = e.bits << (q & 31)          bit = t < 0          oi = e.before + popc(t)   (ones up to and incl. r)
//     q   = bit ? (-n0) - oi : q + oi
// and the word address base + 8 (r>>5) = (base - 8) - 8 (q >>arith 5) is one v_mad_i32_i24.
// Eight VALU instructions and one ds_read_b64 per lookup; the v_cmp that steers the select is also the
// wave ballot of the decoded bit.  Hand-scheduled: left to hipcc the unrolled row body keeps one
// 64-bit SGPR condition per lookup alive to the end of the row and spills (measured: 151 VGPRs and
// 333 v_writelane/v_readlane at 16 columns per thread; this form needs 2 VGPRs per column + 16).
// Scratch registers are named (v64..v79: a kernel of few columns per thread then needs 80 VGPRs and six of its waves
// share a SIMD; the columns of wider ones are allocated around them) and declared as clobbers; every ds_read is waited for
// inside the statement; a lookup needs two of them: the low one is first the LDS address, then the shifted
// word, then the candidate for bit = 1.  BASE operands are (LDS address of the plane-row) - 8, N0 operands are -n0.
// ----------------------------------------------------------------------------------------------------
#define BGTH_TAIL(Q, ELO, EHI, T, MASK, N0)            \
    "v_lshlrev_b32 " ELO ", " Q ", " ELO "\n\t"        \
    "v_bcnt_u32_b32 " EHI ", " ELO ", " EHI "\n\t"     \
    "v_cmp_gt_i32_e64 " MASK ", 0, " ELO "\n\t"        \
    "v_sub_u32 " T ", " N0 ", " EHI "\n\t"             \
    "v_add_u32 " EHI ", " Q ", " EHI "\n\t"            \
    "v_cndmask_b32_e64 " Q ", " EHI ", " T ", " MASK "\n\t"
#define BGTH_ADDR(T, Q, BASE)                          \
    "v_ashrrev_i32 " T ", 5, " Q "\n\t"                \
    "v_mad_i32_i24 " T ", " T ", -8, " BASE "\n\t"

// Two lookups with their dependent chains interleaved: a wave that shares its SIMD with only one other (the wide-cohort
// kernels: 2 waves per SIMD) issues 8 % more lookups per cycle this way, four waves per SIMD are indifferent
// (profiles/r02a_calibration: 4.56 vs 4.94 cycles per instruction at 2 waves, 3.99 vs 4.02 at 4).
#define BGTH_TAIL2(QA, ELA, EHA, MA, N0A, QB, ELB, EHB, MB, N0B)  \
    "v_lshlrev_b32 " ELA ", " QA ", " ELA "\n\t"                  \
    "v_lshlrev_b32 " ELB ", " QB ", " ELB "\n\t"                  \
    "v_bcnt_u32_b32 " EHA ", " ELA ", " EHA "\n\t"                \
    "v_bcnt_u32_b32 " EHB ", " ELB ", " EHB "\n\t"                \
    "v_cmp_gt_i32_e64 " MA ", 0, " ELA "\n\t"                     \
    "v_cmp_gt_i32_e64 " MB ", 0, " ELB "\n\t"                     \
    "v_sub_u32 " ELA ", " N0A ", " EHA "\n\t"                     \
    "v_sub_u32 " ELB ", " N0B ", " EHB "\n\t"                     \
    "v_add_u32 " EHA ", " QA ", " EHA "\n\t"                      \
    "v_add_u32 " EHB ", " QB ", " EHB "\n\t"                      \
    "v_cndmask_b32_e64 " QA ", " EHA ", " ELA ", " MA "\n\t"      \
    "v_cndmask_b32_e64 " QB ", " EHB ", " ELB ", " MB "\n\t"
#define BGTH_ASHR(T, Q)        "v_ashrrev_i32 " T ", 5, " Q "\n\t"
#define BGTH_MAD(T, BASE)      "v_mad_i32_i24 " T ", " T ", -8, " BASE "\n\t"

// per column, on the scalar unit (keeps the VALU for the lookups): ones of plane 0, ones of plane 1,
// ones in both.  M0/M1 are the ballots the v_cmp of the two lookups produced.
#define BGTH_COUNT(M0, M1, CA, CB, CC)                 \
    "s_bcnt1_i32_b64 vcc_lo, " M0 "\n\t"               \
    "s_add_u32 " CA ", " CA ", vcc_lo\n\t"             \
    "s_bcnt1_i32_b64 vcc_lo, " M1 "\n\t"               \
    "s_add_u32 " CB ", " CB ", vcc_lo\n\t"             \
    "s_and_b64 vcc, " M0 ", " M1 "\n\t"                \
    "s_bcnt1_i32_b64 vcc_lo, vcc\n\t"                  \
    "s_add_u32 " CC ", " CC ", vcc_lo\n\t"
// Whole cohort, one group, counts only (WC): the ones of a plane over ALL columns are the row's own (m - n0, known from its
// string), so only the columns with a one in BOTH planes are counted: n(code 1) = ones0 - n(code 3), n(code 2) = ones1 - n(code 3).
// Three scalar instructions per column instead of seven (round 5: C2 -1.6 %, one C4 shard -1.5 %).
#define BGTH_COUNT3(M0, M1, CA, CB, CC)                \
    "s_and_b64 vcc, " M0 ", " M1 "\n\t"                \
    "s_bcnt1_i32_b64 vcc_lo, vcc\n\t"                  \
    "s_add_u32 " CC ", " CC ", vcc_lo\n\t"

// When plane 1 of the row is all zero (no missing call, no <M>: most sites of a fully called panel) its ranks do not
// move (reference pbwt.c:135-138) and its lookups are skipped.  The choice is a SCALAR branch inside the
// statement (operand Z1 != 0), not a second statement: two statements updating the same ranks on two paths made
// hipcc reconcile the register assignments with ~40 copies per row on the common path.
#define BGTH_COUNT1(M0, CA)                            \
    "s_bcnt1_i32_b64 vcc_lo, " M0 "\n\t"               \
    "s_add_u32 " CA ", " CA ", vcc_lo\n\t"

#define BGTH_STEP2_BODY(CNT)                                                             \
        BGTH_ASHR("v64", "%0") BGTH_ASHR("v66", "%1") BGTH_ASHR("v68", "%2") BGTH_ASHR("v70", "%3")  \
        BGTH_MAD("v64", "%11") BGTH_MAD("v66", "%12") BGTH_MAD("v68", "%11") BGTH_MAD("v70", "%12")  \
        "ds_read_b64 v[64:65], v64\n\t"                                               \
        "ds_read_b64 v[66:67], v66\n\t"                                               \
        "ds_read_b64 v[68:69], v68\n\t"                                               \
        "ds_read_b64 v[70:71], v70\n\t"                                               \
        "s_waitcnt lgkmcnt(2)\n\t"                                                       \
        BGTH_TAIL2("%0", "v64", "v65", "%4", "%13", "%1", "v66", "v67", "%5", "%14")  \
        CNT("%4", "%5", "%8", "%9", "%10")                                               \
        "s_waitcnt lgkmcnt(0)\n\t"                                                       \
        BGTH_TAIL2("%2", "v68", "v69", "%6", "%13", "%3", "v70", "v71", "%7", "%14")  \
        CNT("%6", "%7", "%8", "%9", "%10")
#define BGTH_STEP2_BOTH BGTH_STEP2_BODY(BGTH_COUNT)
#define BGTH_STEP2_PLANE0                                                                \
        BGTH_ADDR("v64", "%0", "%11") BGTH_ADDR("v68", "%2", "%11")                    \
        "ds_read_b64 v[64:65], v64\n\t"                                               \
        "ds_read_b64 v[68:69], v68\n\t"                                               \
        "s_mov_b64 %5, 0\n\t"                                                            \
        "s_mov_b64 %7, 0\n\t"                                                            \
        "s_waitcnt lgkmcnt(1)\n\t"                                                       \
        BGTH_TAIL("%0", "v64", "v65", "v64", "%4", "%13")                             \
        BGTH_COUNT1("%4", "%8")                                                          \
        "s_waitcnt lgkmcnt(0)\n\t"                                                       \
        BGTH_TAIL("%2", "v68", "v69", "v68", "%6", "%13")                             \
        BGTH_COUNT1("%6", "%8")
#define BGTH_STEP2_OPERANDS                                                                                          \
        : "+v"(ra0), "+v"(ra1), "+v"(rb0), "+v"(rb1), "=&s"(ma0), "=&s"(ma1), "=&s"(mb0), "=&s"(mb1),               \
          "+s"(ca), "+s"(cb), "+s"(cc)                                                                               \
        : "s"(base0), "s"(base1), "s"(n00), "s"(n01)                                                                 \
        : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "vcc", "scc", "memory"

// two columns x two planes: 4 LDS reads in flight.  ZP: with the all-zero-plane-1 shortcut, requested by passing
// base1 = 0 (never a real operand value: a plane-1 row does not start at LDS address 8)
template <bool ZP, bool WC = false>
__device__ __forceinline__ void step2(uint32_t &ra0, uint32_t &ra1, uint32_t &rb0, uint32_t &rb1,
                                      uint64_t &ma0, uint64_t &ma1, uint64_t &mb0, uint64_t &mb1,
                                      uint32_t &ca, uint32_t &cb, uint32_t &cc,
                                      uint32_t base0, uint32_t base1, uint32_t n00, uint32_t n01)
{
    if constexpr (ZP) {
        asm volatile(
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_cmp_eq_u32 %12, 0\n\t"                         /* base1 == 0: this row's plane 1 is empty */
            "s_cbranch_scc1 .Lbgth_z_%=\n\t"
            BGTH_STEP2_BOTH
            "s_branch .Lbgth_e_%=\n"
            ".Lbgth_z_%=:\n\t"
            BGTH_STEP2_PLANE0
            ".Lbgth_e_%=:\n\t"
            BGTH_STEP2_OPERANDS);
    } else if constexpr (WC) {
        asm volatile(
            "s_waitcnt lgkmcnt(0)\n\t"
            BGTH_STEP2_BODY(BGTH_COUNT3)
            BGTH_STEP2_OPERANDS);
    } else {
        asm volatile(
            "s_waitcnt lgkmcnt(0)\n\t"
            BGTH_STEP2_BOTH
            BGTH_STEP2_OPERANDS);
    }
}

#define BGTH_STEP4_BODY(CNT)                                                             \
        BGTH_ASHR("v64", "%0") BGTH_ASHR("v66", "%1") BGTH_ASHR("v68", "%2") BGTH_ASHR("v70", "%3")  \
        BGTH_ASHR("v72", "%4") BGTH_ASHR("v74", "%5") BGTH_ASHR("v76", "%6") BGTH_ASHR("v78", "%7")  \
        BGTH_MAD("v64", "%19") BGTH_MAD("v66", "%20") BGTH_MAD("v68", "%19") BGTH_MAD("v70", "%20")  \
        BGTH_MAD("v72", "%19") BGTH_MAD("v74", "%20") BGTH_MAD("v76", "%19") BGTH_MAD("v78", "%20")  \
        "ds_read_b64 v[64:65], v64\n\t"                                               \
        "ds_read_b64 v[66:67], v66\n\t"                                               \
        "ds_read_b64 v[68:69], v68\n\t"                                               \
        "ds_read_b64 v[70:71], v70\n\t"                                               \
        "ds_read_b64 v[72:73], v72\n\t"                                               \
        "ds_read_b64 v[74:75], v74\n\t"                                               \
        "ds_read_b64 v[76:77], v76\n\t"                                               \
        "ds_read_b64 v[78:79], v78\n\t"                                               \
        "s_waitcnt lgkmcnt(6)\n\t"                                                       \
        BGTH_TAIL2("%0", "v64", "v65", "%8", "%21", "%1", "v66", "v67", "%9", "%22")  \
        CNT("%8", "%9", "%16", "%17", "%18")                                             \
        "s_waitcnt lgkmcnt(4)\n\t"                                                       \
        BGTH_TAIL2("%2", "v68", "v69", "%10", "%21", "%3", "v70", "v71", "%11", "%22") \
        CNT("%10", "%11", "%16", "%17", "%18")                                           \
        "s_waitcnt lgkmcnt(2)\n\t"                                                       \
        BGTH_TAIL2("%4", "v72", "v73", "%12", "%21", "%5", "v74", "v75", "%13", "%22") \
        CNT("%12", "%13", "%16", "%17", "%18")                                           \
        "s_waitcnt lgkmcnt(0)\n\t"                                                       \
        BGTH_TAIL2("%6", "v76", "v77", "%14", "%21", "%7", "v78", "v79", "%15", "%22") \
        CNT("%14", "%15", "%16", "%17", "%18")
#define BGTH_STEP4_BOTH BGTH_STEP4_BODY(BGTH_COUNT)
#define BGTH_STEP4_PLANE0                                                                \
        BGTH_ADDR("v64", "%0", "%19") BGTH_ADDR("v68", "%2", "%19")                    \
        BGTH_ADDR("v72", "%4", "%19") BGTH_ADDR("v76", "%6", "%19")                    \
        "ds_read_b64 v[64:65], v64\n\t"                                               \
        "ds_read_b64 v[68:69], v68\n\t"                                               \
        "ds_read_b64 v[72:73], v72\n\t"                                               \
        "ds_read_b64 v[76:77], v76\n\t"                                               \
        "s_mov_b64 %9, 0\n\t"                                                            \
        "s_mov_b64 %11, 0\n\t"                                                           \
        "s_mov_b64 %13, 0\n\t"                                                           \
        "s_mov_b64 %15, 0\n\t"                                                           \
        "s_waitcnt lgkmcnt(3)\n\t"                                                       \
        BGTH_TAIL("%0", "v64", "v65", "v64", "%8", "%21")                             \
        BGTH_COUNT1("%8", "%16")                                                         \
        "s_waitcnt lgkmcnt(2)\n\t"                                                       \
        BGTH_TAIL("%2", "v68", "v69", "v68", "%10", "%21")                            \
        BGTH_COUNT1("%10", "%16")                                                        \
        "s_waitcnt lgkmcnt(1)\n\t"                                                       \
        BGTH_TAIL("%4", "v72", "v73", "v72", "%12", "%21")                            \
        BGTH_COUNT1("%12", "%16")                                                        \
        "s_waitcnt lgkmcnt(0)\n\t"                                                       \
        BGTH_TAIL("%6", "v76", "v77", "v76", "%14", "%21")                            \
        BGTH_COUNT1("%14", "%16")
#define BGTH_STEP4_OPERANDS                                                                                                  \
        : "+v"(r0[0]), "+v"(r1[0]), "+v"(r0[1]), "+v"(r1[1]), "+v"(r0[2]), "+v"(r1[2]), "+v"(r0[3]), "+v"(r1[3]),            \
          "=&s"(m0[0]), "=&s"(m1[0]), "=&s"(m0[1]), "=&s"(m1[1]), "=&s"(m0[2]), "=&s"(m1[2]), "=&s"(m0[3]), "=&s"(m1[3]),    \
          "+s"(ca), "+s"(cb), "+s"(cc)                                                                                       \
        : "s"(base0), "s"(base1), "s"(n00), "s"(n01)                                                                         \
        : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75",                    \
          "v76", "v77", "v78", "v79", "vcc", "scc", "memory"

// four columns x two planes: 8 LDS reads in flight
template <bool ZP, bool WC = false>
__device__ __forceinline__ void step4(uint32_t (&r0)[4], uint32_t (&r1)[4], uint64_t (&m0)[4], uint64_t (&m1)[4],
                                      uint32_t &ca, uint32_t &cb, uint32_t &cc,
                                      uint32_t base0, uint32_t base1, uint32_t n00, uint32_t n01)
{
    if constexpr (ZP) {
        asm volatile(
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_cmp_eq_u32 %20, 0\n\t"                         /* base1 == 0: this row's plane 1 is empty */
            "s_cbranch_scc1 .Lbgth_z_%=\n\t"
            BGTH_STEP4_BOTH
            "s_branch .Lbgth_e_%=\n"
            ".Lbgth_z_%=:\n\t"
            BGTH_STEP4_PLANE0
            ".Lbgth_e_%=:\n\t"
            BGTH_STEP4_OPERANDS);
    } else if constexpr (WC) {
        asm volatile(
            "s_waitcnt lgkmcnt(0)\n\t"
            BGTH_STEP4_BODY(BGTH_COUNT3)
            BGTH_STEP4_OPERANDS);
    } else {
        asm volatile(
            "s_waitcnt lgkmcnt(0)\n\t"
            BGTH_STEP4_BOTH
            BGTH_STEP4_OPERANDS);
    }
}

// ---- round 6 A/B: TWO RANKS PER REGISTER where m <= 65,535.  A column's plane-0 rank sits in the low half of
// its register, its plane-1 rank in the high half, both complemented as above (16-bit complements).  What can be shared by the two
// lookups of a column is the rank update: one v_lshl_or packs the two "ones up to r", one v_pk_sub_u16 / v_pk_add_u16 make both
// candidates of both planes, and the select is done per half by v_cndmask_b32_sdwa (dst_unused:UNUSED_PRESERVE) straight from
// vcc -- 5 instructions where the plain form has 6; the address (v_bfe_u32 / v_lshrrev + v_mad), the shift (the high half's shift
// amount through SDWA src0_sel:WORD_1), v_bcnt and v_cmp stay per lookup: 15 VALU instructions per column against 16.
//   BASE operands here = (LDS address of the plane-row) + 8 * 2047: the unsigned 11-bit field (q >> 5) = 2047 - (r >> 5).
//   N0PK = (-n0 of plane 1) << 16 | (-n0 of plane 0) & 0xffff.     Whole-cohort counting only (n(code 3): BGTH_COUNT3's sums).
#define BGTH_PK_COL(Q, E0L, E0H, E0P, E1L, E1H, E1P, MSAVE)                                             \
    "v_lshlrev_b32 " E0L ", " Q ", " E0L "\n\t"                                                          \
    "v_lshlrev_b32_sdwa " E1L ", " Q ", " E1L " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n\t" \
    "v_bcnt_u32_b32 " E0H ", " E0L ", " E0H "\n\t"                                                       \
    "v_bcnt_u32_b32 " E1H ", " E1L ", " E1H "\n\t"                                                       \
    "v_lshl_or_b32 " E0H ", " E1H ", 16, " E0H "\n\t"                                                    \
    "v_pk_sub_u16 " E1H ", %8, " E0H "\n\t"                                                             \
    "v_pk_add_u16 " E0H ", " Q ", " E0H "\n\t"                                                           \
    "v_cmp_gt_i32 vcc, 0, " E0L "\n\t"                                                                   \
    "v_cndmask_b32_sdwa " Q ", " E0H ", " E1H ", vcc dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0\n\t" \
    "s_mov_b64 " MSAVE ", vcc\n\t"                                                                       \
    "v_cmp_gt_i32 vcc, 0, " E1L "\n\t"                                                                   \
    "v_cndmask_b32_sdwa " Q ", " E0H ", " E1H ", vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n\t" \
    "s_and_b64 vcc, " MSAVE ", vcc\n\t"                                                                  \
    "s_bcnt1_i32_b64 vcc_lo, vcc\n\t"                                                                    \
    "s_add_u32 %5, %5, vcc_lo\n\t"
#define BGTH_PK_ADDR(Q, T0, T1)                                                                        \
    "v_bfe_u32 " T0 ", " Q ", 5, 11\n\t"                                                                 \
    "v_lshrrev_b32 " T1 ", 21, " Q "\n\t"                                                                \
    "v_mad_i32_i24 " T0 ", " T0 ", -8, %6\n\t"                                                          \
    "v_mad_i32_i24 " T1 ", " T1 ", -8, %7\n\t"
// four columns x two planes in four registers (operands: %0-%3 the packed ranks, %4 a scratch SGPR pair -- the plane-0 ballot while a
// column's plane 1 is selected --, %5 the n(code 3) sum, %6 / %7 the plane rows' bases, %8 N0PK)
__device__ __forceinline__ void step4pk(uint32_t (&rp)[4], uint32_t &cc, uint32_t base0, uint32_t base1, uint32_t n0pk)
{
    uint64_t ms;
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n\t"
        BGTH_PK_ADDR("%0", "v64", "v66") BGTH_PK_ADDR("%1", "v68", "v70") BGTH_PK_ADDR("%2", "v72", "v74") BGTH_PK_ADDR("%3", "v76", "v78")
        "ds_read_b64 v[64:65], v64\n\t"
        "ds_read_b64 v[66:67], v66\n\t"
        "ds_read_b64 v[68:69], v68\n\t"
        "ds_read_b64 v[70:71], v70\n\t"
        "ds_read_b64 v[72:73], v72\n\t"
        "ds_read_b64 v[74:75], v74\n\t"
        "ds_read_b64 v[76:77], v76\n\t"
        "ds_read_b64 v[78:79], v78\n\t"
        "s_waitcnt lgkmcnt(6)\n\t"
        BGTH_PK_COL("%0", "v64", "v65", "v[64:65]", "v66", "v67", "v[66:67]", "%4")
        "s_waitcnt lgkmcnt(4)\n\t"
        BGTH_PK_COL("%1", "v68", "v69", "v[68:69]", "v70", "v71", "v[70:71]", "%4")
        "s_waitcnt lgkmcnt(2)\n\t"
        BGTH_PK_COL("%2", "v72", "v73", "v[72:73]", "v74", "v75", "v[74:75]", "%4")
        "s_waitcnt lgkmcnt(0)\n\t"
        BGTH_PK_COL("%3", "v76", "v77", "v[76:77]", "v78", "v79", "v[78:79]", "%4")
        : "+v"(rp[0]), "+v"(rp[1]), "+v"(rp[2]), "+v"(rp[3]), "=&s"(ms), "+s"(cc)
        : "s"(base0), "s"(base1), "s"(n0pk)
        : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75",
          "v76", "v77", "v78", "v79", "vcc", "scc", "memory");
}

// (No counts: its only user, the plane-split kernel, takes them from the bit planes -- until round 5 every column still paid an
// s_bcnt1 + s_add into a sum nobody read.)
// One plane only, eight scratch registers just below the VGPR budget: the statement of the plane-split kernels at six waves
// per SIMD (v72..v79, at most 80 VGPRs: three workgroups of 512 threads or two of 768 per CU).  The two-plane statements
// above name sixteen, v64..v79 (until late in round 4: v104..v119, which pinned every kernel at 120 VGPRs whatever it needed).
#define BGTH_DEFINE_STEP4_PLANE(NAME, A, A1, AP, B, B1, BP, C, C1, CP, D, D1, DP)                                                      \
__device__ __forceinline__ void NAME(uint32_t (&r0)[4], uint64_t (&m0)[4], uint32_t base0, uint32_t n00)               \
{                                                                                                                       \
    asm volatile(                                                                                                       \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                      \
        BGTH_ADDR(A, "%0", "%8") BGTH_ADDR(B, "%1", "%8")                                                               \
        BGTH_ADDR(C, "%2", "%8") BGTH_ADDR(D, "%3", "%8")                                                               \
        "ds_read_b64 " AP ", " A "\n\t"                                                                                 \
        "ds_read_b64 " BP ", " B "\n\t"                                                                                 \
        "ds_read_b64 " CP ", " C "\n\t"                                                                                 \
        "ds_read_b64 " DP ", " D "\n\t"                                                                                 \
        "s_waitcnt lgkmcnt(3)\n\t"                                                                                      \
        BGTH_TAIL("%0", A, A1, A, "%4", "%9")                                                                          \
        "s_waitcnt lgkmcnt(2)\n\t"                                                                                      \
        BGTH_TAIL("%1", B, B1, B, "%5", "%9")                                                                          \
        "s_waitcnt lgkmcnt(1)\n\t"                                                                                      \
        BGTH_TAIL("%2", C, C1, C, "%6", "%9")                                                                          \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                      \
        BGTH_TAIL("%3", D, D1, D, "%7", "%9")                                                                          \
        : "+v"(r0[0]), "+v"(r0[1]), "+v"(r0[2]), "+v"(r0[3]), "=&s"(m0[0]), "=&s"(m0[1]), "=&s"(m0[2]), "=&s"(m0[3])            \
        : "s"(base0), "s"(n00)                                                                                          \
        : A, A1, B, B1, C, C1, D, D1, "vcc", "scc", "memory");                                                          \
}
BGTH_DEFINE_STEP4_PLANE(step4_plane_low, "v72", "v73", "v[72:73]", "v74", "v75", "v[74:75]", "v76", "v77", "v[76:77]", "v78", "v79", "v[78:79]")

// ----------------------------------------------------------------------------------------------------
// Phase A for one plane-row, executed by ONE wave: RLE string -> bit-vector + rank directory in LDS.
//   1. clear the row's entries
//   2. every lane decodes 4 code bytes (strings are packed 4-byte aligned), a wave prefix sum of the
//      run lengths gives every run's start, and wherever the bit differs from the previous byte's bit
//      the row "toggles" at the run start: the lane xors the mask ~0 << (start & 31) into the word of
//      the start (LDS atomic; bytes of zero length cancel out).  After all toggles a word holds the
//      prefix parity of its own toggles, and its bit 31 their total parity.
//   3. the row bits are that word, inverted when the parity of the toggles in all earlier words is
//      odd (ballot of bit 31 + mbcnt); a wave prefix sum of the word popcounts is the rank
//      directory.  The number of ones falls out as the final carry.
// LDS operations of one wave complete in order, so no workgroup barrier is needed between the steps;
// the wavefront fences only pin the compiler's ordering.
// ----------------------------------------------------------------------------------------------------
// step 2: RLE string -> toggles (one wave walks the whole string)
__device__ __forceinline__ void rle_toggles(const ScanArgs &a, const uint8_t *__restrict__ rle, uint2 *bd, uint64_t desc,
                                            uint32_t pre0, uint32_t pre1, uint32_t pre2, uint32_t pre3, int npre, int lane)
{
    const int m = a.m;
    const uint32_t *q4 = reinterpret_cast<const uint32_t*>(rle + (desc & kDescOffMask));
    const uint32_t len = (uint32_t)(desc >> kDescLenShift);
    uint32_t pos = 0, prevbit = 0;
    bool stop = false;
    for (uint32_t base = 0; base < len && !stop; base += 256) {
        const uint32_t k0 = base + 4u * (uint32_t)lane;
        // the first 256*npre bytes were fetched one batch ahead; longer strings read on
        uint32_t w;
        if (base == 0 && npre > 0) w = pre0;
        else if (base == 256 && npre > 0) w = pre1;
        else if (base == 512 && npre > 2) w = pre2;
        else if (base == 768 && npre > 2) w = pre3;
        else w = k0 < len ? q4[(base >> 2) + lane] : 0u;
        uint32_t byte[4], l[4];
        bool valid[4];
        bool anyz = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            byte[i] = (w >> (8 * i)) & 255u;
            valid[i] = k0 + i < len;
            anyz = anyz || (valid[i] && byte[i] == 0u);
        }
        const uint64_t z = __ballot(anyz);                  // a zero byte ends the row (ref pbwt.c:73)
        if (z) {
            const int first = __ffsll((unsigned long long)z) - 1;
            bool dead = lane > first;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (lane == first && byte[i] == 0u) dead = true;
                valid[i] = valid[i] && !dead;
            }
            stop = true;
        }
        uint32_t run = 0, before[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { l[i] = valid[i] ? rle_len(byte[i]) : 0u; before[i] = run; run += l[i]; }
        const uint32_t incl = wave_incl_add(run);
        const uint32_t lane_start = pos + incl - run;
        uint32_t pb = wave_shr1(byte[3] & 1u, prevbit);      // bit of the byte before this lane's first
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t b = byte[i] & 1u, start = lane_start + before[i];
            if (valid[i] && b != pb && start < (uint32_t)m)
                atomicXor(&bd[start >> 5].x, 0xffffffffu << (start & 31));
            pb = b;
        }
        prevbit = lane63(byte[3] & 1u);
        pos += lane63(incl);
    }
}

// step 3 over the words [w0,w1) of a row (one wave; w0 a multiple of 4).  A lane owns 4 consecutive words
// per trip (256 words per trip): one ballot/mbcnt for the toggle parity entering the lane and one DPP
// prefix sum of the lane's ones serve four words, the chain across the four is local arithmetic.
__device__ __forceinline__ void directory_pass(uint2 *bd, int w0, int w1, int nw, uint32_t tail_mask,
                                               uint32_t &carry_x, uint32_t &carry_c, int lane)
{
    for (int base = w0; base < w1; base += 256) {
        const int i0 = base + 4 * lane;
        const bool full = base + 256 <= w1 && base + 256 < nw;           // wave-uniform: no partial lane, no tail word
        uint32_t t[4] = {0u, 0u, 0u, 0u};
        if (full || i0 + 3 < w1) {
            const uint4 *src = reinterpret_cast<const uint4*>(bd + i0);  // rows are 16-byte aligned
            const uint4 lo = src[0], hi = src[1];
            t[0] = lo.x; t[1] = lo.z; t[2] = hi.x; t[3] = hi.z;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) if (i0 + k < w1) t[k] = bd[i0 + k].x;
        }
        // bit 31 of a word = parity of its toggles; parity of the lane's four words = sign of their xor
        const uint64_t par = __ballot((int32_t)(t[0] ^ t[1] ^ t[2] ^ t[3]) < 0);
        // cm = all ones when the parity of all toggles before the word is odd (the word is then inverted)
        uint32_t cm = 0u - ((lanes_below(par) ^ carry_x) & 1u);
        uint32_t v[4], pre[4], ones = 0;
        if (full) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[k] = t[k] ^ cm; pre[k] = ones; ones += (uint32_t)__popc(v[k]);
                cm ^= (uint32_t)((int32_t)t[k] >> 31);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t x = t[k] ^ cm;
                if (i0 + k == nw - 1) x &= tail_mask;
                if (i0 + k >= w1) x = 0u;
                v[k] = x; pre[k] = ones; ones += (uint32_t)__popc(x);
                cm ^= (uint32_t)((int32_t)t[k] >> 31);
            }
        }
        const uint32_t incl = wave_incl_add(ones);
        const uint32_t b = carry_c + incl - ones;
        if (full || i0 + 3 < w1) {
            uint4 *dst = reinterpret_cast<uint4*>(bd + i0);
            dst[0] = make_uint4(v[0], b, v[1], b + pre[1]);
            dst[1] = make_uint4(v[2], b + pre[2], v[3], b + pre[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) if (i0 + k < w1) bd[i0 + k] = make_uint2(v[k], b + pre[k]);
        }
        carry_x ^= (uint32_t)__popcll(par) & 1u;
        carry_c += lane63(incl);
    }
}

// Directory trips of a team wave when the toggles live in their own array `tog` (one uint32 per row word,
// rows padded to a multiple of 4 words with words that are never toggled): trip t covers words
// [256 t, 256 t + 256), its carries come from the row index, so trips are independent -- two are processed
// at a time to overlap their LDS latencies and scan chains.  The toggle words are cleared as they are read,
// which leaves the array ready for the next row.
struct TripCarry { uint32_t cx, cnt; };

// (NP = trips in flight: 2, or 1 where the registers are needed for columns; STREAM: the entries go to HBM and are not read
// again by this kernel -- the producer of scan_dir.hip -- so they are stored non-temporally)
typedef uint32_t bgth_u32x4 __attribute__((ext_vector_type(4)));
template <bool STREAM>
__device__ __forceinline__ void store_entries2(uint4 *dst, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    if constexpr (STREAM) { bgth_u32x4 v = {a, b, c, d}; __builtin_nontemporal_store(v, reinterpret_cast<bgth_u32x4*>(dst)); }
    else *dst = make_uint4(a, b, c, d);
}
template <int NP, bool STREAM = false>
__device__ __forceinline__ void directory_trips_tog(uint32_t *tog, uint2 *bd, int tw, int wpp, int ntrip, int nw,
                                                    uint32_t tail_mask, uint32_t cyl, int lane)
{
    int u = 0;
    // A trip that holds neither the row's last word nor anything behind it (all but the last one or two of a row) needs no
    // bounds: its toggle words -> entries without a predicate, and a PAIR of such trips shares one prefix scan, their ones
    // (<= 128 per lane, <= 8192 per trip) in the two halves of a register.  ~35 instead of ~54 VALU instructions per trip.
    auto lean = [&](const uint4 q, uint32_t cy, uint32_t (&v)[4], uint32_t (&pre)[4]) -> uint32_t {
        const uint64_t par = __ballot((int32_t)(q.x ^ q.y ^ q.z ^ q.w) < 0);
        uint32_t cm = 0u - ((lanes_below(par) ^ (cy >> 31)) & 1u);
        v[0] = q.x ^ cm; cm ^= (uint32_t)((int32_t)q.x >> 31);
        v[1] = q.y ^ cm; cm ^= (uint32_t)((int32_t)q.y >> 31);
        v[2] = q.z ^ cm; cm ^= (uint32_t)((int32_t)q.z >> 31);
        v[3] = q.w ^ cm;
        pre[0] = 0u;
        pre[1] = (uint32_t)__popc(v[0]);
        pre[2] = pre[1] + (uint32_t)__popc(v[1]);
        pre[3] = pre[2] + (uint32_t)__popc(v[2]);
        return pre[3] + (uint32_t)__popc(v[3]);
    };
    for (int t = tw; t < ntrip; t += NP * wpp, u += NP) {
        if (NP == 2 && ((t + wpp) << 8) + 256 < nw) {                    // wave-uniform: two trips, both whole
            uint4 *s0 = reinterpret_cast<uint4*>(tog + (t << 8)) + lane, *s1 = reinterpret_cast<uint4*>(tog + ((t + wpp) << 8)) + lane;
            const uint4 qa = *s0, qb = *s1;
            *s0 = make_uint4(0u, 0u, 0u, 0u);
            *s1 = make_uint4(0u, 0u, 0u, 0u);
            const uint32_t cya = (uint32_t)__builtin_amdgcn_readlane((int)cyl, u), cyb = (uint32_t)__builtin_amdgcn_readlane((int)cyl, u + 1);
            uint32_t va[4], vb[4], pa[4], pb[4];
            const uint32_t oa = lean(qa, cya, va, pa), ob = lean(qb, cyb, vb, pb);
            const uint32_t both = oa | ob << 16;
            const uint32_t excl = wave_incl_add(both) - both;            // (no borrow between the halves: inclusive >= own, half by half)
            const uint32_t ba = (cya & 0x7fffffffu) + (excl & 0xffffu), bb = (cyb & 0x7fffffffu) + (excl >> 16);
            uint4 *da = reinterpret_cast<uint4*>(bd + (t << 8) + 4 * lane), *db = reinterpret_cast<uint4*>(bd + ((t + wpp) << 8) + 4 * lane);
            store_entries2<STREAM>(da, va[0], ba, va[1], ba + pa[1]);
            store_entries2<STREAM>(da + 1, va[2], ba + pa[2], va[3], ba + pa[3]);
            store_entries2<STREAM>(db, vb[0], bb, vb[1], bb + pb[1]);
            store_entries2<STREAM>(db + 1, vb[2], bb + pb[2], vb[3], bb + pb[3]);
            continue;
        }
        if ((NP == 1 || t + wpp >= ntrip) && (t << 8) + 256 < nw) {      // wave-uniform: one trip, whole
            uint4 *s0 = reinterpret_cast<uint4*>(tog + (t << 8)) + lane;
            const uint4 qa = *s0;
            *s0 = make_uint4(0u, 0u, 0u, 0u);
            const uint32_t cya = (uint32_t)__builtin_amdgcn_readlane((int)cyl, u);
            uint32_t va[4], pa[4];
            const uint32_t oa = lean(qa, cya, va, pa);
            const uint32_t ba = (cya & 0x7fffffffu) + wave_incl_add(oa) - oa;
            uint4 *da = reinterpret_cast<uint4*>(bd + (t << 8) + 4 * lane);
            store_entries2<STREAM>(da, va[0], ba, va[1], ba + pa[1]);
            store_entries2<STREAM>(da + 1, va[2], ba + pa[2], va[3], ba + pa[3]);
            continue;
        }
        const int tt[2] = {t, t + wpp};
        const bool on[2] = {true, NP > 1 && t + wpp < ntrip};            // wave-uniform
        uint4 q[2];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            uint4 *src = reinterpret_cast<uint4*>(tog + (tt[j] << 8)) + lane;
            const bool in = on[j] && (tt[j] << 8) + 4 * lane < nw;
            q[j] = in ? *src : make_uint4(0u, 0u, 0u, 0u);
            if (in) *src = make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            if (!on[j]) continue;                                        // (wave-uniform: a row's odd last trip has no partner -- before,
                                                                         //  its empty partner was computed and scanned all the same)
            const uint32_t cy = (uint32_t)__builtin_amdgcn_readlane((int)cyl, u + j);
            const int i0 = (tt[j] << 8) + 4 * lane;
            const uint32_t tq[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
            const uint64_t par = __ballot((int32_t)(tq[0] ^ tq[1] ^ tq[2] ^ tq[3]) < 0);
            uint32_t cm = 0u - ((lanes_below(par) ^ (cy >> 31)) & 1u);
            uint32_t v[4], pre[4], ones = 0;
            if ((tt[j] << 8) + 256 < nw) {                        // wave-uniform: the trip holds neither the row's last word nor
#pragma unroll                                                    // anything behind it (all but the last trip of a row)
                for (int k = 0; k < 4; ++k) {
                    v[k] = tq[k] ^ cm; pre[k] = ones; ones += (uint32_t)__popc(v[k]);
                    cm ^= (uint32_t)((int32_t)tq[k] >> 31);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint32_t x = tq[k] ^ cm;
                    if (i0 + k == nw - 1) x &= tail_mask;
                    if (i0 + k >= nw) x = 0u;
                    v[k] = x; pre[k] = ones; ones += (uint32_t)__popc(x);
                    cm ^= (uint32_t)((int32_t)tq[k] >> 31);
                }
            }
            const uint32_t incl = wave_incl_add(ones);
            const uint32_t b = (cy & 0x7fffffffu) + incl - ones;
            if (on[j]) {
                if (i0 + 3 < nw) {
                    uint4 *dst = reinterpret_cast<uint4*>(bd + i0);
                    store_entries2<STREAM>(dst, v[0], b, v[1], b + pre[1]);
                    store_entries2<STREAM>(dst + 1, v[2], b + pre[2], v[3], b + pre[3]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (i0 + k < nw) bd[i0 + k] = make_uint2(v[k], b + pre[k]);
                }
            }
        }
    }
}

// ---- team-parallel RLE decode (wide cohorts): wave tw of a team owns the 256-byte chunks tw, tw + wpp, ...
// of the string.  Where a chunk starts in the row and the bit before it come from the ROW INDEX, a side
// table built once per file (rowindex_kernel), so a chunk becomes toggles without waiting for the others.
struct ChunkDecode { uint32_t l[4], before[4], bit[4], run; bool valid[4]; bool stop; };

__device__ __forceinline__ ChunkDecode decode_chunk(uint32_t w, uint32_t k0, uint32_t len, int lane)
{
    ChunkDecode d;
    bool anyz = false;
    uint32_t byte[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        byte[i] = (w >> (8 * i)) & 255u;
        d.valid[i] = k0 + i < len;
        d.bit[i] = byte[i] & 1u;
        anyz = anyz || (d.valid[i] && byte[i] == 0u);
    }
    const uint64_t z = __ballot(anyz);
    d.stop = z != 0;
    if (z) {
        const int first = __ffsll((unsigned long long)z) - 1;
        bool dead = lane > first;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (lane == first && byte[i] == 0u) dead = true;
            d.valid[i] = d.valid[i] && !dead;
        }
    }
    d.run = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { d.l[i] = d.valid[i] ? rle_len(byte[i]) : 0u; d.before[i] = d.run; d.run += d.l[i]; }
    return d;
}

// bit of the last valid byte of the chunk (wave-uniform), 0 if the chunk is empty
__device__ __forceinline__ uint32_t chunk_last_bit(const ChunkDecode &d, int lane)
{
    uint32_t nv = 0, lastb = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (d.valid[i]) { ++nv; lastb = d.bit[i]; }
    const uint64_t has = __ballot(nv != 0u);
    if (!has) return 0u;
    const int top = 63 - __builtin_clzll((unsigned long long)has);
    return (uint32_t)__builtin_amdgcn_readlane((int)lastb, top);
}

// words[stride * w] is the toggle word of row word w (stride 2: the .x of the {bits, before} entries; stride 1: a
// separate toggle array)
__device__ __forceinline__ void chunk_toggles(const ScanArgs &a, uint32_t *words, int stride, const ChunkDecode &d,
                                              uint32_t pos, uint32_t prevbit, int lane)
{
    const uint32_t m = (uint32_t)a.m;
    const uint32_t incl = wave_incl_add(d.run);
    const uint32_t lane_start = pos + incl - d.run;
    // bit of the byte before this lane's first byte: the last valid byte of the lane below (chunks are
    // dense, so a lane below a lane with data is full)
    uint32_t lastb = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) if (d.valid[i]) lastb = d.bit[i];
    uint32_t pb = wave_shr1(lastb, prevbit);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t start = lane_start + d.before[i];
        if (d.valid[i] && d.bit[i] != pb && start < m && !(BGTH_SKIP(a, 64)))
            atomicXor(words + (size_t)(start >> 5) * stride, 0xffffffffu << (start & 31));
        if (d.valid[i]) pb = d.bit[i];
    }
}

// whole plane-row by one wave
__device__ __forceinline__ void build_plane_row(const ScanArgs &a, const uint8_t *__restrict__ rle, uint2 *bd,
                                                uint32_t *n0_out, uint64_t desc, uint32_t pre0, uint32_t pre1, int lane,
                                                uint32_t tail_mask)
{
    const int nw = a.nw;
    for (int i = lane; i < nw; i += 64) bd[i] = make_uint2(0u, 0u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (!(BGTH_SKIP(a, 2))) rle_toggles(a, rle, bd, desc, pre0, pre1, 0u, 0u, 2, lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    uint32_t carry_x = 0, carry_c = 0;
    if (!(BGTH_SKIP(a, 4))) directory_pass(bd, 0, nw, nw, tail_mask, carry_x, carry_c, lane);
    if (lane == 0) *n0_out = (uint32_t)a.m - carry_c;
}

// Start ranks of a thread's CPT tracked slots (chunk c = chunk0 + j, slot = 64 c + lane) in sub-block `blk`; rk = the sub-block's
// checkpoint, [2][m] ranks by column.  Complemented ranks (see the row step); `pad` for slots behind the selection.
//   a.order0 (whole cohort, one group, counts only -- the counts do not care which lane tracks which column): slot s tracks the
//   column whose plane-0 rank at the checkpoint is s, so that the 64 lanes of a wave start on 64 consecutive ranks and -- PBWT
//   order keeps neighbours together -- read few distinct entries of a plane-0 row instead of 64 random ones (profiles/r05_lds).
// Every load is unconditional (clamped index, the result selected afterwards) and the dependent gathers form a second loop: as
// `if (col >= 0) rank = rk[col]` per column the compiler emitted load - s_waitcnt vmcnt(0) - branch - load - wait ..., THREE
// memory round trips per column one after the other: ~100 of them at the start of every workgroup of the HRC shape, whose
// sub-blocks are only 128 rows long (round 5).
template <int CPT>
__device__ __forceinline__ void load_start_ranks(const ScanArgs &a, int64_t blk, const int32_t *__restrict__ rk, int chunk0, int lane,
                                                 bool skip_all, uint32_t pad, uint32_t (&r0)[CPT], uint32_t (&r1)[CPT])
{
    const int m = a.m;
    constexpr int B = 16;                                               // columns per batch of loads in flight (registers)
    if (a.order0) {
        const int32_t *__restrict__ ord = a.order0 + blk * a.order_blk_stride;
#pragma unroll
        for (int j0 = 0; j0 < CPT; j0 += B) {
            uint32_t t[B];
#pragma unroll
            for (int j = j0; j < j0 + B && j < CPT; ++j) {
                const int slot = (chunk0 + j) * 64 + lane;
                t[j - j0] = (uint32_t)ord[slot < m ? slot : 0];
            }
#pragma unroll
            for (int j = j0; j < j0 + B && j < CPT; ++j) {
                const int slot = (chunk0 + j) * 64 + lane;
                const bool on = !skip_all && chunk0 + j < a.n_chunks && slot < m;
                r0[j] = ~(on ? (uint32_t)slot : pad);
                r1[j] = ~(on ? t[j - j0] : pad);
            }
        }
        return;
    }
    const int last = a.n_chunks * 64 - 1;
#pragma unroll
    for (int j0 = 0; j0 < CPT; j0 += B) {
        int col[B];
        uint32_t t0[B], t1[B];
#pragma unroll
        for (int j = j0; j < j0 + B && j < CPT; ++j) {
            const int slot = (chunk0 + j) * 64 + lane;
            col[j - j0] = a.slot_col[slot <= last ? slot : 0];
        }
#pragma unroll
        for (int j = j0; j < j0 + B && j < CPT; ++j) {
            const int cc = col[j - j0] >= 0 ? col[j - j0] : 0;
            t0[j - j0] = (uint32_t)rk[cc];
            t1[j - j0] = (uint32_t)rk[m + cc];
        }
#pragma unroll
        for (int j = j0; j < j0 + B && j < CPT; ++j) {
            const bool on = !skip_all && chunk0 + j < a.n_chunks && col[j - j0] >= 0;
            r0[j] = ~(on ? t0[j - j0] : pad);
            r1[j] = ~(on ? t1[j - j0] : pad);
        }
    }
}

// SNAP = the image-open pass that snapshots the ranks at every sub-checkpoint row (ScanArgs::snap).  A template switch, not a
// run-time test: as a run-time test its per-column skeleton (a predicate, a branch and two VALU instructions per column and ROW)
// stayed in the walk loop of every scan -- 40 of the 360 VALU instructions of a C2 row.
// Template switches:  MULTI = more than one sample group (per-chunk LDS atomics instead of per-wave
// scalars);  GT = also emit the two bit planes of every row (slot order) for genotype output;
// ZP = with the shortcut for rows whose plane 1 is all zero (chosen per image: worth a taken branch per statement
// only if such rows exist);  TEAM = wide cohort: few rows fit the LDS, every plane-row is built by a team of waves (single batch
// buffer); otherwise a wave builds its plane-rows alone and batches are pipelined over two buffers.
#define BGTH_TICK(slot) do { if (BGTH_TIMES(a)) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    tsum[slot] += now_ - tlast; tlast = now_; } } while (0)

// WC: whole cohort, one group, counts only (ScanArgs::whole_counts; narrow pipelined kernels without the empty-plane shortcut): the
// per-plane ones come from the rows' strings, only n(code 3) is counted (BGTH_COUNT3).
// PK (round 6 A/B, with WC, m <= 65,504): a column's two ranks packed in one register (step4pk)
template <int NT, int CPT, bool MULTI, bool GT, bool TEAM, bool ZP, bool SNAP = false, bool WC = false, bool PK = false>
__global__ __launch_bounds__(NT) void scan_kernel(const ScanArgs a, const uint64_t *__restrict__ rowdesc,
                                                  const uint8_t *__restrict__ rle,
                                                  const uint32_t *__restrict__ chunkinfo,
                                                  const uint32_t *__restrict__ segc)
{   // rowdesc / rle are separate `const __restrict__` arguments (not members of `a`) so that hipcc knows
    // they are invariant: the wave-uniform descriptor loads then go through the scalar cache (s_load,
    // lgkmcnt) and do not force a vmcnt(0) that would drain the prefetched string loads.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NWAVE = NT / 64;
    static_assert(CPT % 2 == 0, "columns per thread are stepped in pairs");
    static_assert(CPT * 64 < 65536, "per-wave counts of a row are kept in 16 bits");
    static_assert(!WC || (!MULTI && !GT && !TEAM && !ZP), "whole-cohort counting serves the plain narrow kernel");
    static_assert(!PK || (WC && !SNAP && CPT % 4 == 0), "packed ranks: whole-cohort counting, four columns per statement");
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
    const int m = a.m, nw = a.nw, nwp = (nw + 2) & ~1, K = a.K, G = a.G;   // rows 16-byte aligned
    uint2    *BD   = reinterpret_cast<uint2*>(smem);                    // [nbuf][2K][nwp]
    int32_t  *lcnt = reinterpret_cast<int32_t*>(smem + (size_t)16 * K * nwp * (TEAM ? 1 : 2));
    //   !MULTI: uint2 [K][NWAVE] one private slot per wave and row, three 16-bit counts (a wave counts at most
    //           64 * CPT < 65536 per row; 8 bytes instead of 16 let C2 keep 8 rows per batch)   MULTI: int32 [K][G][3] (LDS atomics)
    uint32_t *n0s  = reinterpret_cast<uint32_t*>(lcnt + (TEAM ? 1 : 2) * (MULTI ? K * G * 3 : K * NWAVE * 2)); // [nbuf][2K]
    // team mode, when it fits: toggles of the NEXT batch go to their own array [2K][nwt] while the current one is walked
    const int nwt = (nw + 4) & ~3;
    uint32_t *TOG = (TEAM && a.tog_off) ? reinterpret_cast<uint32_t*>(smem + a.tog_off) : nullptr;
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
        load_start_ranks<CPT>(a, blk, rk, chunk0, lane, BGTH_SKIP(a, 8) != 0, pad_rank, r0, r1);
    }
    uint32_t rp[PK ? CPT : 1];                                           // PK: {plane-1 rank : plane-0 rank}, 16-bit complements
    if constexpr (PK) {
#pragma unroll
        for (int j = 0; j < CPT; ++j) rp[j] = (r1[j] << 16) | (r0[j] & 0xffffu);
    }
    if (MULTI) for (int i = tid; i < (TEAM ? 1 : 2) * K * G * 3; i += NT) lcnt[i] = 0;
    // MULTI: the group of every row-step statement of this wave (four chunks, the tail two): slots are laid out group by group,
    // so nearly every statement belongs to ONE group and its counts accumulate on the scalar unit like the ungrouped ones do,
    // leaving for the LDS only where the group changes (255: the statement straddles two groups -- its chunks are counted one
    // by one).  Before: three LDS atomics per chunk and row, 2.6 x the time of the ungrouped scan.
    constexpr int NSTMT = (CPT + 3) / 4;
    // (four group numbers to a word, kept in SGPRs: as a VGPR per statement the table cost the 50-column walk-only kernel 25 registers
    //  it did not have -- it spilled, and a two-group scan of a C4-width cohort took 2.7 x the ungrouped time)
    uint32_t stmt_groups[MULTI ? (NSTMT + 3) / 4 : 1] = {};
    auto stmt_group = [&](int q) -> uint32_t { return (stmt_groups[q >> 2] >> (8 * (q & 3))) & 255u; };
    if constexpr (MULTI) {
        uint32_t last = 0;
#pragma unroll
        for (int q = 0; q < NSTMT; ++q) {
            uint32_t g = 254u;                                                   // 254: no chunk of the selection in this statement
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = chunk0 + 4 * q + u;
                if (4 * q + u < CPT && c < a.n_chunks) {
                    const uint32_t x = a.chunk_desc[c] & 255u;
                    g = g == 254u ? x : (g == x ? g : 255u);
                }
            }
            if (g == 254u) g = last;                                             // (padding chunks count nothing: any group)
            stmt_groups[q >> 2] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(stmt_groups[q >> 2] | g << (8 * (q & 3))));
            if (g != 255u) last = g;
        }
    }
    for (int i = tid; i < (TEAM ? 1 : 2) * 2 * K; i += NT)
        BD[(size_t)i * nwp + nw] = make_uint2(0u, 0u);
    if (TEAM && TOG) {                                                   // cleared before any wave toggles into it
        for (int i = tid; i < 2 * K * nwt; i += NT) TOG[i] = 0u;
        lds_barrier();
    }

    const uint32_t tail_mask = (m & 31) ? ((1u << (m & 31)) - 1u) : 0xffffffffu;

    // ---- who builds which plane-row of a batch.  wpp == 1: wave w builds plane-rows w and w + NWAVE
    // on its own (K <= NWAVE).  wpp > 1 (wide cohorts: few rows fit the LDS): a team of wpp waves builds
    // plane-row (wave / wpp) together, synchronised by workgroup barriers.
    const int wpp = TEAM ? a.wpp : 1;
    // Narrow mode: which plane-row of a batch this wave builds.  Even plane-rows are plane 0 (long strings, several
    // times the build work of plane 1); waves w, w+4, w+8, w+12 share a SIMD, so the parity is flipped for every
    // second group of four waves: each SIMD then builds two plane-0 and two plane-1 rows per batch.
    const int build_slot = wave ^ ((wave >> 2) & 1);
    const int team = wave / wpp, tw = wave - team * wpp;

    // ---- software prefetch of the RLE strings: the row descriptors run two batches ahead of phase A,
    // the first 512 bytes of every string one batch ahead.
    //   !TEAM: slot i = plane-row (wave + i*NWAVE) of the batch, pre[i][0..1] = its chunks 0 and 1
    //   TEAM : both slots describe the team's string; pre[0][j] = data of chunk tw + j*wpp,
    //          pre[1][j] = that chunk's row-index record {start position | bit before << 31}
    uint64_t dsc[2], dsc_next[2];
    uint32_t pre[2][2];
    auto fetch_desc = [&](int64_t rb_, int i) -> uint64_t {
        const int p = wpp == 1 ? build_slot + i * NWAVE : team;
        const int64_t left = blk_end - rb_;
        const int kc = (int)(left < K ? left : K);
        return (rb_ < blk_end && p < 2 * kc) ? rowdesc[2 * rb_ + p] : 0ull;
    };
    auto fetch_data = [&](uint64_t d, int c) -> uint32_t {
        const uint32_t len = (uint32_t)(d >> kDescLenShift);
        const uint32_t k0 = (uint32_t)c * 256u + 4u * (uint32_t)lane;
        return k0 < len ? reinterpret_cast<const uint32_t*>(rle + (d & kDescOffMask))[c * 64 + lane] : 0u;
    };
    auto fetch_info = [&](uint64_t d, int c, int64_t rb_) -> uint32_t {          // TEAM only
        const uint32_t len = (uint32_t)(d >> kDescLenShift);
        return (uint32_t)c * 256u < len ? chunkinfo[(((d & kDescOffMask) + (uint64_t)c * 256u) >> 8) + (uint64_t)(2 * rb_ + team)] : 0u;
    };
    auto fetch_pre = [&](int64_t rb_) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (!TEAM) { pre[i][0] = fetch_data(dsc[i], 0); pre[i][1] = fetch_data(dsc[i], 1); }
            else if (i == 0) { pre[0][0] = fetch_data(dsc[0], tw); pre[0][1] = fetch_data(dsc[0], tw + wpp); }
            else { pre[1][0] = fetch_info(dsc[0], tw, rb_); pre[1][1] = fetch_info(dsc[0], tw + wpp, rb_); }
        }
    };
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        dsc[i] = fetch_desc(blk_beg, i);
        dsc_next[i] = fetch_desc(blk_beg + K, i);
    }
    fetch_pre(blk_beg);
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = BGTH_TIMES(a) ? __builtin_amdgcn_s_memtime() : 0ull;

    // LDS regions of batch buffer `buf`
    const size_t bd_stride = (size_t)2 * K * nwp;                        // uint2 entries per buffer
    const int cnt_stride = MULTI ? K * G * 3 : K * NWAVE * 2;            // ints per buffer

    // ================= phase A: build the bit-vectors of the batch at rows [rbA, rbA+Kc) into buffer buf ===
    // First take the strings fetched while the previous batch was processed and immediately issue the
    // loads of the batch after this one (and the descriptors of the one after that): they have a whole
    // batch to arrive, whatever vmcnt wait the compiler places at their first use.
    auto phase_a = [&](int buf, int64_t rbA) {
        const int Kc = (int)((blk_end - rbA) < K ? (blk_end - rbA) : K);
        uint2 *BDb = BD + buf * bd_stride;
        uint32_t *n0b = n0s + buf * 2 * K;
        uint64_t cdsc[2];
        uint32_t cpre[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { cdsc[i] = dsc[i]; cpre[i][0] = pre[i][0]; cpre[i][1] = pre[i][1]; }
        asm volatile("" : "+v"(cpre[0][0]), "+v"(cpre[0][1]), "+v"(cpre[1][0]), "+v"(cpre[1][1]));   // arrived: pin the wait here
#pragma unroll
        for (int i = 0; i < 2; ++i) dsc[i] = dsc_next[i];
        fetch_pre(rbA + K);
#pragma unroll
        for (int i = 0; i < 2; ++i) dsc_next[i] = fetch_desc(rbA + 2 * K, i);
        BGTH_TICK(0);
        if constexpr (!TEAM) {
            if (!(BGTH_SKIP(a, 0x4000))) set_wave_priority(3);          // the build is a latency chain of few instructions: let it through
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int p = build_slot + i * NWAVE;
                if (p < 2 * Kc)
                    build_plane_row(a, rle, BDb + (size_t)p * nwp, n0b + p, cdsc[i], cpre[i][0], cpre[i][1], lane, tail_mask);
            }
        } else {
            // team mode: wave tw of team `team` (= plane-row of the batch) takes the string's chunks tw, tw+wpp, ..
            // and the directory trips (256 words each) tw, tw+wpp, ..; both start from row-index records, so the
            // only synchronisation is  clear | toggles | directory.
            const bool active = team < 2 * Kc;
            uint2 *bd = BDb + (size_t)team * nwp;
            const int64_t sidx = 2 * rbA + team;
            const uint32_t slen = (uint32_t)(cdsc[0] >> kDescLenShift);
            const int ntrip = (nw + 255) >> 8;
            // carries of this wave's directory trips (lane u <-> trip tw + u*wpp) and the row's number of ones:
            // in flight during the toggles
            uint32_t cyl = 0, tot1 = 0;
            if (active) {
                const uint32_t *sc = segc + (size_t)sidx * (size_t)(a.S8 + 1);
                const int t = tw + lane * wpp;
                cyl = t < ntrip ? sc[t] : 0u;
                tot1 = sc[a.S8];
            }
            if (active) {                                               // clear the row, two entries per store
                uint4 *z = reinterpret_cast<uint4*>(bd);
                for (int i = tw * 64 + lane; 2 * i < nw; i += wpp * 64) z[i] = make_uint4(0u, 0u, 0u, 0u);
            }
            BGTH_TICK(1);
            lds_barrier();
            BGTH_TICK(2);
            if (active && !(BGTH_SKIP(a, 2))) {
                int i = 0;
                for (int c = tw; (uint32_t)c * 256u < slen; c += wpp, ++i) {
                    uint32_t w, ci;
                    if (i == 0) { w = cpre[0][0]; ci = cpre[1][0]; }
                    else if (i == 1) { w = cpre[0][1]; ci = cpre[1][1]; }
                    else { w = fetch_data(cdsc[0], c); ci = fetch_info(cdsc[0], c, rbA); }
                    if (ci & kChunkDead) break;                      // behind a terminating zero byte
                    const ChunkDecode cd = decode_chunk(w, (uint32_t)c * 256u + 4u * (uint32_t)lane, slen, lane);
                    chunk_toggles(a, &bd[0].x, 2, cd, ci & kChunkPosMask, ci >> 31, lane);
                }
            }
            BGTH_TICK(3);
            lds_barrier();
            BGTH_TICK(4);
            if (active && !(BGTH_SKIP(a, 4))) {
                int u = 0;
                for (int t = tw; t < ntrip; t += wpp, ++u) {
                    const uint32_t cy = (uint32_t)__builtin_amdgcn_readlane((int)cyl, u);
                    uint32_t cx = cy >> 31, cnt = cy & 0x7fffffffu;
                    const int w0 = t << 8, w1 = (w0 + 256) < nw ? (w0 + 256) : nw;
                    directory_pass(bd, w0, w1, nw, tail_mask, cx, cnt, lane);
                }
                if (tw == 0 && lane == 0) n0b[team] = (uint32_t)m - tot1;
            }
        }
        BGTH_TICK(5);
    };

    // ================= team mode with a separate toggle array: phase A in two halves =================
    // toggles(rbA) may run while other waves still walk the previous batch (it touches only TOG);
    // directory(rbA) rewrites the bit-vectors and needs every wave past its walk.
    uint32_t keep_cyl = 0, keep_tot = 0;
    auto team_toggles = [&](int64_t rbA) {
        const int Kc = (int)((blk_end - rbA) < K ? (blk_end - rbA) : K);
        const uint64_t cd0 = dsc[0];
        uint32_t c00 = pre[0][0], c01 = pre[0][1], c10 = pre[1][0], c11 = pre[1][1];
        asm volatile("" : "+v"(c00), "+v"(c01), "+v"(c10), "+v"(c11));
#pragma unroll
        for (int i = 0; i < 2; ++i) dsc[i] = dsc_next[i];
        fetch_pre(rbA + K);
#pragma unroll
        for (int i = 0; i < 2; ++i) dsc_next[i] = fetch_desc(rbA + 2 * K, i);
        BGTH_TICK(0);
        const bool active = team < 2 * Kc;
        const int64_t sidx = 2 * rbA + team;
        const uint32_t slen = (uint32_t)(cd0 >> kDescLenShift);
        const int ntrip = (nw + 255) >> 8;
        keep_cyl = 0; keep_tot = 0;
        if (active) {
            const uint32_t *sc = segc + (size_t)sidx * (size_t)(a.S8 + 1);
            const int t = tw + lane * wpp;
            keep_cyl = t < ntrip ? sc[t] : 0u;
            keep_tot = sc[a.S8];
        }
        if (active && !(BGTH_SKIP(a, 2))) {
            uint32_t *trow = TOG + (size_t)team * nwt;
            int i = 0;
            for (int c = tw; (uint32_t)c * 256u < slen; c += wpp, ++i) {
                uint32_t w, ci;
                if (i == 0) { w = c00; ci = c10; }
                else if (i == 1) { w = c01; ci = c11; }
                else { w = fetch_data(cd0, c); ci = fetch_info(cd0, c, rbA); }
                if (ci & kChunkDead) break;
                const ChunkDecode cd = decode_chunk(w, (uint32_t)c * 256u + 4u * (uint32_t)lane, slen, lane);
                chunk_toggles(a, trow, 1, cd, ci & kChunkPosMask, ci >> 31, lane);
            }
        }
        BGTH_TICK(3);
    };
    auto team_directory = [&](int64_t rbA) {
        const int Kc = (int)((blk_end - rbA) < K ? (blk_end - rbA) : K);
        if (team < 2 * Kc && !(BGTH_SKIP(a, 4))) {
            directory_trips_tog<(CPT > 88 ? 1 : 2)>(TOG + (size_t)team * nwt, BD + (size_t)team * nwp, tw, wpp, (nw + 255) >> 8, nw, tail_mask,
                                keep_cyl, lane);
            if (tw == 0 && lane == 0) n0s[team] = (uint32_t)m - keep_tot;
        }
        BGTH_TICK(5);
    };

    // ================= phase B: walk the Kc rows of buffer buf, ranks stay in registers =================
    auto phase_b = [&](int buf, int64_t rb, int Kc) {
        const uint32_t *n0b = n0s + buf * 2 * K;
        int32_t *lcb = lcnt + buf * cnt_stride;
        const uint32_t bufbase = lds0 + (uint32_t)(buf * bd_stride) * 8u;
        // The rows' zero counts: lane l holds the batch's l-th (one LDS read per batch) and a row takes its two by v_readlane --
        // read from the LDS row by row they were a round trip (ds_read, s_waitcnt, two v_readfirstlane) at the top of every row,
        // ~5 % of a 20-column row.
        // (K <= waves per workgroup <= 16: the 2 K values always fit the 64 lanes)
        const uint32_t n0_lane = n0b[lane < 2 * K ? lane : 0];
        if (!TEAM && NT >= 512) __builtin_amdgcn_s_setprio(3);
        if (!(BGTH_SKIP(a, 1)))
        for (int k = 0; k < Kc; ++k) {
            // operands of the row step: (LDS byte address of the plane-row) - 8 and -n0 (see BGTH_TAIL)
            const uint32_t base0 = bufbase + (uint32_t)(2 * k) * (uint32_t)nwp * 8u - 8u;
            uint32_t base1 = base0 + (uint32_t)nwp * 8u;
            const uint32_t n00 = 0u - (uint32_t)__builtin_amdgcn_readlane((int)n0_lane, 2 * k);
            const uint32_t n01 = 0u - (uint32_t)__builtin_amdgcn_readlane((int)n0_lane, 2 * k + 1);
            const bool emit = (rb + k) >= a.row0;
            // The SIMD arbiter prefers its oldest wave: left alone, waves 0-3 race through a batch and idle at the
            // barrier while waves 12-15 finish it nearly alone (measured: walk 96 vs 189 ticks).  Rotating the user
            // priority over the rows of a batch gives the four waves of a SIMD equal progress.
            if (!TEAM && !(BGTH_SKIP(a, 0x2000))) {
                // 512 and 1024 threads (one to three workgroups per CU): a wave's priority falls over the LAST rows of the batch, as in
                // the walk-only kernel (scan_dir.hip): 3 until four rows before the end, then 2, 2, 1, 0.  C2, ms per 1 M sites: no
                // priorities 11.74, rotated over the rows (until round 5) 10.37, falling evenly over the batch 10.54 (+0.3 for the
                // run-time choice in that build), 2, 1, 0 over its last three rows 10.05, 2, 2, 1, 0 over its last four 9.95,
                // 2, 2, 1, 1, 0 10.01; 3,500 / 5,000 / 7,000 / 8,500 samples -4.1 / -4.2 / -2.3 / -4.9 %.  Five workgroups of 256 threads per CU, each at its own point of its batch, keep the rotation:
                // 2,504 samples 2.65 against 2.74.
                if constexpr (NT >= 512) { if (k + 4 >= Kc) set_wave_priority_uniform((uint32_t)(Kc - 1 - k > 2 ? 2 : Kc - 1 - k)); }
                else set_wave_priority_uniform((uint32_t)((wave >> 2) + k));
            }   // (team mode: by columns, below; a priority that falls with the rows of the batch instead measured the same)
            if (ZP && n01 == 0u - (uint32_t)m) base1 = 0u;   // plane 1 all zero: its lookups are skipped (see step2)
            if (SNAP && a.snap && rb + k > blk_beg && ((rb + k) & (((int64_t)1 << a.snap_shift) - 1)) == 0) {
                // sub-checkpoint: the ranks before this row (image-open pass only)
                int32_t *dst = a.snap + ((rb + k) >> a.snap_shift) * (int64_t)(2 * m);
                int ln = lane;
                asm volatile("" : "+v"(ln));      // keeps the CPT slot addresses from being hoisted out of the row loop (VGPRs)
#pragma unroll
                for (int j = 0; j < CPT; ++j) {
                    const int c = chunk0 + j;
                    const int col = c < a.n_chunks ? a.slot_col[c * 64 + ln] : -1;
                    if (col >= 0) { dst[col] = (int32_t)~r0[j]; dst[m + col] = (int32_t)~r1[j]; }
                }
            }
            // ones of plane 0, ones of plane 1, ones in both:  n(code1) = ca - cc, n(code2) = cb - cc
            uint32_t ca = 0, cb = 0, cc = 0;
            constexpr int NKEEP = (CPT + 63) / 64;                     // lane l keeps the masks of chunks l, l + 64
            uint64_t keep0[NKEEP] = {}, keep1[NKEEP] = {};
            uint32_t pa = 0, pb = 0, pc = 0;                          // MULTI: ca / cb / cc at the start of the current run
            uint32_t run_g = MULTI ? (uint32_t)__builtin_amdgcn_readfirstlane((int)stmt_group(0)) : 0u;   // MULTI: the group ca / cb / cc are accumulating for
            // ... and their way into the row's per-group counts (lane 0 adds; the sums themselves are wave-uniform scalars)
#define BGTH_FLUSH_GROUP(GRP)                                                                              \
            do {                                                                                           \
                const uint32_t a_ = ca - pa, b_ = cb - pb, n3_ = cc - pc, n1_ = a_ - n3_, n2_ = b_ - n3_, g_ = (GRP);                         \
                if (emit && g_ < 254u && lane == 0 && !BGTH_SKIP(a, 0x400000)) {   /* (profiling build: no flush, timing only) */ \
                    int32_t *dst = lcb + ((size_t)k * G + g_) * 3;                                         \
                    if (n1_) atomicAdd(dst + 0, (int32_t)n1_);                                             \
                    if (n2_) atomicAdd(dst + 1, (int32_t)n2_);                                             \
                    if (n3_) atomicAdd(dst + 2, (int32_t)n3_);                                             \
                }                                                                                          \
            } while (0)
#pragma unroll
            for (int j = 0; j < CPT; j += 4) {
                uint64_t m0[4] = {0, 0, 0, 0}, m1[4] = {0, 0, 0, 0};
                const int NC = (CPT - j) >= 4 ? 4 : 2;                // CPT is even: the tail is one pair
                const uint32_t sg = MULTI ? (uint32_t)__builtin_amdgcn_readfirstlane((int)stmt_group(j >> 2)) : 0u;   // (uniform, and said so)
                if (MULTI && __builtin_expect(sg != run_g, 0)) {             // (out of line: the common path falls through)
                    BGTH_FLUSH_GROUP(run_g);
                    pa = ca; pb = cb; pc = cc;                            // (the scalar sums only ever grow: a run is a difference)
                    run_g = sg;
                }
                // team mode (one row per barrier): a wave's priority falls as it gets through its columns, so that the waves of
                // a SIMD finish the row together instead of one after the other (scan_dir.hip: -8 % for the walk-only kernel)
                if (TEAM && BGTH_WALK_PRIO(a)) {                         // (the steps close to the end of the row: scan_dir.hip)
                    constexpr int NS = (CPT + 3) / 4;
                    if (j == 0) __builtin_amdgcn_s_setprio(3);
                    else if (j / 4 == NS * 5 / 8) __builtin_amdgcn_s_setprio(2);
                    else if (j / 4 == NS * 13 / 16) __builtin_amdgcn_s_setprio(1);
                    else if (j / 4 == NS * 15 / 16) __builtin_amdgcn_s_setprio(0);
                }
                if constexpr (PK) {
                    uint32_t qp[4] = {rp[j], rp[j + 1], rp[j + 2], rp[j + 3]};
                    // (bases: + 8 for the "- 8" of the plain form, + 8 x 2047 for the unsigned word index; N0PK: both planes' -n0)
                    step4pk(qp, cc, base0 + 16384u, base1 + 16384u, (n01 << 16) | (n00 & 0xffffu));
#pragma unroll
                    for (int u = 0; u < 4; ++u) rp[j + u] = qp[u];
                } else if (NC == 4) {
                    uint32_t q0[4] = {r0[j], r0[j + 1], r0[j + 2], r0[j + 3]};
                    uint32_t q1[4] = {r1[j], r1[j + 1], r1[j + 2], r1[j + 3]};
                    step4<ZP, WC>(q0, q1, m0, m1, ca, cb, cc, base0, base1, n00, n01);
#pragma unroll
                    for (int u = 0; u < 4; ++u) { r0[j + u] = q0[u]; r1[j + u] = q1[u]; }
                } else {
                    step2<ZP, WC>(r0[j], r1[j], r0[j + 1], r1[j + 1], m0[0], m1[0], m0[1], m1[1], ca, cb, cc, base0, base1, n00, n01);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (u >= NC) break;
                    if (GT && lane == ((j + u) & 63)) { keep0[(j + u) >> 6] = m0[u]; keep1[(j + u) >> 6] = m1[u]; }
                    if (MULTI && __builtin_expect(run_g == 255u, 0)) {                            // a statement across two groups: chunk by chunk
                        const int c = chunk0 + j + u;                        // wave-uniform
                        if (emit && lane == 0 && c < a.n_chunks) {
                            int32_t *dst = lcb + ((size_t)k * G + (a.chunk_desc[c] & 255u)) * 3;
                            atomicAdd(dst + 0, __builtin_amdgcn_readfirstlane(__popcll(m0[u] & ~m1[u])));
                            atomicAdd(dst + 1, __builtin_amdgcn_readfirstlane(__popcll(~m0[u] & m1[u])));
                            atomicAdd(dst + 2, __builtin_amdgcn_readfirstlane(__popcll(m0[u] & m1[u])));
                        }
                    }
                }
                if (MULTI && __builtin_expect(run_g == 255u, 0)) { pa = ca; pb = cb; pc = cc; }  // (its sums were taken from the masks)
            }
            if (MULTI) BGTH_FLUSH_GROUP(run_g);
#undef BGTH_FLUSH_GROUP
            if (!MULTI && lane == 0) {
                // WC: slot = {n(code 3) of this wave's columns, the row's ones of plane 0 (wave 0) / plane 1 (wave 1)}: the plane
                // totals enter once, through slice 0 (n00 / n01 are -zeros: ones = m + n0x)
                if constexpr (WC)
                    reinterpret_cast<uint2*>(lcb)[k * NWAVE + wave] =
                        make_uint2(cc, slice == 0 && wave < 2 ? (uint32_t)m + (wave == 0 ? n00 : n01) : 0u);
                else
                    reinterpret_cast<uint2*>(lcb)[k * NWAVE + wave] = make_uint2((ca - cc) | (cb - cc) << 16, cc);
            }
            if (GT && emit) {
#pragma unroll
                for (int q = 0; q < NKEEP; ++q) {
                    const int c = q * 64 + lane;
                    if (c < CPT && chunk0 + c < a.n_chunks) {
                        const size_t at = (size_t)(rb + k - a.row0) * a.n_chunks + chunk0 + c;
                        a.h0[at] = keep0[q];
                        a.h1[at] = keep1[q];
                    }
                }
            }
        }
        BGTH_TICK(7);
    };

    // ================= phase C: per-row counts of this slice, buffer buf -> HBM =================
    auto phase_c = [&](int buf, int64_t rb, int Kc) {
        int32_t *lcb = lcnt + buf * cnt_stride;
        if (MULTI) {
            for (int i = tid; i < Kc * G * 3; i += NT) {
                const int32_t v = lcb[i];
                if (v) {
                    atomicAdd(a.raw_counts + (size_t)(rb - a.row0) * G * 3 + i, v);
                    lcb[i] = 0;
                }
            }
        } else {
            for (int i = tid; i < Kc * 3; i += NT) {
                const int k = i / 3, comp = i - 3 * k;
                if (rb + k >= a.row0) {
                    int32_t v = 0;
                    if constexpr (WC) {                                  // n(code 1) = ones(plane 0) - n(code 3), n(code 2) likewise
                        int32_t c3 = 0;
#pragma unroll
                        for (int w = 0; w < NWAVE; ++w) c3 += lcb[(k * NWAVE + w) * 2];
                        v = comp == 2 ? c3 : lcb[(k * NWAVE + comp) * 2 + 1] - c3;
                    } else
#pragma unroll
                    for (int w = 0; w < NWAVE; ++w) {
                        const uint32_t x = (uint32_t)lcb[(k * NWAVE + w) * 2 + (comp >> 1)];
                        v += (int32_t)(comp == 0 ? x & 0xffffu : comp == 1 ? x >> 16 : x);
                    }
                    int32_t *dst = a.raw_counts + (size_t)(rb + k - a.row0) * 3 + comp;
                    if (a.n_slices == 1) *dst = v;                   // the only writer of this row: no zero-fill, no atomic
                    else if (v) atomicAdd(dst, v);
                }
            }
        }
    };

    if constexpr (!TEAM) {
        // Pipelined batches (two LDS buffers, narrow cohorts): one barrier per batch.  In iteration b a wave
        // flushes the counts of batch b-1, builds its plane-rows of batch b+1 into the other buffer and walks
        // batch b -- so while some waves sit in the latency-bound build, others keep the VALUs busy with
        // lookups.  The barrier at the end makes batch b+1 complete and buffer b reusable.
        int b = -1;                                    // iteration -1 only builds batch 0
        int64_t rb = blk_beg - K;
        for (; rb < blk_end; rb += K, ++b) {
            if (b > 0) phase_c((b - 1) & 1, rb - K, K);
            BGTH_TICK(2);                                  // (narrow mode: slot 2 = counts flush, slot 1 = barrier wait)
            if (rb + K < blk_end) phase_a((b + 1) & 1, rb + K);
            BGTH_TICK(6);
            if (b >= 0) phase_b(b & 1, rb, (int)((blk_end - rb) < K ? (blk_end - rb) : K));
            lds_barrier();
            BGTH_TICK(1);
        }
        if (b > 0) {
            const int64_t last = rb - K;
            phase_c((b - 1) & 1, last, (int)((blk_end - last) < K ? (blk_end - last) : K));
        }
    } else {
        // Team mode, one loop for both variants (iteration -1 only prepares batch 0):
        //   separate toggle array:  walk(b) + toggles(b+1) | counts(b) + directory(b+1) |      two barriers per batch
        //   toggles in place     :  walk(b) | counts(b) + clear | toggles | directory(b+1) |   four
        for (int64_t rb = blk_beg - K; rb < blk_end; rb += K) {
            const int Kc = (int)((blk_end - rb) < K ? (blk_end - rb) : K);
            const bool cur = rb >= blk_beg, more = rb + K < blk_end;
            if (cur) phase_b(0, rb, Kc);
            if (more && TOG) team_toggles(rb + K);
            lds_barrier();
            BGTH_TICK(4);
            if (cur) phase_c(0, rb, Kc);
            if (more) {
                if (TOG) team_directory(rb + K);
                else phase_a(0, rb + K);
            }
            lds_barrier();
            BGTH_TICK(6);
        }
    }

#ifdef BGTH_ABLATE
    if (BGTH_TIMES(a) && lane == 0 && (!(BGTH_SKIP(a, 0x100)) || wave == ((a.debug_skip >> 16) & 15))) {   // 0x100: one wave only (bits 16..19)
        for (int i = 0; i < 8; ++i) atomicAdd(a.debug_times + i, tsum[i]);
    }
#endif
    if (a.final_rank) {
        int32_t *fin = a.final_rank + (int64_t)bl * a.final_blk_stride;
        if constexpr (PK) {                                              // (back to one 32-bit complement per plane)
#pragma unroll
            for (int j = 0; j < CPT; ++j) { r0[j] = 0xffff0000u | (rp[j] & 0xffffu); r1[j] = 0xffff0000u | (rp[j] >> 16); }
        }
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


}  // namespace bgth
