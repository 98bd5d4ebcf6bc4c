#!/usr/bin/env python3
"""Generates issue_bench.hip (at build time, into the build directory: bgt_amd/csrc/Makefile): the VALU issue experiments of round 4 (profiles/r04_issue/).

Question: the row step of the scan kernels issues at ~4.0 cycles per VALU wave-instruction although
three of its eight instructions are of the class that streams at ~2.2-2.5 cycles.  Every experiment here is a kernel whose
loop body is a fixed list of VALU instructions over eight independent register slots; the table below names the body, the
generator lays it out (slot-major = dependent neighbours, or instruction-major = eight independent neighbours), and the
harness measures shader cycles per wave with s_memtime at 1 / 2 / 4 / 8 waves per SIMD.  No memory is touched: VALU timing
does not depend on the data.

Run:  python gen_issue_bench.py OUT.hip   (`make -C bgt_amd/csrc` does, and builds the result into libbgt_hip_bench.so)
"""
import os
import sys

OUT = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "build", "gen", "issue_bench.hip")

# register slots: eight independent copies of every operand
REG = {"A": 10, "B": 18, "C": 26, "D": 34, "E": 42, "F": 50}          # v10..v57: < 64 VGPRs, so 8 waves per SIMD fit
MASK0 = 36                                                             # s[36:37] .. s[50:51]
CLOBBER = ["v%d" % i for i in range(10, 58)] + ["s%d" % i for i in list(range(20, 32)) + list(range(36, 52))] + ["vcc", "scc"]


def fmt(t, slot):
    d = {k: "v%d" % (v + slot) for k, v in REG.items()}
    d["M"] = "s[%d:%d]" % (MASK0 + 2 * slot, MASK0 + 2 * slot + 1)
    d["A2"] = "v%d" % (REG["A"] + (slot ^ 1))                          # the neighbour slot's A (for pairwise ops)
    d["M2"] = "s[%d:%d]" % (MASK0 + 2 * (slot ^ 1), MASK0 + 2 * (slot ^ 1) + 1)
    d["E2"] = "v%d" % (REG["E"] + (slot ^ 1))                          # the neighbour slot's sign mask
    return t.format(**d)


def body(templates, layout="slot", reps=1, pair=None):
    """layout 'slot': all instructions of slot 0, then slot 1 ... (neighbours depend on each other when the templates
    chain through the same registers); 'instr': instruction k for all eight slots, then k+1 (eight independent neighbours);
    'il2' / 'il4': slots in groups of 2 / 4, instruction-major inside a group.
    pair: templates issued once per PAIR of slots (after both slots' instructions), formatted with the even slot."""
    lines = []
    groups = {"slot": 1, "il2": 2, "il4": 4, "instr": 8}[layout]
    for _ in range(reps):
        for g0 in range(0, 8, groups):
            for t in templates:
                for s in range(g0, g0 + groups):
                    lines.append(fmt(t, s))
            if pair:
                for s in range(g0, g0 + groups, 2):
                    for t in pair:
                        lines.append(fmt(t, s))
    return lines


def pure(instr, reps=8):
    return body([instr], "instr", reps)


V = []   # (name, lines, note)


def add(name, lines, note=""):
    V.append((name, lines, note))


# ---------------------------------------------------------------------------------------------------------------
# 1. single instructions by operand kind: is the 2-cycle class a property of the opcode or of where operands come from?
# ---------------------------------------------------------------------------------------------------------------
for nm, ins in [
    ("v_add_u32 vgpr,vgpr", "v_add_u32 {A}, {B}, {A}"),
    ("v_add_u32 sgpr,vgpr", "v_add_u32 {A}, s20, {A}"),
    ("v_add_u32 inline,vgpr", "v_add_u32 {A}, 7, {A}"),
    ("v_add_u32 literal,vgpr", "v_add_u32 {A}, 0x12345, {A}"),
    ("v_sub_u32 vgpr,vgpr", "v_sub_u32 {A}, {B}, {A}"),
    ("v_sub_u32 sgpr,vgpr", "v_sub_u32 {A}, s20, {A}"),
    ("v_subrev_u32 sgpr,vgpr", "v_subrev_u32 {A}, s20, {A}"),
    ("v_and_b32 sgpr,vgpr", "v_and_b32 {A}, s20, {A}"),
    ("v_and_b32 inline,vgpr", "v_and_b32 {A}, 31, {A}"),
    ("v_xor_b32 sgpr,vgpr", "v_xor_b32 {A}, s20, {A}"),
    ("v_mov_b32 vgpr", "v_mov_b32 {A}, {B}"),
    ("v_mov_b32 sgpr", "v_mov_b32 {A}, s20"),
    ("v_not_b32 vgpr", "v_not_b32 {A}, {B}"),
    ("v_xnor_b32 vgpr,vgpr", "v_xnor_b32 {A}, {B}, {A}"),
    ("v_bitop3_b32 vgpr x3", "v_bitop3_b32 {A}, {A}, {B}, {C} bitop3:0x36"),
    ("v_bitop3_b32 inline const", "v_bitop3_b32 {A}, {A}, -8, {C} bitop3:0x36"),
    ("v_bitop3_b32 sgpr", "v_bitop3_b32 {A}, {A}, s20, {C} bitop3:0x36"),
    ("v_ashrrev_i32 imm 31", "v_ashrrev_i32 {A}, 31, {A}"),
    ("v_ashrrev_i32 imm 2", "v_ashrrev_i32 {A}, 2, {A}"),
    ("v_ashrrev_i32 vgpr amount", "v_ashrrev_i32 {A}, {B}, {A}"),
    ("v_lshrrev_b32 vgpr amount", "v_lshrrev_b32 {A}, {B}, {A}"),
    ("v_lshlrev_b32 imm 1", "v_lshlrev_b32 {A}, 1, {A}"),
    ("v_lshlrev_b32 vgpr amount", "v_lshlrev_b32 {A}, {B}, {A}"),
    ("v_fma_f32 vgpr x3", "v_fma_f32 {A}, {A}, {B}, {C}"),
    ("v_fma_f32 sgpr", "v_fma_f32 {A}, s20, {B}, {C}"),
    ("v_fmac_f32 vgpr", "v_fmac_f32 {A}, {B}, {C}"),
    ("v_mul_f32 vgpr", "v_mul_f32 {A}, {B}, {A}"),
    ("v_add_f32 vgpr", "v_add_f32 {A}, {B}, {A}"),
    ("v_max_f32 vgpr", "v_max_f32 {A}, {B}, {A}"),
    ("v_max_i32 vgpr", "v_max_i32 {A}, {B}, {A}"),
    ("v_min_u32 vgpr", "v_min_u32 {A}, {B}, {A}"),
    ("v_med3_i32", "v_med3_i32 {A}, {A}, {B}, {C}"),
    ("v_cndmask_b32 vcc (VOP2)", "v_cndmask_b32 {A}, {B}, {A}, vcc"),
    ("v_cndmask_b32_e64 sgpr pair", "v_cndmask_b32_e64 {A}, {B}, {A}, s[22:23]"),
    ("v_cmp_gt_i32 vcc (VOPC)", "v_cmp_gt_i32 vcc, 0, {A}"),
    ("v_cmp_gt_i32_e64 sgpr pair", "v_cmp_gt_i32_e64 {M}, 0, {A}"),
    ("v_cmp_lt_i32_e64 vgpr,vgpr", "v_cmp_lt_i32_e64 {M}, {A}, {B}"),
    ("v_bcnt_u32_b32 vgpr,vgpr", "v_bcnt_u32_b32 {A}, {B}, {A}"),
    ("v_bcnt_u32_b32 vgpr,0", "v_bcnt_u32_b32 {A}, {B}, 0"),
    ("v_mad_i32_i24 vgpr,-8,vgpr", "v_mad_i32_i24 {A}, {A}, -8, {B}"),
    ("v_mad_i32_i24 vgpr,-8,sgpr", "v_mad_i32_i24 {A}, {A}, -8, s20"),
    ("v_mul_i32_i24 vgpr", "v_mul_i32_i24 {A}, {B}, {A}"),
    ("v_lshl_add_u32", "v_lshl_add_u32 {A}, {A}, 3, {B}"),
    ("v_pk_lshlrev_b16", "v_pk_lshlrev_b16 {A}, {B}, {A}"),
    ("v_pk_add_u16", "v_pk_add_u16 {A}, {A}, {B}"),
    ("v_bfm_b32", "v_bfm_b32 {A}, {B}, {C}"),
    ("v_alignbit_b32 imm", "v_alignbit_b32 {A}, {B}, {A}, 5"),
    ("v_lshrrev_b64 imm", None),      # 64-bit forms need register pairs: generated below
    ("v_add_u32_sdwa word", "v_add_u32_sdwa {A}, {B}, {A} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD"),
    ("v_mov_b32_sdwa byte", "v_mov_b32_sdwa {A}, {B} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1"),
    ("v_and_b32 dpp quad_perm", "v_and_b32_dpp {A}, {B}, {A} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"),
    ("v_accvgpr_write_b32", "v_accvgpr_write_b32 a{slot}, {A}"),
    ("v_accvgpr_read_b32", "v_accvgpr_read_b32 {A}, a{slot}"),
]:
    if ins is None:
        lines = []
        for _ in range(8):
            for p in range(0, 8, 2):
                lines.append("v_lshrrev_b64 v[%d:%d], 5, v[%d:%d]" % (10 + p, 11 + p, 10 + p, 11 + p))
                lines.append("v_lshrrev_b64 v[%d:%d], 5, v[%d:%d]" % (18 + p, 19 + p, 18 + p, 19 + p))
        add("1 " + nm, lines)
    elif "{slot}" in ins:
        lines = []
        for _ in range(8):
            for s in range(8):
                lines.append(fmt(ins.replace("{slot}", str(s)), s))
        add("1 " + nm, lines)
    else:
        add("1 " + nm, pure(ins))

# ---------------------------------------------------------------------------------------------------------------
# 2. two-instruction mixes: do the classes add (2.x + 4.x per pair) or does a 4-cycle neighbour drag the 2-cycle one?
# ---------------------------------------------------------------------------------------------------------------
FAST = "v_add_u32 {A}, {B}, {A}"
FAST2 = "v_xor_b32 {C}, {D}, {C}"
SLOW = "v_bcnt_u32_b32 {E}, {F}, {E}"
SLOW2 = "v_lshlrev_b32 {C}, {D}, {C}"
add("2 F,S alternating, independent registers", body([FAST, SLOW], "slot", 4), "1 fast + 1 slow")
add("2 F,F,S,S per slot, independent", body([FAST, FAST2, SLOW, "v_mad_i32_i24 {D}, {D}, -8, {B}"], "slot", 2), "2 fast + 2 slow")
add("2 eight F then eight S (blocks)", body([FAST, SLOW], "instr", 4), "1 fast + 1 slow")
add("2 F,F,F,S (3:1)", body([FAST, FAST2, "v_sub_u32 {D}, {B}, {D}", SLOW], "slot", 2), "3 fast + 1 slow")
add("2 F,S,S,S (1:3)", body([FAST, SLOW, SLOW2, "v_mad_i32_i24 {D}, {D}, -8, {B}"], "slot", 2), "1 fast + 3 slow")
add("2 F->S dependent chain (S reads F's result)", body(["v_add_u32 {A}, {B}, {A}", "v_bcnt_u32_b32 {A}, {A}, {C}"], "slot", 4), "1 fast + 1 slow, dependent")
add("2 F->F dependent chain", body(["v_add_u32 {A}, {B}, {A}", "v_xor_b32 {A}, {C}, {A}"], "slot", 4), "2 fast, dependent")
add("2 S->S dependent chain", body(["v_bcnt_u32_b32 {A}, {A}, {B}", "v_lshlrev_b32 {A}, {C}, {A}"], "slot", 4), "2 slow, dependent")
add("2 F->F dependent, 8 slots apart", body(["v_add_u32 {A}, {B}, {A}", "v_xor_b32 {A}, {C}, {A}"], "instr", 4), "2 fast, dependent at distance 8")
add("2 S->S dependent, 8 slots apart", body(["v_bcnt_u32_b32 {A}, {A}, {B}", "v_lshlrev_b32 {A}, {C}, {A}"], "instr", 4), "2 slow, dependent at distance 8")
# VGPR bank patterns: the three sources of one instruction from the same bank (index mod 4) or from different ones
add("2 v_bitop3 all sources bank 2 (v10,v14,v18)", ["v_bitop3_b32 v%d, v10, v14, v18 bitop3:0x36" % (26 + (i % 8)) for i in range(64)], "same bank")
add("2 v_bitop3 sources banks 2,3,0 (v10,v15,v20)", ["v_bitop3_b32 v%d, v10, v15, v20 bitop3:0x36" % (26 + (i % 8)) for i in range(64)], "different banks")
add("2 v_mad_i32_i24 sources same bank", ["v_mad_i32_i24 v%d, v10, v14, v18" % (26 + (i % 8)) for i in range(64)], "same bank")
add("2 v_mad_i32_i24 sources different banks", ["v_mad_i32_i24 v%d, v10, v15, v20" % (26 + (i % 8)) for i in range(64)], "different banks")
add("2 v_add_u32 sources same bank", ["v_add_u32 v%d, v10, v14" % (26 + (i % 8)) for i in range(64)], "same bank")
add("2 v_add_u32 sources different banks", ["v_add_u32 v%d, v10, v15" % (26 + (i % 8)) for i in range(64)], "different banks")

# ---------------------------------------------------------------------------------------------------------------
# 3. the row step and candidate replacements, VALU only (entries = whatever the registers hold).
#    A = complemented rank q, B/C = entry {bits, before} (stand-ins: never loaded), D/E = scratch, F = per-lane accumulator
#    s20 = base, s21 = -n0 ; v58 = -n0 in a VGPR ; v59 = row-constant mask
# ---------------------------------------------------------------------------------------------------------------
ADDR_NOW = ["v_ashrrev_i32 {D}, 5, {A}", "v_mad_i32_i24 {D}, {D}, -8, s20"]
TAIL_NOW = ["v_lshlrev_b32 {B}, {A}, {B}", "v_bcnt_u32_b32 {C}, {B}, {C}", "v_cmp_gt_i32_e64 {M}, 0, {B}",
            "v_sub_u32 {B}, s21, {C}", "v_add_u32 {C}, {A}, {C}", "v_cndmask_b32_e64 {A}, {C}, {B}, {M}"]
TAIL_VN0 = [t.replace("s21", "v58") for t in TAIL_NOW]
ADDR_FOLD = ["v_ashrrev_i32 {D}, 2, {A}", "v_bitop3_b32 {D}, {D}, -8, {D} bitop3:0x0c"]      # (~a & b): base folded into the rank
TAIL_SIGN = ["v_lshlrev_b32 {B}, {A}, {B}", "v_bcnt_u32_b32 {C}, {B}, {C}", "v_ashrrev_i32 {E}, 31, {B}",
             "v_sub_u32 {B}, v58, {C}", "v_add_u32 {C}, {A}, {C}", "v_bitop3_b32 {A}, {E}, {B}, {C} bitop3:0xca"]   # E ? B : C
for lay in ("slot", "il2", "instr"):
    add("3 row step as shipped (addr 2 + tail 6, base and -n0 in SGPRs) [%s]" % lay, body(ADDR_NOW + TAIL_NOW, lay, 1), "8 per lookup")
add("3 row step, -n0 in a VGPR [il2]", body(ADDR_NOW + TAIL_VN0, "il2", 1), "8 per lookup")
add("3 row step, -n0 in a VGPR, address = ashr 2 + bitop3 (base folded into the ranks) [il2]", body(ADDR_FOLD + TAIL_VN0, "il2", 1), "8 per lookup")
add("3 ballot-free step: ashr31 + bitop3 select, address folded, joint count per pair of lookups (and + sub) [il2]",
    body(ADDR_FOLD + TAIL_SIGN, "il2", 1, pair=["v_and_b32 {D}, {E}, {E2}", "v_sub_u32 {F}, {F}, {D}"]),
    "8 per lookup + 2 per pair = 9 per lookup")
add("3 ballot-free step, three per-lane counters per pair (plane 0, plane 1, joint: sub, sub, and, sub) [il2]",
    body(ADDR_FOLD + TAIL_SIGN, "il2", 1, pair=["v_sub_u32 {F}, {F}, {E}", "v_sub_u32 {D}, {D}, {E2}", "v_and_b32 {B}, {E}, {E2}", "v_sub_u32 {C}, {C}, {B}"]),
    "8 per lookup + 4 per pair = 10 per lookup")
add("3 ballot-free step with the shipped address (ashr 5 + mad24 sgpr) and joint count [il2]",
    body(ADDR_NOW + TAIL_SIGN, "il2", 1, pair=["v_and_b32 {D}, {E}, {E2}", "v_sub_u32 {F}, {F}, {D}"]), "9 per lookup")
add("3 shipped tail, folded address [il2]", body(ADDR_FOLD + TAIL_NOW, "il2", 1), "8 per lookup")
# v_cmp once per PAIR of lookups on the and of the two shifted words (joint ballot only), selects by sign mask
add("3 sign-mask selects + ONE v_cmp per pair on (t0 & t1) [il2]",
    body(ADDR_FOLD + TAIL_SIGN, "il2", 1, pair=["v_and_b32 {D}, {E}, {E2}", "v_cmp_gt_i32_e64 {M}, 0, {D}"]), "8 per lookup + 2 per pair")


# ---------------------------------------------------------------------------------------------------------------
# 4. block-size sweep: n fast instructions, then n slow ones (all independent): how long must a stretch of fast
#    instructions be before it issues at the fast rate?
# ---------------------------------------------------------------------------------------------------------------
FASTS = ["v_add_u32 {A}, {B}, {A}", "v_xor_b32 {C}, {D}, {C}", "v_sub_u32 {E}, {B}, {E}", "v_and_b32 {F}, {D}, {F}"]
SLOWS = ["v_bcnt_u32_b32 {A}, {B}, {A}", "v_lshlrev_b32 {C}, {D}, {C}", "v_max_i32 {E}, {B}, {E}", "v_mul_i32_i24 {F}, {D}, {F}"]


def blocks(n, total=64):
    lines, k = [], 0
    while len(lines) < 2 * total:
        for kind in (FASTS, SLOWS):
            for i in range(n):
                lines.append(fmt(kind[(k + i) % 4], (k + i) % 8))
            k += n
    return lines


for n in (1, 2, 3, 4, 6, 8, 16, 32, 64):
    add("4 blocks of %d fast then %d slow" % (n, n), blocks(n), "1 fast : 1 slow")
# one slow instruction in a stretch of n fast ones
for n in (3, 7, 15, 31):
    lines = []
    for r in range(128 // (n + 1)):
        for i in range(n):
            lines.append(fmt(FASTS[i % 4], (r + i) % 8))
        lines.append(fmt(SLOWS[r % 4], r % 8))
    add("4 %d fast then 1 slow" % n, lines, "%d fast : 1 slow" % n)
for n in (3, 7, 15):
    lines = []
    for r in range(128 // (n + 1)):
        for i in range(n):
            lines.append(fmt(SLOWS[i % 4], (r + i) % 8))
        lines.append(fmt(FASTS[r % 4], r % 8))
    add("4 %d slow then 1 fast" % n, lines, "1 fast : %d slow" % n)
# does a SALU instruction or an s_nop between fast instructions matter?
add("4 fast stream with one s_nop 0 after every instruction", sum([[fmt(FASTS[i % 4], i % 8), "s_nop 0"] for i in range(64)], []), "64 VALU")
add("4 fast stream with one SALU (s_add_u32) after every instruction", sum([[fmt(FASTS[i % 4], i % 8), "s_add_u32 s24, s24, s25"] for i in range(64)], []), "64 VALU")
add("4 slow stream with one SALU (s_add_u32) after every instruction", sum([[fmt(SLOWS[i % 4], i % 8), "s_add_u32 s24, s24, s25"] for i in range(64)], []), "64 VALU")
# S sub-classes
add("4 v_cmp_gt_i32_e64 + v_cndmask_b32_e64 pairs, adjacent", sum([[fmt("v_cmp_gt_i32_e64 {M}, 0, {A}", i % 8), fmt("v_cndmask_b32_e64 {B}, {C}, {D}, {M}", i % 8)] for i in range(32)], []), "64 VALU")
add("4 v_cmp_gt_i32_e64 x8 then v_cndmask_b32_e64 x8", body(["v_cmp_gt_i32_e64 {M}, 0, {A}", "v_cndmask_b32_e64 {B}, {C}, {D}, {M}"], "instr", 4), "64 VALU")

# ---------------------------------------------------------------------------------------------------------------
# 5. two instruction streams on ONE SIMD: waves whose (wave >> 2) is even run stream X, the others stream Y (a workgroup's
#    waves go to the SIMDs cyclically, so every SIMD hosts both kinds).  Cycles are reported per kind: out[0] = X, out[3] = Y.
# ---------------------------------------------------------------------------------------------------------------
SPLIT = []   # (name, linesX, linesY, note)
PF = [fmt(FASTS[i % 4], i % 8) for i in range(64)]
PS = [fmt(SLOWS[i % 4], i % 8) for i in range(64)]
ALT = sum([[fmt(FASTS[i % 4], i % 8), fmt(SLOWS[i % 4], i % 8)] for i in range(32)], [])
SPLIT.append(("5 half the waves of a SIMD pure fast | the other half pure slow", PF, PS, "X fast, Y slow"))
SPLIT.append(("5 half pure fast | half alternating fast,slow", PF, ALT, "X fast, Y alternating"))
SPLIT.append(("5 half pure slow | half alternating fast,slow", PS, ALT, "X slow, Y alternating"))
SPLIT.append(("5 half pure fast | half pure fast (control)", PF, PF, "both fast"))

# ---------------------------------------------------------------------------------------------------------------
# 6 / 7. the row step WITH its LDS gathers on a valid random directory (true LF-mapping, ranks stay in the row), candidate
#    forms and layouts.  Registers: q = v10+s (s = 0..NL-1 lookups in flight), entry pair v[30+2s : 31+2s], sign mask v70+s,
#    accumulators v90..v93; constants: s20 = base - 8, s21 = -n0 (shipped form);  v94 = N' = -n0 - c (folded form, c = 4 * row address)
#    Bank rule found in pass 1: a three-source instruction whose sources share a VGPR bank (index mod 4) issues slow.
# ---------------------------------------------------------------------------------------------------------------
def q(s): return "v%d" % (10 + s)
def elo(s): return "v%d" % (30 + 2 * s)
def ehi(s): return "v%d" % (31 + 2 * s)
def epair(s): return "v[%d:%d]" % (30 + 2 * s, 31 + 2 * s)
def sg(s): return "v%d" % (70 + s)
def mk(s): return "s[%d:%d]" % (36 + 2 * (s % 8), 37 + 2 * (s % 8))


def addr_shipped(s): return ["v_ashrrev_i32 %s, 5, %s" % (elo(s), q(s)), "v_mad_i32_i24 %s, %s, -8, s20" % (elo(s), elo(s))]
def addr_folded(s): return ["v_ashrrev_i32 %s, 2, %s" % (elo(s), q(s)), "v_bitop3_b32 %s, %s, -8, 0 bitop3:0x0c" % (elo(s), elo(s))]
def rd(s): return ["ds_read_b64 %s, %s" % (epair(s), elo(s))]
COUNT3 = ["s_bcnt1_i32_b64 vcc_lo, {M0}", "s_add_u32 s24, s24, vcc_lo", "s_bcnt1_i32_b64 vcc_lo, {M1}", "s_add_u32 s25, s25, vcc_lo",
          "s_and_b64 vcc, {M0}, {M1}", "s_bcnt1_i32_b64 vcc_lo, vcc", "s_add_u32 s26, s26, vcc_lo"]


def tail_ops(kind):
    """list of per-lookup instruction templates (callables of s) in dependency order"""
    if kind == "shipped":
        return [lambda s: "v_lshlrev_b32 %s, %s, %s" % (elo(s), q(s), elo(s)),
                lambda s: "v_bcnt_u32_b32 %s, %s, %s" % (ehi(s), elo(s), ehi(s)),
                lambda s: "v_cmp_gt_i32_e64 %s, 0, %s" % (mk(s), elo(s)),
                lambda s: "v_sub_u32 %s, s21, %s" % (elo(s), ehi(s)),
                lambda s: "v_add_u32 %s, %s, %s" % (ehi(s), q(s), ehi(s)),
                lambda s: "v_cndmask_b32_e64 %s, %s, %s, %s" % (q(s), ehi(s), elo(s), mk(s))]
    if kind == "vn0":      # -n0 (folded: N') in a VGPR
        t = tail_ops("shipped")
        t[3] = lambda s: "v_sub_u32 %s, v94, %s" % (elo(s), ehi(s))
        return t
    if kind == "sign":     # no ballot: sign mask + bitop3 select
        return [lambda s: "v_lshlrev_b32 %s, %s, %s" % (elo(s), q(s), elo(s)),
                lambda s: "v_bcnt_u32_b32 %s, %s, %s" % (ehi(s), elo(s), ehi(s)),
                lambda s: "v_ashrrev_i32 %s, 31, %s" % (sg(s), elo(s)),
                lambda s: "v_sub_u32 %s, v94, %s" % (elo(s), ehi(s)),
                lambda s: "v_add_u32 %s, %s, %s" % (ehi(s), q(s), ehi(s)),
                # sources: sign v70+s (bank (2+s)%4), c1 v30+2s (bank (2+2s)%4), c0 v31+2s (bank (3+2s)%4): distinct unless s%4 == 0 -> see sg_alt
                lambda s: "v_bitop3_b32 %s, %s, %s, %s bitop3:0xca" % (q(s), sgb(s), elo(s), ehi(s))]
    if kind == "signcmp":  # ballot kept (v_cmp), select by sign mask
        t = tail_ops("sign")
        t.insert(3, lambda s: "v_cmp_gt_i32_e64 %s, 0, %s" % (mk(s), elo(s)))
        return t
    raise ValueError(kind)


# the sign mask register is chosen per lookup so that the select's three sources sit in three different banks
SG_BASE = [72]          # first sign register: behind the entry pairs (v30 .. v30 + 2 nl - 1), a multiple of 4


def sgb(s):
    # entry pair v30+2s / v31+2s: banks {2,3} for even s, {0,1} for odd s -> sign in bank 0/1 for even s, 2/3 for odd s
    r = SG_BASE[0] + (s & ~3) + {0: 0, 1: 2, 2: 1, 3: 3}[s & 3]
    assert r % 4 not in ((30 + 2 * s) % 4, (31 + 2 * s) % 4) and r < 94, (s, r)
    return "v%d" % r


def sg(s): return sgb(s)


def step(nl, addr, tail, layout, joint=None, counts=False, waits="all"):
    """nl lookups in flight (pairs = the two planes of a column).  layout 'pair': tails of a column's two lookups interleaved
    (the shipped BGTH_TAIL2); 'instr': instruction-major over all nl lookups."""
    L = ["s_waitcnt lgkmcnt(0)"]
    A = [addr(s) for s in range(nl)]
    for k in range(2):
        for s in range(nl):
            L.append(A[s][k])
    for s in range(nl):
        L += rd(s)
    T = tail_ops(tail)
    if layout == "instr":
        L.append("s_waitcnt lgkmcnt(0)")
        for t in T:
            for s in range(nl):
                L.append(t(s))
        for s in range(0, nl, 2):
            if joint == "and_sub":
                L += ["v_and_b32 %s, %s, %s" % (elo(s), sg(s), sg(s + 1)), "v_sub_u32 v%d, v%d, %s" % (100 + (s // 2) % 4, 100 + (s // 2) % 4, elo(s))]
            if joint == "three":
                L += ["v_sub_u32 v100, v100, %s" % sg(s), "v_sub_u32 v101, v101, %s" % sg(s + 1),
                      "v_and_b32 %s, %s, %s" % (elo(s), sg(s), sg(s + 1)), "v_sub_u32 v102, v102, %s" % elo(s)]
            if counts:
                L += [c.replace("{M0}", mk(s)).replace("{M1}", mk(s + 1)) for c in COUNT3]
    else:
        for s in range(0, nl, 2):
            L.append("s_waitcnt lgkmcnt(%d)" % (nl - s - 2))
            for t in T:
                L.append(t(s)); L.append(t(s + 1))
            if joint == "and_sub":
                L += ["v_and_b32 %s, %s, %s" % (elo(s), sg(s), sg(s + 1)), "v_sub_u32 v%d, v%d, %s" % (100 + (s // 2) % 4, 100 + (s // 2) % 4, elo(s))]
            if joint == "three":
                L += ["v_sub_u32 v100, v100, %s" % sg(s), "v_sub_u32 v101, v101, %s" % sg(s + 1),
                      "v_and_b32 %s, %s, %s" % (elo(s), sg(s), sg(s + 1)), "v_sub_u32 v102, v102, %s" % elo(s)]
            if counts:
                L += [c.replace("{M0}", mk(s)).replace("{M1}", mk(s + 1)) for c in COUNT3]
    return L


LDSV = []   # (name, lines, note, nl, folded)


def addl(name, lines, note, nl, folded):
    LDSV.append((name, lines, note, nl, folded))


for nl in (8, 16):
    addl("7 shipped step (ashr5 + mad24 sgpr | lshl bcnt cmp sub(sgpr) add cndmask + SALU counts), tails in pairs, %d lookups in flight" % nl,
         step(nl, addr_shipped, "shipped", "pair", counts=True), "8 VALU per lookup", nl, False)
    addl("7 shipped instructions, tails instruction-major over %d lookups" % nl,
         step(nl, addr_shipped, "shipped", "instr", counts=True), "8 VALU per lookup", nl, False)
    addl("7 folded address (ashr2 + bitop3) + N' in a VGPR, ballot kept (cmp + cndmask + SALU counts), tails in pairs, %d in flight" % nl,
         step(nl, addr_folded, "vn0", "pair", counts=True), "8 VALU per lookup", nl, True)
    addl("7 folded address + N' in a VGPR, ballot kept, instruction-major over %d" % nl,
         step(nl, addr_folded, "vn0", "instr", counts=True), "8 VALU per lookup", nl, True)
    addl("7 folded address, ballot kept (cmp) but select by sign mask (ashr31 + bitop3), instruction-major over %d" % nl,
         step(nl, addr_folded, "signcmp", "instr", counts=True), "9 VALU per lookup", nl, True)
    addl("7 ballot-free: folded address, sign-mask select, joint count per column (and + sub), tails in pairs, %d in flight" % nl,
         step(nl, addr_folded, "sign", "pair", joint="and_sub"), "9 VALU per lookup", nl, True)
    addl("7 ballot-free, joint count, instruction-major over %d" % nl,
         step(nl, addr_folded, "sign", "instr", joint="and_sub"), "9 VALU per lookup", nl, True)
    addl("7 ballot-free, three per-lane counters per column, instruction-major over %d" % nl,
         step(nl, addr_folded, "sign", "instr", joint="three"), "10 VALU per lookup", nl, True)


# ---------------------------------------------------------------------------------------------------------------
# 8. pass 3: what exactly breaks the fast rate?  31 fast instructions + ONE instruction X per 32 (all independent)
# ---------------------------------------------------------------------------------------------------------------
F4 = ["v_add_u32 {A}, {B}, {A}", "v_sub_u32 {C}, {D}, {C}", "v_ashrrev_i32 {E}, 31, {E}", "v_and_b32 {F}, {D}, {F}"]
for nm, x in [("nothing (32 fast)", "v_add_u32 {A}, {B}, {A}"), ("v_bcnt_u32_b32", "v_bcnt_u32_b32 {A}, {B}, {A}"), ("v_lshlrev_b32 vgpr", "v_lshlrev_b32 {C}, {D}, {C}"),
              ("v_max_i32", "v_max_i32 {E}, {B}, {E}"), ("v_mul_i32_i24", "v_mul_i32_i24 {F}, {D}, {F}"), ("v_cmp_gt_i32_e64", "v_cmp_gt_i32_e64 {M}, 0, {A}"),
              ("v_cndmask_b32_e64", "v_cndmask_b32_e64 {A}, {B}, {A}, s[22:23]"), ("v_mad_i32_i24", "v_mad_i32_i24 {A}, {A}, -8, {B}"),
              ("v_add_u32 with an SGPR source", "v_add_u32 {A}, s20, {A}"), ("v_readlane_b32", "v_readlane_b32 s24, {A}, 3"),
              ("v_mov_b32 dpp", "v_mov_b32_dpp {A}, {B} row_shr:1 row_mask:0xf bank_mask:0xf"), ("s_add_u32 (SALU)", "s_add_u32 s24, s24, s25"),
              ("ds_read_b32 (+ wait)", "ds_read_b32 {A}, {B}\n\ts_waitcnt lgkmcnt(0)")]:
    lines = []
    for r in range(4):
        for k in range(31):
            lines.append(fmt(F4[k % 4], (r + k) % 8))
        lines += fmt(x, r % 8).split("\n\t")
    add("8 31 fast + 1 x %s" % nm, lines, "per 32")
S2 = ["v_lshlrev_b32 {A}, {B}, {A}", "v_bcnt_u32_b32 {C}, {D}, {C}"]
for n in (1, 4, 16, 64):
    lines, k = [], 0
    while len(lines) < 128:
        for i_ in range(n):
            lines.append(fmt(F4[(k + i_) % 4], (k + i_) % 8))
        for i_ in range(n):
            lines.append(fmt(S2[(k + i_) % 2], (k + i_) % 8))
        k += n
    add("8 blocks of %d fast then %d of {lshl, bcnt}" % (n, n), lines, "1 fast : 1 slow")

# VALU-only timing of the two 2-instruction address forms (pass 1 ran the bitop3 with one register as all three sources)
add("4 address: v_ashrrev 2 + v_bitop3 (~a & -8), third source an inline constant", body(["v_ashrrev_i32 {D}, 2, {A}", "v_bitop3_b32 {D}, {D}, -8, 0 bitop3:0x0c"], "slot", 4), "2 per lookup")
add("4 address: v_ashrrev 2 + v_and_b32 -8 (descending table)", body(["v_ashrrev_i32 {D}, 2, {A}", "v_and_b32 {D}, -8, {D}"], "slot", 4), "2 per lookup")
add("4 select: v_bitop3 a ? b : c, sources in three banks", ["v_bitop3_b32 v%d, v%d, v%d, v%d bitop3:0xca" % (10 + (i % 8), 72 + 2 * (i % 8), 30 + 2 * (i % 8), 31 + 2 * (i % 8)) for i in range(64)], "1")


# ---- pass 3: more forms of the ballot-free step on the real directory
def step3(nl, form, halves=False):
    SG_BASE[0] = 72
    if nl > 20: raise ValueError("more than 20 lookups do not fit this register map (pairs v30.., signs v72.., accumulators v90..)")
    """form: 'joint' (9 VALU), 'bits' (per-lane bit accumulators of both planes: acc = 2 acc - sign, 10 VALU),
    'and' (address by v_and on a descending table, joint), 'lshr' (words bit-reversed: variable shift right is a fast
    instruction; mask = 0 - (t & 1)), 'cmpsel' (cmp + cndmask, no SALU counts)"""
    L = ["s_waitcnt lgkmcnt(0)"]
    for s_ in range(nl):
        L.append("v_ashrrev_i32 %s, 2, %s" % (elo(s_), q(s_)))
    for s_ in range(nl):
        if form == "and":
            L.append("v_and_b32 %s, -8, %s" % (elo(s_), elo(s_)))
        else:
            L.append("v_bitop3_b32 %s, %s, -8, 0 bitop3:0x0c" % (elo(s_), elo(s_)))
    for s_ in range(nl):
        L += rd(s_)
    groups = [range(0, nl // 2), range(nl // 2, nl)] if halves else [range(nl)]
    for gi, g in enumerate(groups):
        L.append("s_waitcnt lgkmcnt(%d)" % (nl - g[-1] - 1))
        if form == "lshr":
            for s_ in g: L.append("v_lshrrev_b32 %s, %s, %s" % (elo(s_), q(s_), elo(s_)))
            for s_ in g: L.append("v_bcnt_u32_b32 %s, %s, %s" % (ehi(s_), elo(s_), ehi(s_)))
            for s_ in g: L.append("v_and_b32 %s, 1, %s" % (sg(s_), elo(s_)))
            for s_ in g: L.append("v_sub_u32 %s, 0, %s" % (sg(s_), sg(s_)))
        else:
            for s_ in g: L.append("v_lshlrev_b32 %s, %s, %s" % (elo(s_), q(s_), elo(s_)))
            for s_ in g: L.append("v_bcnt_u32_b32 %s, %s, %s" % (ehi(s_), elo(s_), ehi(s_)))
            if form == "cmpsel":
                for s_ in g: L.append("v_cmp_gt_i32_e64 %s, 0, %s" % (mk(s_), elo(s_)))
            else:
                for s_ in g: L.append("v_ashrrev_i32 %s, 31, %s" % (sg(s_), elo(s_)))
        for s_ in g: L.append("v_sub_u32 %s, v94, %s" % (elo(s_), ehi(s_)))
        for s_ in g: L.append("v_add_u32 %s, %s, %s" % (ehi(s_), q(s_), ehi(s_)))
        if form == "cmpsel":
            for s_ in g: L.append("v_cndmask_b32_e64 %s, %s, %s, %s" % (q(s_), ehi(s_), elo(s_), mk(s_)))
        else:
            for s_ in g: L.append("v_bitop3_b32 %s, %s, %s, %s bitop3:0xca" % (q(s_), sg(s_), elo(s_), ehi(s_)))
        if form == "bits":
            for s_ in g: L.append("v_add_u32 v%d, v%d, v%d" % (100 + s_ % 2, 100 + s_ % 2, 100 + s_ % 2))
            for s_ in g: L.append("v_sub_u32 v%d, v%d, %s" % (100 + s_ % 2, 100 + s_ % 2, sg(s_)))
        elif form != "cmpsel":
            for s_ in g:
                if s_ % 2 == 0: L.append("v_and_b32 %s, %s, %s" % (elo(s_), sg(s_), sg(s_ + 1)))
            for s_ in g:
                if s_ % 2 == 0: L.append("v_sub_u32 v%d, v%d, %s" % (100 + (s_ // 2) % 4, 100 + (s_ // 2) % 4, elo(s_)))
    return L


for nl in (16, 20):
    addl("9 ballot-free, joint count, instruction-major over %d" % nl, step3(nl, "joint"), "9 VALU per lookup", nl, "fold")
    addl("9 ballot-free, per-lane bit accumulators of both planes (acc = 2 acc - sign), instruction-major over %d" % nl, step3(nl, "bits"), "10 VALU per lookup", nl, "fold")
addl("9 ballot-free, joint count, two halves of 8 (second half's reads in flight during the first half's tails)", step3(16, "joint", halves=True), "9 VALU per lookup", 16, "fold")
addl("9 ballot-free, joint count, 12 lookups in flight", step3(12, "joint"), "9 VALU per lookup", 12, "fold")
addl("9 ballot-free, per-lane bit accumulators, two halves of 8", step3(16, "bits", halves=True), "10 VALU per lookup", 16, "fold")
addl("9 ballot-free, joint count, address by v_and on a descending table, instruction-major over 16", step3(16, "and"), "9 VALU per lookup", 16, "desc")
addl("9 ballot-free, bit-reversed words + v_lshrrev (fast) + mask = 0 - (t & 1), joint count, instruction-major over 16", step3(16, "lshr"), "10 VALU per lookup", 16, "foldrev")
addl("9 v_cmp + v_cndmask (ballot kept) but NO SALU counts, folded address, instruction-major over 16", step3(16, "cmpsel"), "8 VALU per lookup", 16, "fold")


def emit():
    o = []
    w = o.append
    clob = CLOBBER + ["v%d" % i for i in range(58, 106)] + ["s24", "s25", "s26"]
    clob = sorted(set(clob), key=lambda x: (x[0], int(x[1:]) if x[1:].isdigit() else 999))
    w("// GENERATED by bgt_amd/csrc/gen/gen_issue_bench.py -- do not edit.  VALU issue experiments (profiles/r04_issue/): every kernel")
    w("// runs one fixed list of instructions in a loop; s_memtime brackets the loop.  Measurement tool, part of")
    w("// libbgt_hip_bench.so, not of the product library.")
    w("#include <hip/hip_runtime.h>")
    w("#include <stdint.h>")
    w("namespace bgth_issue {")
    w("#define BGTH_IB_CLOBBER " + ", ".join('"%s"' % c for c in clob))
    w("__device__ __forceinline__ void ib_setup()")
    w("{")
    w("    asm volatile(")
    for i in range(10, 106):
        w('        "v_mov_b32 v%d, %d\\n\\t"' % (i, (i * 7) % 61))
    w('        "s_mov_b32 s20, 0x4000\\n\\ts_mov_b32 s21, -77\\n\\ts_mov_b64 s[22:23], 0x5555\\n\\ts_mov_b64 vcc, 0x3333\\n\\ts_mov_b32 s24, 0\\n\\ts_mov_b32 s25, 1\\n\\ts_mov_b32 s26, 0\\n\\t"')
    w("        ::: BGTH_IB_CLOBBER);")
    w("}")

    def asm_block(lines, indent="        "):
        w(indent + "asm volatile(")
        for ln in lines:
            w(indent + '    "%s\\n\\t"' % ln)
        w(indent + "    ::: BGTH_IB_CLOBBER, \"memory\");")

    infos = []
    for k, (name, lines, note) in enumerate(V):
        w("// %d: %s" % (k, name))
        w("__global__ void __launch_bounds__(1024) ib_kernel_%d(int iters, unsigned long long *cycles, uint32_t *sink, uint32_t seed)" % k)
        w("{")
        w("    ib_setup();")
        w("    const unsigned long long t0 = __builtin_amdgcn_s_memtime();")
        w("    for (int it = 0; it < iters; ++it) {")
        asm_block(lines)
        w("    }")
        w("    const unsigned long long t1 = __builtin_amdgcn_s_memtime();")
        w("    uint32_t acc;")
        w('    asm volatile("v_xor_b32 %0, v10, v17" : "=v"(acc) :: BGTH_IB_CLOBBER);')
        w("    if ((threadIdx.x & 63) == 0) atomicMax(cycles, t1 - t0);")
        w("    if (acc == 0x9e3779b9u) *sink = acc;")
        w("}")
        nv = sum(1 for ln in lines if ln.startswith("v_"))
        infos.append((name, note, nv, "ib_kernel_%d" % k))
    for k, (name, lx, ly, note) in enumerate(SPLIT):
        w("// split %d: %s" % (k, name))
        w("__global__ void __launch_bounds__(1024) ibs_kernel_%d(int iters, unsigned long long *cycles, uint32_t *sink, uint32_t seed)" % k)
        w("{")
        w("    ib_setup();")
        w("    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) & 1;")
        w("    const unsigned long long t0 = __builtin_amdgcn_s_memtime();")
        w("    if (role == 0) {")
        w("        for (int it = 0; it < iters; ++it) {")
        asm_block(lx, "            ")
        w("        }")
        w("    } else {")
        w("        for (int it = 0; it < iters; ++it) {")
        asm_block(ly, "            ")
        w("        }")
        w("    }")
        w("    const unsigned long long t1 = __builtin_amdgcn_s_memtime();")
        w("    uint32_t acc;")
        w('    asm volatile("v_xor_b32 %0, v10, v17" : "=v"(acc) :: BGTH_IB_CLOBBER);')
        w("    if ((threadIdx.x & 63) == 0) atomicMax(cycles + role, t1 - t0);")
        w("    if (acc == 0x9e3779b9u) *sink = acc;")
        w("}")
        assert len([l for l in lx if l.startswith("v_")]) == len([l for l in ly if l.startswith("v_")])
        infos.append((name, note, sum(1 for ln in lx if ln.startswith("v_")), "ibs_kernel_%d" % k))
    for k, (name, lines, note, nl, folded) in enumerate(LDSV):
        w("// lds %d: %s" % (k, name))
        w("__global__ void __launch_bounds__(1024) ibl_kernel_%d(int iters, unsigned long long *cycles, uint32_t *sink, uint32_t seed)" % k)
        w("{")
        w("    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];")
        w("    uint2 *tab = reinterpret_cast<uint2*>(smem);")
        w("    constexpr int NE = 4096;                          // 32 KB of {bits, ones before}: a row of 131,072 columns")
        mode = {True: "fold", False: "plain"}.get(folded, folded)
        desc = mode == "desc"
        w("    // word w of the row sits in entry %s" % ("NE - 1 - w (descending table)" if desc else "w"))
        w("    for (int i = threadIdx.x; i < NE; i += blockDim.x) tab[%s] = make_uint2(0x9e3779b9u * (uint32_t)(i + 1) * (uint32_t)(i + 7), 0u);" % ("NE - 1 - i" if desc else "i"))
        w("    __syncthreads();")
        w("    if (threadIdx.x == 0) {")
        w("        uint32_t run = 0;")
        w("        for (int i = 0; i < NE; ++i) { uint2 &e = tab[%s]; e.y = run; run += (uint32_t)__popc(e.x); %s }" % ("NE - 1 - i" if desc else "i", "e.x = __brev(e.x);" if mode == "foldrev" else ""))
        w("        tab[NE] = make_uint2((uint32_t)NE * 32u - run, 0u);")
        w("    }")
        w("    __syncthreads();")
        w("    const uint32_t rowaddr = __builtin_amdgcn_groupstaticsize();")
        w("    const uint32_t n0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)tab[NE].x);")
        if desc:      # entry of word w at TOP - 8 w, TOP = rowaddr + 8 (NE - 1); the registers hold q + c', c' = 4 (TOP + 8)
            w("    const uint32_t c = 0u - 4u * (rowaddr + 8u * (uint32_t)NE);")
        else:
            w("    const uint32_t c = %s;" % ("0u" if mode == "plain" else "4u * rowaddr"))
        w("    uint32_t qv[%d];" % nl)
        w("    for (int s = 0; s < %d; ++s) qv[s] = ~(((((uint32_t)(threadIdx.x * %d + s) * 2654435761u) ^ seed) %% (NE * 32u)) + c);" % (nl, nl))
        w("    ib_setup();")
        w("    asm volatile(")
        for s_ in range(nl):
            w('        "v_mov_b32 v%d, %%%d\\n\\t"' % (10 + s_, s_))
        w('        "s_mov_b32 s20, %%%d\\n\\ts_mov_b32 s21, %%%d\\n\\tv_mov_b32 v94, %%%d\\n\\t"' % (nl, nl + 1, nl + 2))
        w('        "v_mov_b32 v100, 0\\n\\tv_mov_b32 v101, 0\\n\\tv_mov_b32 v102, 0\\n\\tv_mov_b32 v103, 0\\n\\ts_mov_b32 s24, 0\\n\\ts_mov_b32 s25, 0\\n\\ts_mov_b32 s26, 0\\n\\t"')
        w("        :: " + ", ".join('"v"(qv[%d])' % s_ for s_ in range(nl)) + ', "s"(rowaddr - 8u), "s"(0u - n0), "v"(0u - n0 - c)')
        w("        : BGTH_IB_CLOBBER);")
        w("    const unsigned long long t0 = __builtin_amdgcn_s_memtime();")
        w("    for (int it = 0; it < iters; ++it) {")
        asm_block(lines)
        w("    }")
        w("    const unsigned long long t1 = __builtin_amdgcn_s_memtime();")
        w("    uint32_t q0, qn;")
        w('    asm volatile("s_waitcnt lgkmcnt(0)\\n\\tv_mov_b32 %%0, v10\\n\\tv_mov_b32 %%1, v%d" : "=v"(q0), "=v"(qn) :: BGTH_IB_CLOBBER);' % (10 + nl - 1))
        w("    if ((threadIdx.x & 63) == 0) atomicMax(cycles, t1 - t0);")
        w("    // the ranks must still lie inside the row: a broken step shows here")
        w("    if (~q0 - c >= NE * 32u || ~qn - c >= NE * 32u) atomicAdd(sink + 1, 1u);")
        w("    if ((q0 ^ qn) == 0x9e3779b9u) *sink = q0;")
        w("}")
        nv = sum(1 for ln in lines if ln.startswith("v_"))
        infos.append((name, note + ", %d ds_read_b64" % nl, nv, "ibl_kernel_%d" % k))
    w("struct IbInfo { const char *name; const char *note; int valu_per_iter; void (*fn)(int, unsigned long long*, uint32_t*, uint32_t); };")
    w("static const IbInfo kIb[] = {")
    for (name, note, nv, fn) in infos:
        w('    {"%s", "%s", %d, %s},' % (name.replace('"', "'"), note, nv, fn))
    w("};")
    w("constexpr int kNumIb = %d;" % len(infos))
    w("}  // namespace bgth_issue")
    w("")
    w('extern "C" int bgth_issue_bench_count(void) { return bgth_issue::kNumIb; }')
    w('extern "C" const char *bgth_issue_bench_name(int i) { return i >= 0 && i < bgth_issue::kNumIb ? bgth_issue::kIb[i].name : nullptr; }')
    w('extern "C" const char *bgth_issue_bench_note(int i) { return i >= 0 && i < bgth_issue::kNumIb ? bgth_issue::kIb[i].note : nullptr; }')
    w("// waves_per_simd 1, 2, 4: one workgroup of 256 * w threads per CU; 8: two workgroups of 1024 threads per CU.")
    w("// out[0] = shader cycles of the slowest wave (of the waves running stream X in the two-stream experiments), out[1] = ms of")
    w("// the launch, out[2] = VALU wave-instructions per wave, out[3] = cycles of the slowest wave running stream Y (else 0),")
    w("// out[4] = waves whose ranks left the row (must be 0)")
    w('extern "C" int bgth_issue_bench_run(int device, int i, int waves_per_simd, int iters, double out[5])')
    w("{")
    w("    using namespace bgth_issue;")
    w("    if (i < 0 || i >= kNumIb || hipSetDevice(device) != hipSuccess) return -1;")
    w("    if (waves_per_simd != 1 && waves_per_simd != 2 && waves_per_simd != 4 && waves_per_simd != 8) return -1;")
    w("    hipDeviceProp_t prop;")
    w("    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -1;")
    w("    const int cus = prop.multiProcessorCount;")
    w("    const int threads = waves_per_simd == 8 ? 1024 : 256 * waves_per_simd, grid = waves_per_simd == 8 ? 2 * cus : cus;")
    w("    // one workgroup per CU needs more than half the LDS; two per CU at most a third")
    w("    const int lds = waves_per_simd == 8 ? 48 * 1024 : 96 * 1024;")
    w("    unsigned long long *cyc = nullptr; uint32_t *sink = nullptr;")
    w("    if (hipMalloc((void**)&cyc, 16) != hipSuccess || hipMalloc((void**)&sink, 8) != hipSuccess) return -1;")
    w("    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);")
    w("    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kIb[i].fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds);")
    w("    float ms = 0; hipError_t e = hipSuccess;")
    w("    for (int pass = 0; pass < 2 && e == hipSuccess; ++pass) {")
    w("        (void)hipMemset(cyc, 0, 16); (void)hipMemset(sink, 0, 8);")
    w("        (void)hipEventRecord(e0, nullptr);")
    w("        hipLaunchKernelGGL(kIb[i].fn, dim3(grid), dim3(threads), lds, nullptr, iters, cyc, sink, 0x5bd1e995u);")
    w("        e = hipGetLastError();")
    w("        (void)hipEventRecord(e1, nullptr);")
    w("        if (e == hipSuccess) e = hipEventSynchronize(e1);")
    w("        if (e == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);")
    w("    }")
    w("    unsigned long long h[2] = {0, 0}; uint32_t sk[2] = {0, 0};")
    w("    if (e == hipSuccess) e = hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);")
    w("    if (e == hipSuccess) e = hipMemcpy(sk, sink, 8, hipMemcpyDeviceToHost);")
    w("    out[0] = (double)h[0]; out[1] = ms; out[2] = (double)kIb[i].valu_per_iter * iters; out[3] = (double)h[1]; out[4] = (double)sk[1];")
    w("    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(cyc); (void)hipFree(sink);")
    w("    return e == hipSuccess ? 0 : -1;")
    w("}")
    open(OUT, "w").write("\n".join(o) + "\n")
    print("wrote %s: %d kernels" % (OUT, len(infos)))


if __name__ == "__main__":
    emit()
