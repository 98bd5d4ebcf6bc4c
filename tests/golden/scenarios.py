"""Shared definitions of the codec-level golden cases.

Used by make_golden.py (which replays every scenario on the COMPILED REFERENCE, oracle/_ref/libbgt_ref.so)
and by the tests (which replay them on the oracle restatement and on the HIP path).
Inputs are seeded numpy draws; nothing here comes from the reference.
"""
import numpy as np


def ld_matrix(rng, rows, m, n_founders=6, switch=0.05, p_missing=0.02, p_multi=0.03):
    """Genotype-code matrix (rows, m) with haplotype-block structure so that PBWT runs form:
    columns are mosaics of a few founder haplotypes; a sprinkle of code 2 (missing) and 3 (<M>)."""
    founders = (rng.random((rows, n_founders)) < rng.random((rows, 1)) * 0.6).astype(np.uint8)
    src = rng.integers(0, n_founders, m)
    out = np.zeros((rows, m), np.uint8)
    for r in range(rows):
        sw = rng.random(m) < switch
        src = np.where(sw, rng.integers(0, n_founders, m), src)
        out[r] = founders[r, src]
    out[rng.random((rows, m)) < p_missing] = 2
    multi_rows = rng.random(rows) < 0.3
    mask = (rng.random((rows, m)) < p_multi) & multi_rows[:, None]
    out[mask] = 3
    return out


def long_run_matrix(m=70000):
    """Rows whose PBWT-order bit strings hit every branch of the run-length code: constant rows, a
    single 1, and run lengths straddling each hex-digit boundary (15/16, 255/256, 4095/4096,
    65535/65536).  Row 0 is decoded with the identity permutation, so its runs are literal; later
    rows exercise the same lengths through a non-trivial permutation."""
    rows = []

    def from_runs(runs, first_bit=0):
        v = np.zeros(m, np.uint8)
        at, b = 0, first_bit
        for l in runs:
            v[at:at + l] = b
            at += l
            b ^= 1
        v[at:] = b
        return v

    rows.append(from_runs([15, 16, 255, 256, 4095, 4096, 1, 17, 0x111]))
    rows.append(np.zeros(m, np.uint8))
    rows.append(np.ones(m, np.uint8))
    rows.append(from_runs([65535, 1]))
    rows.append(from_runs([65536, 3], first_bit=1))
    one = np.zeros(m, np.uint8); one[0] = 1; rows.append(one)
    one = np.zeros(m, np.uint8); one[m - 1] = 1; rows.append(one)
    one = np.zeros(m, np.uint8); one[31337] = 1; rows.append(one)
    rows.append(from_runs([0x10, 0x100, 0x1000, 0x10000, 0xf, 0xff, 0xfff]))
    rows.append(from_runs([1] * 400))
    rows.append((np.arange(m) % 2).astype(np.uint8))          # worst case: m one-symbol runs
    p0 = np.stack(rows)
    # plane 1: shifted variants so the two planes differ
    p1 = np.roll(p0, 4321, axis=1)[::-1].copy()
    return (p0 | (p1 << 1)).astype(np.uint8)


def cases():
    """name -> (matrix, shift)"""
    rng = np.random.default_rng(20240928)
    out = {}
    ex1 = np.array([[0, 1, 2, 0], [2, 0, 1, 1], [1, 0, 1, 1], [0, 1, 0, 1],
                    [1, 2, 0, 0], [1, 0, 1, 2], [0, 1, 1, 1]], np.uint8)   # = reference ex1.pim
    out["ex1"] = (ex1, 13)
    out["blocks"] = (ld_matrix(rng, 150, 37), 4)
    out["wide"] = (ld_matrix(rng, 70, 300, n_founders=10), 5)
    out["odd"] = (ld_matrix(rng, 33, 65, n_founders=3, switch=0.2), 3)
    out["longrun"] = (long_run_matrix(), 13)
    return out


def scenarios(name, rows, m, shift):
    """List of op lists. ops: ('subset', [cols]) ('seek', k) ('read', n)"""
    blk = 1 << shift
    s = []
    s.append([("read", rows)])                                     # plain sequential full decode
    if name == "longrun":
        s.append([("subset", [0, 1, 31337, m - 1, 4096, 65535, 65536]), ("read", rows)])
        s.append([("seek", 5), ("read", 3)])
        return s
    half = list(range(0, m, 2))
    few = sorted(set([0, m - 1, m // 3, m // 2]))
    s.append([("subset", few), ("read", rows)])                    # subset from row 0 (bgt_prepare order)
    s.append([("subset", half), ("read", rows)])
    s.append([("seek", 3), ("read", 2)])                           # pbfview -r3 -n2
    s.append([("seek", min(rows - 1, blk + 2)), ("read", 3)])      # forward seek across a checkpoint
    if rows > 3 * blk:
        s.append([("seek", 3 * blk + 1), ("read", 4)])             # jump: restart from checkpoint
        s.append([("subset", few), ("seek", 3 * blk + 1), ("read", 4),   # subset, jump, re-init ranks
                  ("seek", 2), ("read", 3),                               # backward jump
                  ("seek", 2 * blk), ("read", blk + 1)])                   # read through a checkpoint
        s.append([("seek", 2 * blk + 3), ("subset", half), ("read", blk + 2)])  # pbfview order: seek, subset
        s.append([("subset", half), ("read", 2), ("seek", 2 + blk), ("read", 2),  # exactly 1<<shift ahead
                  ("seek", 2 + 2 * blk + 3 + blk), ("read", 1)])                 # 1 more than the window
    # unsorted / duplicated column list as bgt never produces but pbfview allows
    s.append([("subset", [m - 1, 0, m // 2]), ("read", min(rows, 10))])
    return s
