"""Pins the CPU oracle (oracle/liborc.so) to the reference: every golden fixture under tests/golden/ was
produced by the COMPILED reference (tests/golden/make_golden.py); when oracle/_ref is present (it travels
with the repo to the GPU box, it is never committed) the oracle is additionally compared live with the
reference library on fresh random inputs."""
import hashlib
import os

import numpy as np
import pytest

import orc
import scenarios

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "codec.npz"))
REF_SO = os.path.join(orc.ROOT, "oracle", "_ref", "libbgt_ref.so")


def replay(p, ops):
    out = []
    for op in ops:
        if op[0] == "subset":
            p.subset(op[1])
        elif op[0] == "seek":
            p.seek(op[1])
        else:
            for _ in range(op[1]):
                a = p.read()
                if a is None:
                    break
                out.append(a)
    return np.stack(out) if out else np.zeros((0, p.g, p.width()), np.uint8)


def test_rle_table_formula():
    # ref pbwt.c:12-21: entry [16k+d] = d << 4k
    for k in range(8):
        for d in range(16):
            for bit in (0, 1):
                assert orc.rle_len(((16 * k + d) << 1) | bit) == (d << (4 * k)) & 0xFFFFFFFF


@pytest.mark.parametrize("length", [1, 2, 15, 16, 17, 255, 256, 257, 4095, 4096, 65535, 65536, 0x12345,
                                    0xF0F0F, 199999, 0x7FFFFFFF])
def test_rle_put_run_roundtrip(length):
    for bit in (0, 1):
        b = orc.rle_put_run(length, bit)
        assert sum(orc.rle_len(x) for x in b) == length
        assert all((x & 1) == bit for x in b)
        if length >= 16:    # one byte per non-zero hex digit, high digit first
            assert len(b) == sum(1 for i in range(8) if (length >> (4 * i)) & 15)
            lens = [orc.rle_len(x) for x in b]
            assert lens == sorted(lens, reverse=True)


def test_ex1_pbf_bytes():
    ex1 = open(os.path.join(HERE, "golden", "ex1.pbf"), "rb").read()
    assert hashlib.md5(ex1).hexdigest() == "ffeac837ea3d039ec92a2da801901bd5"
    mat, shift = scenarios.cases()["ex1"]
    assert orc.encode_pbf(mat, 2, shift) == ex1
    # first 'B' record of ex1 per SURVEY App. A: 03 00 00 00 | 02 03 04
    at = 16 + 1 + 2 * 4 * 4
    assert ex1[at:at + 8] == b"B\x03\x00\x00\x00\x02\x03\x04"


@pytest.mark.parametrize("name", list(scenarios.cases().keys()))
def test_writer_matches_reference_bytes(name):
    mat, shift = scenarios.cases()[name]
    assert orc.encode_pbf(mat, 2, shift) == bytes(GOLD[name + "/pbf"])


@pytest.mark.parametrize("name", list(scenarios.cases().keys()))
def test_reader_scenarios_match_reference(name):
    mat, shift = scenarios.cases()[name]
    rows, m = mat.shape
    data = bytes(GOLD[name + "/pbf"])
    for i, ops in enumerate(scenarios.scenarios(name, rows, m, shift)):
        got = replay(orc.Pbf(data), ops)
        exp = GOLD["%s/s%d" % (name, i)]
        if name == "longrun":
            w = int(GOLD["%s/s%d_w" % (name, i)])
            exp = np.unpackbits(exp, axis=-1)[..., :w]
        assert got.shape == exp.shape, (name, i)
        assert np.array_equal(got, exp), (name, i)


@pytest.mark.parametrize("name", ["blocks", "wide", "odd", "ex1"])
def test_full_decode_recovers_matrix(name):
    mat, shift = scenarios.cases()[name]
    got = replay(orc.Pbf(bytes(GOLD[name + "/pbf"])), [("read", mat.shape[0])])
    assert np.array_equal(got[:, 0] | (got[:, 1] << 1), mat)


def test_allele_counts_by_hand():
    a0 = np.array([0, 1, 0, 1, 1, 0, 0, 0], np.uint8)
    a1 = np.array([0, 0, 1, 1, 0, 0, 1, 0], np.uint8)      # codes 0 1 2 3 1 0 2 0
    assert list(orc.allele_counts(a0, a1)) == [6, 2, 1]
    grp = np.array([1, 2, 2, 1], np.uint32)                 # samples 0,3 -> g1 ; 1,2 -> g2
    got = list(orc.allele_counts(a0, a1, grp, 2))
    assert got == [6, 2, 1, 3, 1, 0, 3, 1, 1]


def test_scan_counts_equal_matrix_histogram():
    mat, shift = scenarios.cases()["blocks"]
    p = orc.Pbf(bytes(GOLD["blocks/pbf"]))
    counts, gt = p.scan(0, mat.shape[0], want_gt=True)
    assert np.array_equal(counts[:, 1], (mat == 1).sum(1))
    assert np.array_equal(counts[:, 2], (mat == 3).sum(1))
    assert np.array_equal(counts[:, 0], (mat != 2).sum(1))
    codes = np.stack([(gt >> (2 * k)) & 3 for k in range(4)], -1).reshape(mat.shape[0], -1)[:, :mat.shape[1]]
    assert np.array_equal(codes, mat)


# ---------------------------------------------------------------------------------------------------
# live comparison with the compiled reference (present in the build container and on the GPU box)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed,m,rows,shift", [(1, 64, 40, 3), (2, 129, 300, 5), (3, 1000, 50, 13),
                                               (4, 7, 100, 2), (5, 2504 * 2, 30, 13)])
def test_live_against_reference(tmp_path, seed, m, rows, shift):
    from conftest import require_ref
    require_ref("libbgt_ref.so")
    import make_golden_lib as mg
    rng = np.random.default_rng(seed)
    mat = scenarios.ld_matrix(rng, rows, m, n_founders=int(rng.integers(2, 12)), switch=float(rng.random() * 0.2))
    path = str(tmp_path / "x.pbf")
    mg.ref_write_pbf(path, mat, shift)
    data = open(path, "rb").read()
    assert orc.encode_pbf(mat, 2, shift) == data
    blk = 1 << shift
    sub = sorted(rng.choice(m, size=max(1, m // 5), replace=False).tolist())
    for ops in ([("read", rows)], [("subset", sub), ("read", rows)],
                [("subset", sub), ("seek", min(rows - 1, 2 * blk + 1)), ("read", 5), ("seek", 1), ("read", 2)],
                [("seek", rows // 2), ("subset", sub), ("read", rows)]):
        exp = mg.ref_replay(path, m, 2, ops)
        got = replay(orc.Pbf(data), ops)
        assert np.array_equal(got, exp), ops
