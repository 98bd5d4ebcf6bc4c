"""The synthetic-cohort generator (host code of the product) checked with the oracle's run-length code:
strings cover exactly m symbols, are canonical (re-encoding the decoded bits reproduces the bytes, i.e.
what the reference encoder would write), reproducible and independent of the thread count."""
import numpy as np
import pytest

import bgt_amd
import orc


def decode_bits(s, m):
    lens = np.array([orc.rle_len(int(b)) for b in s], np.int64)
    bits = np.repeat(np.frombuffer(s, np.uint8) & 1, lens)
    assert bits.size == m
    return bits.astype(np.uint8)


@pytest.mark.parametrize("m", [1, 2, 64, 5008, 20000, 200000])
def test_strings_are_canonical(m):
    n = 40 if m > 50000 else 200
    rle, lens = bgt_amd.synth_rows(m, 0, n, seed=3, n_threads=2)
    assert lens.sum() == rle.size
    off = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
    ones0 = []
    for i in range(2 * n):
        s = bytes(rle[off[i]:off[i + 1]])
        assert 0 not in s
        bits = decode_bits(s, m)
        assert orc.rle_encode(bits) == s
        if i % 2 == 0:
            ones0.append(int(bits.sum()))
    if m >= 64:
        assert len(set(ones0)) > 5               # a spectrum of allele counts, not a constant


def test_reproducible_and_thread_independent():
    a = bgt_amd.synth_rows(5008, 100, 5000, seed=9, n_threads=1)
    b = bgt_amd.synth_rows(5008, 100, 5000, seed=9, n_threads=4)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    c = bgt_amd.synth_rows(5008, 0, 5100, seed=9, n_threads=3)       # row ranges compose
    off = int(np.cumsum(c[1].astype(np.int64))[199])
    assert np.array_equal(c[1][200:], a[1]) and np.array_equal(c[0][off:], a[0])
    d = bgt_amd.synth_rows(5008, 100, 5000, seed=10)
    assert not np.array_equal(a[1], d[1])


def test_row_sizes_resemble_the_survey_measurement():
    # ~220 B + ~130 B per row at m = 20,000
    rle, lens = bgt_amd.synth_rows(20000, 0, 20000, seed=2)
    per_row = lens.reshape(-1, 2).mean(0)
    assert 100 < per_row[0] < 500 and 40 < per_row[1] < 300, per_row
