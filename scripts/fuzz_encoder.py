#!/usr/bin/env python3
"""Differential fuzz of the device writer on the GPU box: random widths / rows / checkpoint spacings / unit sizes /
call splits against the oracle writer (whole image compared).  usage: python scripts/fuzz_encoder.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import bgt_amd  # noqa: E402
import orc  # noqa: E402
import scenarios  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
n_case = 0
while time.time() < t_end:
    m = int(rng.choice([1, 2, 3, 31, 32, 33, 63, 64, 65, 500, 1023, 1024, 1025, 4095, 4096, 4097, 5008, 8191, 8192, 8193, 12345,
                        20000, 20479, 20480, 20481, 30000, 32767, 32768, 32769, 50001, 65536, 65537, 100000, 131072, 131073, 200000,
                        262145, 400000, 524289, 700001, 1100000]))          # (round 6, beyond 262,144: the directories in memory)
    rows = int(rng.integers(1, 500 if m < 6000 else 120 if m < 40000 else 30 if m < 250000 else 10))
    shift = int(rng.integers(0, 9))
    g = int(rng.choice([1, 2, 2, 2]))
    os.environ["BGTH_ENC_UNIT_SHIFT"] = str(int(rng.integers(1, 8)))
    style = int(rng.integers(0, 6))
    if style == 0:
        mat = rng.integers(0, 4, (rows, m)).astype(np.uint8)
    else:
        mat = scenarios.ld_matrix(rng, rows, m, n_founders=int(rng.integers(1, 12)), switch=float(rng.choice([0.0, 0.0, 0.001, 0.05])))
        if style == 2:
            mat[rng.random(rows) < 0.3] = 0
        elif style == 3:
            mat[rng.random(rows) < 0.2] = 3
        elif style == 4:
            mat &= 1
        elif style == 5:
            mat = np.repeat(mat[:, :max(1, m // 7)], 7, axis=1)[:, :m]          # blocks of identical neighbours
            if mat.shape[1] < m:
                mat = np.concatenate([mat, np.zeros((rows, m - mat.shape[1]), np.uint8)], axis=1)
    if g == 1:
        mat &= 1
    enc = bgt_amd.HipEncoder(m, g, shift)
    cuts = np.unique(np.concatenate([[0, rows], rng.integers(0, rows + 1, int(rng.integers(0, 4)))]))
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        if rng.random() < 0.4:                                # the same rows four columns to a byte
            pad = np.zeros((hi - lo, (m + 3) // 4 * 4), np.uint8)
            pad[:, :m] = mat[lo:hi]
            enc.write_packed((pad[:, 0::4] | pad[:, 1::4] << 2 | pad[:, 2::4] << 4 | pad[:, 3::4] << 6).astype(np.uint8))
        else:
            enc.write(mat[lo:hi])
    got = enc.finish()
    enc.close()
    want = orc.encode_pbf(mat, g, shift)
    if got != want:
        np.save(os.path.join(ROOT, "gpurun_out", "fuzz_encoder_fail.npy"), mat)
        print("MISMATCH m=%d rows=%d shift=%d g=%d unit=%s style=%d cuts=%s" % (m, rows, shift, g, os.environ["BGTH_ENC_UNIT_SHIFT"], style, cuts))
        sys.exit(1)
    n_case += 1
print("fuzz_encoder: %d images identical to the oracle writer's" % n_case)
