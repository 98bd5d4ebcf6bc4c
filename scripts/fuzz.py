#!/usr/bin/env python3
"""Differential fuzz on the GPU box: random cohorts / selections / groups / launch geometries / row ranges through
the C ABI against the CPU oracle.  usage: python scripts/fuzz.py [seconds] [seed]"""
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
GEOMS = [(0, 0, 0), (256, 2, 1), (256, 8, 3), (256, 20, 2), (512, 4, 1), (512, 10, 2), (512, 20, 5), (512, 48, 1), (512, 64, 1),
         (512, 80, 1), (512, 98, 1), (1024, 4, 9), (1024, 8, 1), (1024, 10, 0), (1024, 16, 2), (1024, 20, 0), (1024, 24, 1)]
t_end = time.time() + budget
n_case = n_check = 0
while time.time() < t_end:
    # (round 6: 400,000 = one plane per workgroup in the LDS; 700,001 / 1,100,000 = toggles and directory entries in memory)
    m = int(rng.choice([1, 2, 3, 63, 64, 65, 127, 500, 1000, 2049, 5008, 9000, 33000, 70002, 120001, 400000, 700001, 1100000],
                       p=[1 / 16] * 15 + [1 / 48] * 3))
    rows = int(rng.integers(1, 400 if m < 10000 else 60 if m < 50000 else 12 if m < 300000 else 6))
    shift = int(rng.integers(2, 9))
    mat = scenarios.ld_matrix(rng, rows, m, n_founders=int(rng.integers(2, 30)), switch=float(rng.choice([0.0, 0.01, 0.1, 0.5])))
    style = rng.integers(0, 4)
    if style == 1:
        mat &= 1                                              # fully called: plane 1 empty everywhere
    elif style == 2:
        mat[rng.random(rows) < 0.5] &= 1
    elif style == 3 and rows > 3:
        mat[rng.integers(0, rows)] = rng.integers(0, 4); mat[rng.integers(0, rows)] = 3
    data = orc.encode_pbf(mat, 2, shift)
    bgt_amd.force_kernels(0)
    # kernel variants with identical results: 1 toggles in place, 2 / 4 never / always the empty-plane kernels, 32 the
    # directory path (producer + walk-only kernels) forced, 4096 the plane-split kernels forced, 128 no arena reuse
    flag = int(rng.choice([0, 0, 0, 8, 1, 2, 4, 4 | 1, 32, 32, 32 | 8, 32 | 128, 4096, 4096, 4096 | 2, 32 | 2]))
    if flag:
        bgt_amd.force_kernels(int(str(flag)))
    os.environ["BGTH_SUB_SHIFT"] = str(int(rng.integers(1, 12)))
    os.environ.pop("BGTH_DIR_ARENA_MB", None)                 # (a cap of the case before must not meet this one's width)
    pbf = bgt_amd.HipPbf.from_bytes(data)
    for _ in range(3):
        ora = orc.Pbf(data)
        rd = bgt_amd.HipReader(pbf)
        th, cpt, K = GEOMS[int(rng.integers(0, len(GEOMS)))]
        if flag & (32 | 4096):
            th = cpt = K = 0                                  # (a tuned geometry names a classic kernel)
            if rng.random() < 0.3:
                # several passes over the arena (a cohort of one plane per workgroup needs room for ONE file block of its rows:
                # 2^shift <= 256 rows x up to 0.8 MB here)
                os.environ["BGTH_DIR_ARENA_MB"] = "1" if m < 300000 else "300"
            else:
                os.environ.pop("BGTH_DIR_ARENA_MB", None)
        rd.tune(th, cpt, K)
        cols = group = None
        G = 1
        if m >= 2 and rng.random() < 0.6:
            ns = m // 2
            pick = np.sort(rng.choice(ns, int(rng.integers(1, ns + 1)), replace=False))
            cols = np.stack([2 * pick, 2 * pick + 1], 1).reshape(-1).astype(np.int32)
            if rng.random() < 0.5:
                G = int(rng.integers(2, 7))
                group = rng.integers(1, G + 1, pick.size).astype(np.uint32)
            rd.select(cols, group=group, n_groups=G)
            ora.subset(cols)
        a = int(rng.integers(0, rows)); b = int(rng.integers(a + 1, rows + 1))
        try:
            if rng.random() < 0.3:
                counts, gt = rd.scan(a, b), None              # counts only: the plane-split kernels use their own planes
            else:
                counts, gt = rd.scan(a, b, want_gt=True)
        except RuntimeError as e:
            if "no launch geometry" in str(e) or "directory arena" in str(e):
                continue
            raise
        oc, ogt = ora.scan(a, b, group=group, n_groups=G, want_gt=True)
        ok = np.array_equal(counts.reshape(b - a, -1), oc.reshape(b - a, -1)) and (gt is None or np.array_equal(gt, ogt))
        if ok and rng.random() < 0.2:                         # the same reader again: an arena kept from the scan before
            c2 = rd.scan(a, b)
            ok = np.array_equal(c2.reshape(b - a, -1), oc.reshape(b - a, -1))
        n_check += 1
        if not ok:
            print("MISMATCH m=%d rows=%d shift=%d geom=%s flag=%d sub=%s range=[%d,%d) G=%d cols=%s" %
                  (m, rows, shift, (th, cpt, K), flag, os.environ["BGTH_SUB_SHIFT"], a, b, G, None if cols is None else cols.size))
            sys.exit(1)
        rd.close()
    pbf.close()
    n_case += 1
print("fuzz ok: %d images, %d scans checked in %.0f s" % (n_case, n_check, budget))
