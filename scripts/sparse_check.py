#!/usr/bin/env python3
"""The sparse plane-1 tracker (scan_sparse.hip) against the CPU oracle on small shapes (GPU box): forced with
BGTH_VARIANT = 4096 (plane-split kernels) + 262144 (tracker whenever possible), tiny tails so that epochs turn over.
usage: python scripts/sparse_check.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import bgt_amd  # noqa: E402
import orc  # noqa: E402
import scenarios  # noqa: E402

bad = 0
for seed, m, rows, shift, n_sel, pm in [(1, 700, 90, 4, 40, 0.02), (2, 5000, 300, 6, 300, 0.004), (3, 41000, 40, 3, 900, 0.001),
                                        (4, 64, 20, 2, 3, 0.05), (5, 5000, 2100, 13, 1, 0.002), (6, 9000, 300, 6, 4500, 0.2),
                                        (7, 300, 600, 5, 150, 0.3)]:
    rng = np.random.default_rng(seed)
    mat = scenarios.ld_matrix(rng, rows, m, n_founders=7, switch=0.1, p_missing=pm, p_multi=pm)
    mat[2] = 0; mat[3] = 1; mat[4] = 3
    mat[5] = rng.integers(0, 4, m)
    data = orc.encode_pbf(mat, 2, shift)
    pbf = bgt_amd.HipPbf.from_bytes(data)
    rd = bgt_amd.HipReader(pbf)
    smp = np.sort(rng.choice(m // 2, n_sel, replace=False))
    cols = np.stack([2 * smp, 2 * smp + 1], 1).reshape(-1).astype(np.int32)
    for tcap in ("4096", None):
        os.environ["BGTH_VARIANT"] = str(4096 + 262144)
        if tcap:
            os.environ["BGTH_SPARSE_TCAP"] = tcap
        else:
            os.environ.pop("BGTH_SPARSE_TCAP", None)
        for n_groups in (1, 3):
            group = (1 + (np.arange(n_sel) % n_groups)).astype(np.uint32) if n_groups > 1 else None
            rd.select(cols, group=group, n_groups=n_groups)
            o = orc.Pbf(data)
            o.subset(cols)
            oc, ogt = o.scan(0, rows, group=group, n_groups=n_groups, want_gt=True)
            oc = oc.reshape(rows, -1)
            c, g = rd.scan(0, rows, want_gt=True)
            p = rd.path()
            ok = np.array_equal(c.reshape(rows, -1), oc) and np.array_equal(g, ogt)
            a, b = rows // 3, rows - 1
            ok2 = np.array_equal(rd.scan(a, b).reshape(b - a, -1), oc[a:b])
            print("seed %d m %d rows %d sel %d groups %d tcap %s: path %s  full %s  mid-block %s" % (seed, m, rows, n_sel, n_groups, tcap, p, ok, ok2), flush=True)
            bad += (not ok) + (not ok2) + (pm < 0.1 and not p["sparse_plane1"])
    rd.close(); pbf.close()
print("FAILED %d" % bad if bad else "sparse tracker: all identical to the oracle")
sys.exit(1 if bad else 0)
