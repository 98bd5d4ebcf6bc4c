#!/usr/bin/env python3
"""A/B of the narrow scan kernel's row step on the C2 cohort (GPU box): the shipped ballot step (BGTH_VARIANT=65536) against
the ballot-free instruction-major step.  usage: python scripts/c2_ab.py [samples] [sites]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bgt_amd  # noqa: E402

samples = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
sites = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
m = 2 * samples
t0 = time.time()
rle, lens = bgt_amd.synth_rows(m, 0, sites, 2)
pbf = bgt_amd.HipPbf.from_rle(m, 13, rle, lens)
print("cohort m=%d sites=%d setup %.1fs" % (m, sites, time.time() - t0), flush=True)
rd = bgt_amd.HipReader(pbf)
res = {}
for label, var in (("both planes dense", None), ("plane 1 by the sparse tracker (BGTH_VARIANT=262144)", "262144"), ("dense again", None), ("tracker again", "262144")):
    if var is None:
        os.environ.pop("BGTH_VARIANT", None)
    else:
        os.environ["BGTH_VARIANT"] = var
    rd.scan(0, min(sites, 8192))
    best = 1e9
    for _ in range(3):
        counts = rd.scan(0, sites)
        best = min(best, rd.timing()["scan_ms"])
    res[label] = counts
    g = rd.geometry()
    print("%-40s %4d thr x %2d cols x %d slices K %d sparse %s: %8.3f ms  %7.2f M sites/s" % (label, g["threads"], g["cols_per_thread"], g["slices"], g["rows_per_batch"], rd.path()["sparse_plane1"], best, sites / best / 1e3), flush=True)
keys = list(res)
print("same counts:", all(np.array_equal(res[keys[0]], res[k]) for k in keys[1:]))
