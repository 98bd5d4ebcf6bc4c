#!/usr/bin/env python3
"""C3 (every 20th sample of 100,000, 1 M sites by default): plane 1 by the dense plane kernel against the sparse tracker.
usage: python scripts/c3_ab.py [samples] [sites] [every-nth]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bgt_amd  # noqa: E402

samples = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
sites = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
sub = int(sys.argv[3]) if len(sys.argv) > 3 else 20
m = 2 * samples
t0 = time.time()
rle, lens = bgt_amd.synth_rows(m, 0, sites, 3)
pbf = bgt_amd.HipPbf.from_rle(m, 13, rle, lens)
print("cohort m=%d sites=%d setup %.1fs" % (m, sites, time.time() - t0), flush=True)
rd = bgt_amd.HipReader(pbf)
if sub:
    s = np.arange(0, samples, sub)
    rd.select(np.stack([2 * s, 2 * s + 1], 1).reshape(-1))
res = []
force = int(os.environ.get("FORCE_PLANE", "0"))          # FORCE_PLANE=4096: the plane-split kernels on a narrow cohort too
for label, var in (("dense plane 1", 0), ("sparse tracker (BGTH_VARIANT=262144)", 262144), ("dense again", 0), ("sparse again", 262144)):
    os.environ["BGTH_VARIANT"] = str(var + force)
    rd.scan(0, min(sites, 8192))
    best = 1e9
    for _ in range(3):
        counts = rd.scan(0, sites)
        best = min(best, rd.timing()["scan_ms"])
    res.append(counts)
    print("%-38s %s %s : %8.3f ms  %7.2f M sites/s" % (label, rd.path(), rd.geometry(), best, sites / best / 1e3), flush=True)
print("same counts:", all(np.array_equal(res[0], r) for r in res[1:]))
