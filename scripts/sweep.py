#!/usr/bin/env python3
"""Launch-geometry sweep of the scan kernel on one synthetic cohort (tuning aid, GPU box).
usage: python scripts/sweep.py [samples] [sites] [configs "threads,cpt,K;..."]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bgt_amd  # noqa: E402

samples = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
sites = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
cfgs = sys.argv[3] if len(sys.argv) > 3 else "0,0,0;1024,24,0;1024,16,0;1024,8,0;512,16,0;512,8,0;256,16,0;256,8,0;256,16,4;512,16,4;1024,16,4;1024,16,16"
sub = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # every sub-th sample only (0 = all)
m = 2 * samples
t0 = time.time()
rle, lens = bgt_amd.synth_rows(m, 0, sites, 2)
pbf = bgt_amd.HipPbf.from_rle(m, 13, rle, lens)
print("cohort m=%d sites=%d rle=%.1f MB setup %.1fs" % (m, sites, rle.size / 1e6, time.time() - t0), flush=True)
rd = bgt_amd.HipReader(pbf)
if sub:
    s = np.arange(0, samples, sub)
    rd.select(np.stack([2 * s, 2 * s + 1], 1).reshape(-1))
ref = None
for c in cfgs.split(";"):
    th, cpt, K = (int(x) for x in c.split(","))
    rd.tune(th, cpt, K)
    try:
        rd.scan(0, min(sites, 8192))                      # warm
        best = 1e9
        for _ in range(2):
            counts = rd.scan(0, sites)
            best = min(best, rd.timing()["scan_ms"])
    except RuntimeError as e:
        print(c, "->", e)
        continue
    if ref is None:
        ref = counts
    ok = np.array_equal(ref, counts)
    g = rd.geometry()
    print("cfg %-12s -> %4dthr cpt%2d slices%2d K%2d lds%6d wgs%5d : %8.2f ms  %7.2f Msites/s  same=%s" %
          (c, g["threads"], g["cols_per_thread"], g["slices"], g["rows_per_batch"], g["lds_bytes"], g["workgroups"],
           best, sites / best / 1e3, ok), flush=True)
