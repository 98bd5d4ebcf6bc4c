#!/usr/bin/env python3
"""Sparse selections of wide cohorts (plane-split kernels): the selection's compact start-rank table (round 6, gathered once per
selection) against every workgroup gathering its columns' ranks from the [2][m] checkpoint records (BGTH_FORCE_COLUMN_ORDER).
usage: python scripts/start_ranks_ab.py [samples:sites:every,...]   (kernel time by HIP events, best of 5; same counts checked)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bgt_amd  # noqa: E402

shapes = sys.argv[1] if len(sys.argv) > 1 else "100000:1000000:20,32488:142000:13,100000:262144:10"
for sh in shapes.split(","):
    samples, sites, sub = (int(x) for x in sh.split(":"))
    m = 2 * samples
    rle, lens = bgt_amd.synth_rows(m, 0, sites, 3)
    pbf = bgt_amd.HipPbf.from_rle(m, 13, rle, lens)
    del rle
    rd = bgt_amd.HipReader(pbf)
    s = np.arange(0, samples, sub)
    rd.select(np.stack([2 * s, 2 * s + 1], 1).reshape(-1))
    res = []
    for label, var in (("compact table", 0), ("gather per workgroup", bgt_amd.hip.FORCE_COLUMN_ORDER), ("compact table", 0)):
        bgt_amd.force_kernels(var)
        rd.scan(0, min(sites, 8192))
        best = 1e9
        for _ in range(5):
            counts = rd.scan(0, sites)
            best = min(best, rd.timing()["scan_ms"])
        res.append(counts)
        g, p = rd.geometry(), rd.path()
        print("%-22s %-22s %8.3f ms  %4d thr x %2d col  unit %d rows  %s" % (sh, label, best, g["threads"], g["cols_per_thread"], pbf.unit_rows,
              "plane" if p["plane_split"] else "other"), flush=True)
    print("   same counts:", all(np.array_equal(res[0], r) for r in res[1:]), flush=True)
    bgt_amd.force_kernels(0)
    rd.close(); pbf.close()
