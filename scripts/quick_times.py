#!/usr/bin/env python3
"""Kernel times of the benchmark shapes, one line each (GPU box): best of N scans by the HIP events around the scan kernels, the
launch geometry, and an md5 of the counts (the same cohort must give the same md5 under every build).
usage: python scripts/quick_times.py [shape ...]     shapes: c2 c3 hrc hrcsub c4 hrc13k small   (default: c2 c3 hrc hrcsub)"""
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bgt_amd  # noqa: E402

SHAPES = {  # name: (samples, sites, seed, every, reps)
    "c2": (10000, 1000000, 2, 0, 8), "c3": (100000, 1000000, 3, 20, 5), "hrc": (32488, 142000, 7, 0, 10),
    "hrcsub": (32488, 142000, 7, 13, 10), "c4": (100000, 153 * 8192, 4, 0, 2), "small": (2504, 1000000, 1, 0, 8),
    "team40k": (20000, 262144, 11, 0, 4), "mid10k": (5000, 1000000, 12, 0, 5), "c3half": (50000, 1000000, 3, 10, 5), "hrc13k": (13000, 1000000, 9, 0, 5), "hrclong": (32488, 524288, 7, 0, 4),
}
for name in (sys.argv[1:] or ["c2", "c3", "hrc", "hrcsub"]):
    samples, sites, seed, every, reps = SHAPES[name]
    m = 2 * samples
    rle, lens = bgt_amd.synth_rows(m, 0, sites, seed)
    pbf = bgt_amd.HipPbf.from_rle(m, 13, rle, lens)
    del rle
    rd = bgt_amd.HipReader(pbf)
    if every:
        s = np.arange(0, samples, every)
        rd.select(np.stack([2 * s, 2 * s + 1], 1).reshape(-1))
    bgt_amd.force_kernels(bgt_amd.hip.FORCE_REBUILD_ROWS)       # one-shot figures: every scan builds its rows
    best, out = None, None
    for _ in range(reps):
        out = rd.scan(0, sites)
        t = rd.timing()["scan_ms"]
        best = t if best is None or t < best else best
    g, p = rd.geometry(), rd.path()
    print("%-7s m=%6d T=%6d sites=%7d : %8.3f ms  %7.2f M sites/s  %5.2f T lookups/s  %dx%dx%d K%d wgs %d %s md5 %s" % (
        name, m, rd.width, sites, best, sites / best / 1e3, 2.0 * rd.width * sites / best / 1e9, g["threads"], g["cols_per_thread"],
        g["slices"], g["rows_per_batch"], g["workgroups"], "dir" if p["directory_path"] else "plane" if p["plane_split"] else "scan",
        hashlib.md5(np.ascontiguousarray(out).tobytes()).hexdigest()[:12]), flush=True)
    bgt_amd.force_kernels(0)
    rd.close()
    pbf.close()
