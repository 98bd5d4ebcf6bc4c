#!/usr/bin/env python3
"""Whole-cohort scan on the directory path with the walk-only kernel's geometry forced (BGTH_WALK_GEOM=threads,cols; one process
per setting).  usage: BGTH_WALK_GEOM=512,32 python scripts/walk_geom_ab.py [samples] [sites]"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bgt_amd  # noqa: E402

samples = int(sys.argv[1]) if len(sys.argv) > 1 else 32488
sites = int(sys.argv[2]) if len(sys.argv) > 2 else 524288
m = 2 * samples
rle, lens = bgt_amd.synth_rows(m, 0, sites, 7)
pbf = bgt_amd.HipPbf.from_rle(m, 13, rle, lens)
rd = bgt_amd.HipReader(pbf)
bgt_amd.force_kernels(int("128"))                     # every scan builds its rows
rd.scan(0, min(sites, 16384))
best, walk = 1e9, 1e9
for _ in range(3):
    counts = rd.scan(0, sites)
    t = rd.timing()
    if t["scan_ms"] < best:
        best, walk = t["scan_ms"], t["scan_ms"] - rd.path().get("producer_ms", 0.0)
g = rd.geometry()
print("WALK_GEOM=%-8s m=%d sites=%d: %d thr x %d col x %d slices dir=%s : %8.3f ms (producer %.2f)  %7.2f M sites/s  %.2f T lookups/s  counts %s" % (
    os.environ.get("BGTH_WALK_GEOM", "auto"), m, sites, g["threads"], g["cols_per_thread"], g["slices"], rd.path()["directory_path"], best,
    rd.path().get("producer_ms", 0.0), sites / best / 1e3, 2.0 * m * sites / best / 1e9, hashlib.md5(np.ascontiguousarray(counts).tobytes()).hexdigest()[:8]), flush=True)
