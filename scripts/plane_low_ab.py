#!/usr/bin/env python3
"""Plane-split kernels at two or three workgroups per CU (BGTH_PLANE_LOW=0 / 1: the <= 80-VGPR statement): one process per
setting.  usage: python scripts/plane_low_ab.py [samples] [sites] [every-nth]   (reads BGTH_PLANE_LOW from the environment)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bgt_amd  # noqa: E402

samples = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
sites = int(sys.argv[2]) if len(sys.argv) > 2 else 524288
sub = int(sys.argv[3]) if len(sys.argv) > 3 else 20
m = 2 * samples
rle, lens = bgt_amd.synth_rows(m, 0, sites, 3)
pbf = bgt_amd.HipPbf.from_rle(m, 13, rle, lens)
rd = bgt_amd.HipReader(pbf)
s = np.arange(0, samples, sub)
rd.select(np.stack([2 * s, 2 * s + 1], 1).reshape(-1))
rd.scan(0, min(sites, 8192))
best = 1e9
for _ in range(4):
    counts = rd.scan(0, sites)
    best = min(best, rd.timing()["scan_ms"])
import hashlib
print("BGTH_PLANE_LOW=%s m=%d sites=%d every %d: %s %s : %8.3f ms  %7.2f M sites/s  counts %s" % (
    os.environ.get("BGTH_PLANE_LOW", "auto"), m, sites, sub, rd.path()["plane_split"], rd.geometry(), best, sites / best / 1e3,
    hashlib.md5(np.ascontiguousarray(counts).tobytes()).hexdigest()[:8]), flush=True)
