#!/usr/bin/env python3
"""Wide-cohort scan: the team kernels (every column slice rebuilds the row) against the directory path (rows built once
into an HBM arena, walk-only slices), one-shot and with the arena reused.  GPU box.
usage: python scripts/wide_bench.py [samples] [sites] [every-nth-sample]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bgt_amd  # noqa: E402

samples = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
sites = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
sub = int(sys.argv[3]) if len(sys.argv) > 3 else 0
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 2
m = 2 * samples
t0 = time.time()
rle, lens = bgt_amd.synth_rows(m, 0, sites, seed)
pbf = bgt_amd.HipPbf.from_rle(m, 13, rle, lens)
print("cohort m=%d sites=%d rle=%.1f MB setup %.1fs" % (m, sites, rle.size / 1e6, time.time() - t0), flush=True)
rd = bgt_amd.HipReader(pbf)
if sub:
    s = np.arange(0, samples, sub)
    rd.select(np.stack([2 * s, 2 * s + 1], 1).reshape(-1))


def run(label, variant, reps=3):
    if variant is None:
        bgt_amd.force_kernels(0)
    else:
        bgt_amd.force_kernels(int(str(variant)))
    out = None
    for i in range(reps):
        t = time.time()
        out = rd.scan(0, sites)
        wall = (time.time() - t) * 1e3
        tm, g, p = rd.timing(), rd.geometry(), rd.path()
        print("%-22s rep %d: scan %8.2f ms (producer %6.2f ms, passes %d, built %d) wall %8.1f ms  %6.2f Msites/s   %dthr x %d cols x %d slices, lds %d, wgs %d" %
              (label, i, tm["scan_ms"], p["producer_ms"], p["passes"], p["producer_launches"], wall, sites / tm["scan_ms"] / 1e3,
               g["threads"], g["cols_per_thread"], g["slices"], g["lds_bytes"], g["workgroups"]), flush=True)
    return out


rd.scan(0, min(sites, 8192))
if sub:                                               # sparse selection: team kernel against the plane-split kernels
    ref = run("team kernel, no prio", 64 | 2048 | 16384, 3)
    ref = run("team kernel", 64 | 2048, 3)
    run("plane-split, no prio", 64 | 16384, 3)
    a = run("plane-split kernels", 64, 3)
    print("same counts:", np.array_equal(ref, a), rd.path())
    sys.exit(0)
run("team kernels, no prio", 64 | 16384, 2)
ref = run("team kernels", 64, 2)
a = run("directory, arena kept", None)
b = run("directory, one-shot", 128, 2)
c = run("directory, no priorities", 16384, 3)
os.environ["BGTH_WALK_GEOM"] = "1024,50"
d = run("directory 1024x50", None, 3)
e = run("dir 1024x50, no warm", 256, 3)
os.environ.pop("BGTH_WALK_GEOM")
print("same counts:", [np.array_equal(ref, x) for x in (a, b, c, d, e)])
