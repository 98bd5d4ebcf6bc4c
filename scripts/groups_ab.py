#!/usr/bin/env python3
"""Whole-cohort scans with sample groups (-s A -s B ...: per-group AN / AC beside the totals) against the same scan without:
what the per-group counting costs.  usage: python scripts/groups_ab.py [samples] [sites]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bgt_amd  # noqa: E402

samples = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
sites = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
m = 2 * samples
rle, lens = bgt_amd.synth_rows(m, 0, sites, 2)
pbf = bgt_amd.HipPbf.from_rle(m, 13, rle, lens)
rd = bgt_amd.HipReader(pbf)
cols = np.arange(m, dtype=np.int32)
bgt_amd.force_kernels(int("128"))
base = None
for G, how in ((1, "none"), (2, "halves"), (2, "alternating samples"), (4, "quarters"), (8, "eighths")):
    if G == 1:
        rd.select(None)
    else:
        g = (np.arange(samples) * G // samples) if how != "alternating samples" else (np.arange(samples) % G)
        rd.select(cols, group=(1 + g).astype(np.uint32), n_groups=G)      # one id in 1..G per SAMPLE
    rd.scan(0, min(sites, 8192))
    best = 1e9
    for _ in range(3):
        c = rd.scan(0, sites)
        best = min(best, rd.timing()["scan_ms"])
    if base is None:
        base = c[:, 0].copy()
    ok = np.array_equal(c[:, 0], base) and (G == 1 or np.array_equal(c[:, 1:].sum(1), base))
    g_ = rd.geometry()
    print("groups %d (%-19s): %8.3f ms  %7.2f M sites/s  %d thr x %d col x %d slices K %d  %s  totals and group sums agree: %s" % (
        G, how, best, sites / best / 1e3, g_["threads"], g_["cols_per_thread"], g_["slices"], g_["rows_per_batch"],
        "dir" if rd.path()["directory_path"] else "plane" if rd.path()["plane_split"] else "scan", ok), flush=True)
