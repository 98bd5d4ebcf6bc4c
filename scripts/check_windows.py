#!/usr/bin/env python3
"""Oracle windows at several places of a big synthetic cohort (GPU box): python scripts/check_windows.py samples sites seed every"""
import os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bgt_amd
import bench

n_samples, sites, seed, every = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
m = 2 * n_samples
rle, lens = bgt_amd.synth_rows(m, 0, sites, seed)
pbf = bgt_amd.HipPbf.from_rle(m, 13, rle, lens)
rd = bgt_amd.HipReader(pbf)
cols = None
if every > 1:
    sel = np.arange(0, n_samples, every)
    cols = np.stack([2 * sel, 2 * sel + 1], 1).reshape(-1)
    rd.select(cols)
got = rd.scan(0, sites)
print("scan done", rd.geometry(), rd.path(), flush=True)
with tempfile.TemporaryDirectory() as tmp:
    for frac in (0.0, 0.1, 0.25, 0.5, 0.75, 0.95):
        mid = max(8192, int(sites * frac) // 8192 * 8192)
        if mid + 512 > sites:
            continue
        oc, t = bench.oracle_window(bgt_amd, np, pbf, m, 13, seed, mid - 2048, mid - 2048, 2560, tmp, 0, cols)
        g = got[mid - 2048: mid + 512]
        bad = np.nonzero((oc != g).any(axis=(1, 2)))[0]
        print("window at %d: %s" % (mid, "ok" if bad.size == 0 else "MISMATCH first at +%d (%d rows differ) oracle %s gpu %s" % (bad[0], bad.size, oc[bad[0]].tolist(), g[bad[0]].tolist())), flush=True)
