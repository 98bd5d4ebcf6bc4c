#!/usr/bin/env python3
"""Sample subsets: the automatic kernel family against the plane-split kernels forced (BGTH_VARIANT 4096) and forbidden (2048),
and the directory path forced (1024).  usage: python scripts/subset_ab.py [samples:sites:every,...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bgt_amd  # noqa: E402

shapes = sys.argv[1] if len(sys.argv) > 1 else "32488:142000:13,32488:142000:4,32488:142000:2,100000:262144:20,100000:262144:5,50000:262144:10,10000:1000000:10"
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
    for label, var in (("auto", 0), ("plane-split forced", 4096), ("plane-split never", 2048)):
        bgt_amd.force_kernels(int(str(128 + var)))
        try:
            rd.scan(0, min(sites, 8192))
            best = 1e9
            for _ in range(3):
                counts = rd.scan(0, sites)
                best = min(best, rd.timing()["scan_ms"])
        except Exception as e:                                    # (a forced family that has no geometry for this shape)
            print("%s %-20s failed: %s" % (sh, label, str(e)[:120]), flush=True)
            continue
        res.append(counts)
        g, p = rd.geometry(), rd.path()
        print("%-22s %-20s %8.3f ms  %4d thr x %2d col x %d slices K %d  %s" % (sh, label, best, g["threads"], g["cols_per_thread"], g["slices"], g["rows_per_batch"],
              "dir" if p["directory_path"] else "plane" if p["plane_split"] else "scan"), flush=True)
    print("   same counts:", all(np.array_equal(res[0], r) for r in res[1:]), flush=True)
    bgt_amd.force_kernels(0)
    rd.close(); pbf.close()
