#!/usr/bin/env python3
"""Round 6 experiment (profiling build): the plane-split kernel of a sparse selection with TEN waves per workgroup instead of
twelve where ten hold the selection (C3: 160 chunk slots for 157 chunks instead of 192).
    make -C bgt_amd/csrc ABLATE=1 && BGT_AMD_LIB=bgt_amd/lib/libbgt_hip_ablate.so python scripts/plane_threads_ab.py [samples:sites:every,...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bgt_amd  # noqa: E402

shapes = sys.argv[1] if len(sys.argv) > 1 else "100000:1000000:20,100000:262144:16"
for sh in shapes.split(","):
    samples, sites, sub = (int(x) for x in sh.split(":"))
    m = 2 * samples
    rle, lens = bgt_amd.synth_rows(m, 0, sites, 3)
    pbf = bgt_amd.HipPbf.from_rle(m, 13, rle, lens)
    del rle
    s = np.arange(0, samples, sub)
    cols = np.stack([2 * s, 2 * s + 1], 1).reshape(-1)
    res = []
    for label, thr in (("768 threads", "768"), ("640 threads", "640"), ("768 threads", "768"), ("640 threads", "640")):
        os.environ["BGTH_PLANE_THREADS"] = thr                   # (read when a reader's geometry is chosen: profiling build)
        rd = bgt_amd.HipReader(pbf)
        rd.select(cols)
        rd.scan(0, min(sites, 8192))
        times = []
        for _ in range(5):
            counts = rd.scan(0, sites)
            times.append(rd.timing()["scan_ms"])
        res.append(counts)
        g = rd.geometry()
        print("%-22s %-12s best %8.3f ms  median %8.3f   %4d thr x %2d col  %s" % (sh, label, min(times), sorted(times)[2], g["threads"], g["cols_per_thread"],
              "plane" if rd.path()["plane_split"] else "other"), flush=True)
        rd.close()
    print("   same counts:", all(np.array_equal(res[0], r) for r in res[1:]), flush=True)
    pbf.close()
