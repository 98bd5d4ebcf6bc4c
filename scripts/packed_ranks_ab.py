#!/usr/bin/env python3
"""Round 6 A/B: whole-cohort counts with a column's two ranks packed in one register (step4pk, 15 VALU
instructions per column) against the shipped statement (16), IN THE KERNEL, same run, alternating.  The packed statement lost
(profiles/r06_pk16) and is compiled into the PROFILING build only:
    make -C bgt_amd/csrc ABLATE=1 && BGT_AMD_LIB=bgt_amd/lib/libbgt_hip_ablate.so python scripts/packed_ranks_ab.py [samples:sites,...]
(kernel time by HIP events, best of 7; counts compared)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bgt_amd  # noqa: E402

shapes = sys.argv[1] if len(sys.argv) > 1 else "10000:1000000,2504:1000000,5000:1000000,20000:524288,32488:142000"
for sh in shapes.split(","):
    samples, sites = (int(x) for x in sh.split(":"))
    m = 2 * samples
    rle, lens = bgt_amd.synth_rows(m, 0, sites, 2)
    pbf = bgt_amd.HipPbf.from_rle(m, 13, rle, lens)
    del rle
    rd = bgt_amd.HipReader(pbf)
    res = []
    for label, var in (("two registers", 0), ("packed", 32768), ("two registers", 0), ("packed", 32768)):
        os.environ["BGTH_VARIANT"] = str(var)                     # (read by the profiling build at every scan)
        bgt_amd.force_kernels(bgt_amd.hip.FORCE_REBUILD_ROWS)
        rd.scan(0, min(sites, 8192))
        best, times = 1e9, []
        for _ in range(7):
            counts = rd.scan(0, sites)
            times.append(rd.timing()["scan_ms"])
        res.append(counts)
        g, p = rd.geometry(), rd.path()
        print("%-16s %-14s best %8.3f ms  median %8.3f   %4d thr x %2d col x %d  K %d  %s" % (sh, label, min(times), sorted(times)[3], g["threads"], g["cols_per_thread"],
              g["slices"], g["rows_per_batch"], "dir" if p["directory_path"] else "plane" if p["plane_split"] else "scan"), flush=True)
    print("   same counts:", all(np.array_equal(res[0], r) for r in res[1:]), flush=True)
    bgt_amd.force_kernels(0)
    rd.close(); pbf.close()
