#!/usr/bin/env python3
"""What the genotype planes cost a scan: counts only against counts + bit planes (the input of bgt_gen_gt's vector / text).
usage: python scripts/gt_ab.py [samples] [sites]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bgt_amd  # noqa: E402

samples = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
sites = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
m = 2 * samples
rle, lens = bgt_amd.synth_rows(m, 0, sites, 2)
pbf = bgt_amd.HipPbf.from_rle(m, 13, rle, lens)
rd = bgt_amd.HipReader(pbf)
bgt_amd.force_kernels(int("128"))
for want in (False, True, False, True):
    rd.scan(0, min(sites, 8192), want_gt=want)
    best = 1e9
    for _ in range(3):
        rd.scan(0, sites, want_gt=want)
        best = min(best, rd.timing()["scan_ms"])
    g = rd.geometry()
    print("want_gt=%-5s m=%d sites=%d: %8.3f ms  %7.2f M sites/s  %d thr x %d col x %d slices K %d %s" % (
        want, m, sites, best, sites / best / 1e3, g["threads"], g["cols_per_thread"], g["slices"], g["rows_per_batch"],
        "dir" if rd.path()["directory_path"] else "plane" if rd.path()["plane_split"] else "scan"), flush=True)
