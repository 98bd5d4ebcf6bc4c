#!/usr/bin/env python3
"""Sub-checkpoint spacing (BGTH_SUB_SHIFT = 8 ... 11, i.e. sub-blocks of 256 ... 2048 rows) against scan time, for images of
different lengths (GPU box).  A sub-block x column slice is the unit one workgroup decodes: a short image at the default
spacing of 2048 rows does not fill the 256 CUs.  usage: python scripts/subshift_ab.py [samples:sites,...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys
sys.path.insert(0, %r)
import bgt_amd
n, sites = int(sys.argv[1]), int(sys.argv[2])
m = 2 * n
rle, lens = bgt_amd.synth_rows(m, 0, sites, 7)
pbf = bgt_amd.HipPbf.from_rle(m, 13, rle, lens)
rd = bgt_amd.HipReader(pbf)
bgt_amd.force_kernels(int("128"))
rd.scan(0, sites)
best = min((rd.scan(0, sites), rd.timing()["scan_ms"])[1] for _ in range(5))
g = rd.geometry()
print("%%.3f ms  %%d thr x %%d col x %%d slices K %%d  %%s" %% (best, g["threads"], g["cols_per_thread"], g["slices"], g["rows_per_batch"],
      "dir" if rd.path()["directory_path"] else "plane" if rd.path()["plane_split"] else "scan"))
""" % ROOT

shapes = sys.argv[1] if len(sys.argv) > 1 else "32488:142000,10000:1000000,2504:50000,10000:142000,100000:142000,50000:60000"
for sh in shapes.split(","):
    n, sites = (int(x) for x in sh.split(":"))
    for s in os.environ.get("AB_SHIFTS", "auto,11,10,9,8").split(","):
        env = dict(os.environ)
        env.pop("BGTH_SUB_SHIFT", None)
        if s != "auto":
            env["BGTH_SUB_SHIFT"] = s
        out = subprocess.run([sys.executable, "-c", CHILD, str(n), str(sites)], env=env, capture_output=True, text=True)
        print("%7d samples x %8d sites  sub_shift %-4s  %s" % (n, sites, s, (out.stdout.strip() or out.stderr.strip()[-300:])), flush=True)
