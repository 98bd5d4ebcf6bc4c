#!/usr/bin/env python3
"""Why the C3 scan of bench.py's pipeline differs from the same scan alone: bench.Pipeline on the C3 selection with its
parts switched off one by one.  usage: python scripts/plane_bench_probe.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bgt_amd  # noqa: E402
import bench  # noqa: E402

samples, sites, sub = 100000, 1000000, 20
m = 2 * samples
rle, lens = bgt_amd.synth_rows(m, 0, sites, 3)
pbf = bgt_amd.HipPbf.from_rle(m, 13, rle, lens)
rd = bgt_amd.HipReader(pbf)
s = np.arange(0, samples, sub)
rd.select(np.stack([2 * s, 2 * s + 1], 1).reshape(-1))
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def run(label, copies=True, flt=True, fresh=True):
    pipe = bench.Pipeline(torch, bgt_amd, rd, 0, sites, dev, 0, 1, 0, None)
    if not copies:
        pipe.rank = 1                                        # (rank 0 alone copies to the host)
    if not flt:
        pipe.flt.apply_device = lambda *a, **k: None
    res = []
    for _ in range(3):
        dt, k_ms, last = pipe.run(6, 2)
        res.append("%.2f/%.2f" % (dt / 6 * 1e3, k_ms))
    print("%-44s ms per step / last kernel: %s" % (label, "  ".join(res)), flush=True)
    del pipe


def run_sync(label, sync):
    pipe = bench.Pipeline(torch, bgt_amd, rd, 0, sites, dev, 0, 1, 0, None)
    res = []
    for _ in range(8):
        pipe.step()
        if sync:
            torch.cuda.synchronize()
        res.append("%.2f" % rd.timing()["scan_ms"] if sync else "-")
    torch.cuda.synchronize()
    print("%-44s kernel ms: %s (last %.2f)" % (label, " ".join(res), rd.timing()["scan_ms"]), flush=True)
    del pipe


for rep in range(2):
    run("pipeline")
    run_sync("steps with a device synchronisation between", True)
    run_sync("steps enqueued back to back", False)
