#!/usr/bin/env python3
"""C3-shaped scan through bgth_reader_scan_device on torch's NULL stream, on a created stream, and through the library's own
stream (bgth_reader_scan), several launches each: where a launch of the plane-split kernels runs is not where it is timed.
usage: python scripts/plane_stream_ab.py [samples] [sites] [every]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bgt_amd  # noqa: E402

samples = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
sites = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
sub = int(sys.argv[3]) if len(sys.argv) > 3 else 20
m = 2 * samples
rle, lens = bgt_amd.synth_rows(m, 0, sites, 3)
pbf = bgt_amd.HipPbf.from_rle(m, 13, rle, lens)
rd = bgt_amd.HipReader(pbf)
s = np.arange(0, samples, sub)
rd.select(np.stack([2 * s, 2 * s + 1], 1).reshape(-1))
dev = torch.device("cuda", 0)
counts = torch.empty((sites, 1, 3), dtype=torch.int32, device=dev)
side = torch.cuda.Stream(device=dev)
for label, stream in (("library stream (scan)", None), ("NULL stream", 0), ("created stream", side.cuda_stream), ("NULL stream", 0), ("library stream (scan)", None)):
    ts = []
    for i in range(6):
        if stream is None:
            rd.scan(0, sites)
        else:
            rd.scan_device(0, sites, counts.data_ptr(), stream=stream)
            torch.cuda.synchronize()
        ts.append(rd.timing()["scan_ms"])
    print("%-24s %s  geometry %d thr x %d" % (label, " ".join("%.2f" % t for t in ts), rd.geometry()["threads"], rd.geometry()["cols_per_thread"]), flush=True)
# back to back without a synchronisation in between (what bench.py's pipeline does), with and without a copy of the counts
# to pinned host memory on a second stream beside the next scan
host = torch.empty((sites, 1, 3), dtype=torch.int32).pin_memory()
for label, copy in (("back to back", False), ("back to back + D2H beside", True), ("back to back", False)):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    main = torch.cuda.current_stream()
    e0.record(main)
    for i in range(6):
        rd.scan_device(0, sites, counts.data_ptr(), stream=0)
        if copy:
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                host.copy_(counts, non_blocking=True)
    e1.record(main)
    torch.cuda.synchronize()
    print("%-28s %.2f ms per scan (last kernel %.2f)" % (label, e0.elapsed_time(e1) / 6, rd.timing()["scan_ms"]), flush=True)
