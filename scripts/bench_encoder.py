#!/usr/bin/env python3
"""Throughput of the device writer (bgth_encoder_*) at a C2-shaped width: the rows come from the synthetic cohort
(decoded on the device), go through the encoder, and the image is checked against the oracle writer on a prefix and
by scanning it back.  The CPU oracle writer (a port of pbf_write/pbc_enc) is timed beside it.
usage: python scripts/bench_encoder.py [--samples 10000] [--rows 65536]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import bgt_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, default=10000)
ap.add_argument("--rows", type=int, default=65536)
ap.add_argument("--seed", type=int, default=2)
ap.add_argument("--cpu-rows", type=int, default=2048)
args = ap.parse_args()
m, rows, shift = 2 * args.samples, args.rows, 13

rle, lens = bgt_amd.synth_rows(m, 0, rows, args.seed)
src = bgt_amd.HipPbf.from_rle(m, shift, rle, lens)
rd = bgt_amd.HipReader(src)
counts, gt = rd.scan(0, rows, want_gt=True)
codes = np.empty((rows, m), np.uint8)
for k in range(4):
    codes[:, k::4] = ((gt >> (2 * k)) & 3)[:, :(m - k + 3) // 4]

enc = bgt_amd.HipEncoder(m, 2, shift)
t0 = time.time()
enc.write(codes)
image = enc.finish()
wall = time.time() - t0
kernel_s = enc.kernel_ms / 1e3

encp = bgt_amd.HipEncoder(m, 2, shift)                       # the same rows as the reader hands them out: four columns to a byte
t0 = time.time()
encp.write_packed(gt)
image_p = encp.finish()
wall_packed = time.time() - t0
assert image_p == image, "packed input: a different image"
encp.close()

back = bgt_amd.HipReader(bgt_amd.HipPbf.from_bytes(image))
assert np.array_equal(back.scan(0, rows), counts), "the encoded image does not scan back to the same counts"

import orc  # noqa: E402  (checker + CPU baseline)
n_cpu = min(args.cpu_rows, rows)
ref = orc.encode_pbf(codes[:n_cpu], 2, shift)
# the CPU baseline: the oracle writer alone, its g byte arrays per row prepared beforehand (what import.c hands over)
import ctypes as C  # noqa: E402
p0 = np.ascontiguousarray(codes[:n_cpu] & 1)
p1 = np.ascontiguousarray(codes[:n_cpu] >> 1)
w = orc.lib.orc_pbw_new(m, 2, shift)
planes = (orc.u8p * 2)()
a0, a1 = p0.ctypes.data, p1.ctypes.data
t0 = time.time()
for r in range(n_cpu):
    planes[0] = C.cast(a0 + r * m, orc.u8p)
    planes[1] = C.cast(a1 + r * m, orc.u8p)
    orc.lib.orc_pbw_row(w, planes)
cpu_s = time.time() - t0
out = orc.u8p()
n_out = orc.lib.orc_pbw_finish(w, C.byref(out))
assert C.string_at(out, n_out) == ref
enc2 = bgt_amd.HipEncoder(m, 2, shift)
enc2.write(codes[:n_cpu])
assert enc2.finish() == ref, "device image differs from the oracle writer"

print(json.dumps({"metric": "rows/sec pbf_write (encode)", "columns": m, "rows": rows, "image_bytes": len(image),
                  "rows_per_s_kernel": rows / kernel_s, "rows_per_s_wall_incl_upload_and_assembly": rows / wall,
                  "rows_per_s_wall_packed_2bit_input": rows / wall_packed,
                  "kernel_us_per_row": 1e6 * kernel_s / rows,
                  "cpu_oracle_rows_per_s": n_cpu / cpu_s, "cpu_rows": n_cpu,
                  "parity": "image == oracle writer on the first %d rows; whole image scans back to the input counts" % n_cpu}))
