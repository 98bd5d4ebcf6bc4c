#!/usr/bin/env python3
"""Extract a sub-cohort of a .pbf into a new .pbf without leaving the device formats: the reader's 2-bit genotype rows of
the selected haplotypes go straight into the writer (bgth_reader_scan -> bgth_encoder_write_packed).  With the reference
this is `bgt view -s ... -b` followed by `bgt import` of the BCF.
usage: python scripts/subset_pbf.py in.pbf out.pbf <first-sample> <n-samples> [chunk-rows]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bgt_amd  # noqa: E402


def subset_pbf(src_path, dst_path, cols, chunk=65536):
    """cols: haplotype columns of the source (2 s, 2 s + 1 of sample s), ascending"""
    pbf = bgt_amd.HipPbf.open(src_path)
    rd = bgt_amd.HipReader(pbf)
    rd.select(cols=np.asarray(cols, np.int32))
    enc = bgt_amd.HipEncoder(len(cols), pbf.g, pbf.shift)
    total = 0
    with open(dst_path, "wb") as fp:
        for r0 in range(0, pbf.n, chunk):
            _, gt = rd.scan(r0, min(pbf.n, r0 + chunk), want_gt=True)
            enc.write_packed(gt)
            total += fp.write(enc.take())                     # streamed: the image is never held whole
        total += fp.write(enc.finish())
    enc.close()
    return total


if __name__ == "__main__":
    if len(sys.argv) < 5:
        sys.exit(__doc__)
    s0, ns = int(sys.argv[3]), int(sys.argv[4])
    n = subset_pbf(sys.argv[1], sys.argv[2], np.arange(2 * s0, 2 * (s0 + ns)), int(sys.argv[5]) if len(sys.argv) > 5 else 65536)
    print("%s: %d bytes" % (sys.argv[2], n))
