#!/usr/bin/env python3
"""Round 6: what the kernels beyond the LDS widths cost.  A cohort of 500,000 samples (1,000,000 haplotypes) x `sites` rows is
written by the device writer (encode_huge_kernel: directories in memory), saved, opened and scanned by the device reader
(dirbuild_mem_kernel + walk_mem_kernel: toggles and directory entries in memory); the CPU oracle decodes the first rows of the
same file for comparison and for a rate.  usage: python scripts/wide_cohort_times.py [samples] [sites]"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import bgt_amd  # noqa: E402
import orc  # noqa: E402  (checker / CPU rate only)

n_samples = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
sites = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
m, K, per_call = 2 * n_samples, 24, 1024
rng = np.random.default_rng(6)
founder_of = rng.integers(0, K, m).astype(np.int64)
enc = bgt_amd.HipEncoder(m, 2, 11)
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "wide.pbf")
t_enc = 0.0
with open(path, "wb") as f:
    for r0 in range(0, sites, per_call):
        n = min(per_call, sites - r0)
        freq = np.where(rng.random(n) < 0.5, 1.0 / rng.integers(2, 202, n), rng.random(n) * 0.5)
        chunk = np.take((rng.random((n, K)) < freq[:, None]).astype(np.uint8), founder_of, axis=1)   # (C-contiguous; F[:, idx] comes back transposed)
        rr, cc = rng.integers(0, n, 4000), rng.integers(0, m, 4000)
        chunk[rr[:3000], cc[:3000]] = 2
        chunk[rr[3000:], cc[3000:]] = 3
        t0 = time.perf_counter()
        enc.write(chunk)
        f.write(enc.take())
        t_enc += time.perf_counter() - t0
        founder_of[rng.integers(0, m, 3000)] = rng.integers(0, K, 3000)
    f.write(enc.finish())
print("writer: %d rows x %d columns: device kernels %.1f ms (%.0f rows/s), write() calls incl. PCIe and records %.2f s; file %.1f MB" %
      (sites, m, enc.kernel_ms, sites / (enc.kernel_ms * 1e-3), t_enc, os.path.getsize(path) / 1e6), flush=True)
enc.close()
t0 = time.perf_counter()
pbf = bgt_amd.HipPbf.open(path)
t_open = time.perf_counter() - t0
rd = bgt_amd.HipReader(pbf)
best = 1e9
for _ in range(3):
    counts = rd.scan(0, sites)
    best = min(best, rd.timing()["total_ms"])
p = rd.path()
print("reader: open %.2f s; whole-cohort scan of %d sites %.1f ms (%.0f sites/s; producer %.1f ms of it; %s)" %
      (t_open, sites, best, sites / (best * 1e-3), p["producer_ms"], rd.geometry()), flush=True)
sel = np.arange(0, n_samples, 100)
rd.select(np.stack([2 * sel, 2 * sel + 1], 1).reshape(-1).astype(np.int32))
best = 1e9
for _ in range(3):
    sub = rd.scan(0, sites)
    best = min(best, rd.timing()["total_ms"])
print("reader: every 100th sample (%d samples): %.1f ms (%.0f sites/s)" % (sel.size, best, sites / (best * 1e-3)), flush=True)
ora = orc.Pbf(np.fromfile(path, np.uint8))
n_or = 200
t0 = time.perf_counter()
oc = ora.scan(0, n_or)
t_or = time.perf_counter() - t0
print("oracle (one CPU core): %d sites in %.2f s = %.0f sites/s; counts equal: %s" % (n_or, t_or, n_or / t_or, bool(np.array_equal(oc.reshape(n_or, 1, 3), counts[:n_or]))))
rd.close(); pbf.close()
os.remove(path)
