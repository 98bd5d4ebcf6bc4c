#!/usr/bin/env python3
"""Genotype rows of a mid-file window: GPU (big image) vs oracle (window image).  GPU box."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bgt_amd, orc

n_samples, sites, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
m = 2 * n_samples
rle, lens = bgt_amd.synth_rows(m, 0, sites, seed)
pbf = bgt_amd.HipPbf.from_rle(m, 13, rle, lens)
rd = bgt_amd.HipReader(pbf)
mid = (sites // 2) // 8192 * 8192
a = mid - 2048
n = 64
c, g = rd.scan(a, a + n, want_gt=True)
wr, wl = bgt_amd.synth_rows(m, a, n, seed)
w = bgt_amd.HipPbf.from_rle(m, 13, wr, wl)
w.rebase(pbf.ranks_at(a))
with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "w.pbf")
    w.save(path)
    data = open(path, "rb").read()
oc, og = orc.Pbf(data).scan(0, n, want_gt=True)
print("counts equal", np.array_equal(oc.reshape(n, 1, 3), c), "gt equal", np.array_equal(og, g))
if not np.array_equal(og, g):
    def unpack(x): return np.stack([(x >> (2 * k)) & 3 for k in range(4)], -1).reshape(x.shape[0], -1)[:, :m]
    A, B = unpack(og), unpack(g)
    d = (A != B)
    print("rows differing", d.any(1).sum(), "cells differing in row 0:", d[0].sum(), "cols", np.nonzero(d[0])[0][:20])
    print("row0 code histogram oracle", np.bincount(A[0], minlength=4), "gpu", np.bincount(B[0], minlength=4))
# the same window decoded by the GPU itself from the window image
wrd = bgt_amd.HipReader(w)
c2, g2 = wrd.scan(0, n, want_gt=True)
print("window image on GPU == big image on GPU:", np.array_equal(c2, c), np.array_equal(g2, g))
# big image with sequentially derived checkpoints
bgt_amd.force_kernels(int("512"))
pbf2 = bgt_amd.HipPbf.from_rle(m, 13, rle, lens)
bgt_amd.force_kernels(0)
rd2 = bgt_amd.HipReader(pbf2)
c3, g3 = rd2.scan(a, a + n, want_gt=True)
print("sequential checkpoints == parallel checkpoints:", np.array_equal(c3, c), np.array_equal(g3, g), "ranks equal", np.array_equal(pbf2.ranks_at(a), pbf.ranks_at(a)))
# subset: counts from the genotype matrix restricted to the selected columns vs the subset scans of GPU and oracle
def unpack(x): return np.stack([(x >> (2 * k)) & 3 for k in range(4)], -1).reshape(x.shape[0], -1)[:, :m]
M = unpack(g)
sel = np.arange(0, n_samples, 20)
cols = np.stack([2 * sel, 2 * sel + 1], 1).reshape(-1)
sub = M[:, cols]
want = np.stack([(sub != 2).sum(1), (sub == 1).sum(1), (sub == 3).sum(1)], 1).reshape(n, 1, 3)
rd.select(cols)
gs = rd.scan(a, a + n)
op = orc.Pbf(data); op.subset(cols)
os_ = op.scan(0, n).reshape(n, 1, 3)
print("subset: GPU == from-matrix", np.array_equal(gs, want), " oracle == from-matrix", np.array_equal(os_, want))
if not np.array_equal(os_, want):
    bad = np.nonzero((os_ != want).any(axis=(1, 2)))[0]
    print("oracle subset differs in", bad.size, "rows; first", bad[0], os_[bad[0]].tolist(), want[bad[0]].tolist())
