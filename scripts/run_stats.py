#!/usr/bin/env python3
"""Run structure of the synthetic cohort's plane-rows (CPU only): toggles per row, directory trips without a toggle, and the
probes a bucketed run-table lookup would need (wave-max over 64 random ranks).  usage: python scripts/run_stats.py [m] [sites] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bgt_amd  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
sites = int(sys.argv[2]) if len(sys.argv) > 2 else 400
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rle, lens = bgt_amd.synth_rows(m, 0, sites, seed)
tab = np.array([c if c < 16 else (c & 15) << (4 * (c >> 4)) for c in range(128)], np.int64)
rng = np.random.default_rng(1)
off, res, tog, boring = 0, {}, {0: [], 1: []}, {0: [], 1: []}
for i, l in enumerate(lens):
    b = rle[off:off + l]; off += int(l)
    ln, bit = tab[b >> 1], b & 1
    starts = np.concatenate([[0], np.cumsum(ln)[:-1]])
    prev = np.concatenate([[0], bit[:-1]])
    tg = np.sort(starts[(bit != prev) & (ln > 0)])
    tog[i & 1].append(len(tg))
    ntrip = (m + 8191) // 8192
    boring[i & 1].append(int((np.bincount((tg >> 13).astype(int), minlength=ntrip)[:ntrip] == 0).sum()))
    for bs in (5, 6, 7, 8):
        for _ in range(8):
            r = rng.integers(0, m, 64)
            k = np.searchsorted(tg, r, "right") - np.searchsorted(tg, (r >> bs) << bs, "left")
            res.setdefault((i & 1, bs), []).append((k.max(), k.mean()))
print("m = %d, %d sites, seed %d" % (m, sites, seed))
for p in (0, 1):
    print("plane %d: toggles per row mean %.0f median %.0f; directory trips (8192 positions) without a toggle: %.2f of %d"
          % (p, np.mean(tog[p]), np.median(tog[p]), np.mean(boring[p]), (m + 8191) // 8192))
for key in sorted(res):
    a = np.array(res[key])
    print("plane %d, buckets of %3d positions: probes per wave (max over 64 lanes) mean %.2f p90 %.0f p99 %.0f max %.0f; per lane mean %.3f"
          % (key[0], 1 << key[1], a[:, 0].mean(), np.percentile(a[:, 0], 90), np.percentile(a[:, 0], 99), a[:, 0].max(), a[:, 1].mean()))
