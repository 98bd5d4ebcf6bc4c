#!/usr/bin/env python3
"""C3 plane-split timing under different launch conditions (GPU box)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bgt_amd
n_samples, sites, seed = 100000, 1000000, 3
m = 2 * n_samples
rle, lens = bgt_amd.synth_rows(m, 0, sites, seed)
pbf = bgt_amd.HipPbf.from_rle(m, 13, rle, lens)
rd = bgt_amd.HipReader(pbf)
sel = np.arange(0, n_samples, 20)
rd.select(np.stack([2 * sel, 2 * sel + 1], 1).reshape(-1))
dev = torch.device("cuda", 0)
out = torch.empty((sites, 1, 3), dtype=torch.int32, device=dev)
def t(label, stream, sync_each):
    for i in range(4):
        rd.scan_device(0, sites, out.data_ptr(), stream=stream)
        if sync_each:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(label, rd.timing(), rd.path()["plane_split"], flush=True)
t("reader stream, sync each", None, True)
t("torch current (null) stream, sync each", torch.cuda.current_stream().cuda_stream, True)
t("torch current (null) stream, back to back", torch.cuda.current_stream().cuda_stream, False)
s2 = torch.cuda.Stream()
t("torch side stream, back to back", s2.cuda_stream, False)
c = rd.scan(0, sites)
print("host scan", rd.timing())
sys.path.insert(0, os.path.join(ROOT))
import bench
pipe = bench.Pipeline(torch, bgt_amd, rd, 0, sites, dev, 0, 1, 0, None)
dt, k_ms, last = pipe.run(5, 2)
print("Pipeline: ms/step", dt / 5 * 1e3, "kernel", k_ms, flush=True)
# without the side-stream copies
class P2(bench.Pipeline):
    def step(self):
        b = self.done & 1
        self.done += 1
        self.rd.scan_device(self.row0, self.row1, self.counts[b].data_ptr(), stream=self.main.cuda_stream)
        self.n_pass_d[b].zero_()
        self.flt.apply_device(self.counts[b].data_ptr(), self.n, 3, self.flags[b].data_ptr(), self.n_pass_d[b].data_ptr(), self.main.cuda_stream)
        return b
p2 = P2(torch, bgt_amd, rd, 0, sites, dev, 0, 1, 0, None)
dt, k_ms, last = p2.run(5, 2)
print("no side copies: ms/step", dt / 5 * 1e3, "kernel", k_ms, flush=True)
class P3(bench.Pipeline):
    def step(self):
        b = self.done & 1
        self.done += 1
        self.rd.scan_device(self.row0, self.row1, self.counts[b].data_ptr(), stream=self.main.cuda_stream)
        return b
p3 = P3(torch, bgt_amd, rd, 0, sites, dev, 0, 1, 0, None)
dt, k_ms, last = p3.run(5, 2)
print("scan only: ms/step", dt / 5 * 1e3, "kernel", k_ms, flush=True)


def variant(label, extra):
    class P(bench.Pipeline):
        def step(self):
            b = self.done & 1
            self.done += 1
            self.rd.scan_device(self.row0, self.row1, self.counts[b].data_ptr(), stream=self.main.cuda_stream)
            extra(self, b)
            return b
    p = P(torch, bgt_amd, rd, 0, sites, dev, 0, 1, 0, None)
    dt, k_ms, last = p.run(5, 2)
    print("%-40s ms/step %.2f kernel %.2f" % (label, dt / 5 * 1e3, k_ms), flush=True)
variant("scan + counts.add_(0) [12 MB rw]", lambda self, b: self.counts[b].add_(0))
variant("scan + flags.zero_() [1 MB]", lambda self, b: self.flags[b].zero_())
variant("scan + counts.sum()", lambda self, b: self.counts[b].sum())
big = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
variant("scan + 64 MB fill", lambda self, b: big.zero_())
variant("scan + filter", lambda self, b: self.flt.apply_device(self.counts[b].data_ptr(), self.n, 3, self.flags[b].data_ptr(), self.n_pass_d[b].data_ptr(), self.main.cuda_stream))
