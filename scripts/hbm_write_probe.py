#!/usr/bin/env python3
"""What does this MI355X sustain for plain HBM writes?  (The directory path's producer writes 100 KB per row at 4.1 TB/s:
how far is that from a fill?)  torch fill / copy of 8 GiB, HIP events."""
import torch
n = 8 << 30
x = torch.empty(n, dtype=torch.uint8, device="cuda")
y = torch.empty(n, dtype=torch.uint8, device="cuda")
def t(f, reps=5):
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best
ms = t(lambda: x.zero_())
print("fill  8 GiB: %.3f ms  %.2f TB/s written" % (ms, n / ms / 1e9))
ms = t(lambda: x.view(torch.int32).fill_(7))
print("fill32 8 GiB: %.3f ms  %.2f TB/s written" % (ms, n / ms / 1e9))
ms = t(lambda: y.copy_(x))
print("copy  8 GiB: %.3f ms  %.2f TB/s read + %.2f TB/s written" % (ms, n / ms / 1e9, n / ms / 1e9))
ms = t(lambda: x.view(torch.int32).sum())
print("read  8 GiB: %.3f ms  %.2f TB/s read" % (ms, n / ms / 1e9))
