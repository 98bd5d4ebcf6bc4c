#!/usr/bin/env python3
"""does the LDS bank-conflict share of the random ds_read_b64 gather cost the row step anything?  The product's own
8-lookup statement alone on every SIMD at 4 waves per SIMD with (a) random entries -- what the kernels see today --, (b) 64
consecutive ranks per wave and lookup on a row of long runs -- what slots in checkpoint-rank order would see --, (c) one distinct
consecutive entry per lane (conflict-free, no broadcast).  GPU box; prints a markdown table.
usage: python scripts/lds_conflict_ab.py [iters]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bgt_amd  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000      # (clustered ranks drift apart over tens of thousands of steps)
L = bgt_amd.bench_lib()
L.bgth_debug_issue_rate_name.restype = C.c_char_p
print("| mix | waves/SIMD | cycles per VALU instr | G lookups/s | LDS cycles per ds_read_b64 and CU |")
print("|---|---|---|---|---|")
for mix in (7, 14, 6, 10, 9):
    for waves in (2, 4):
        best = None
        for _ in range(3):
            out = (C.c_double * 4)()
            if L.bgth_debug_issue_rate(0, mix, waves, iters, out) != 0:
                raise SystemExit("bgth_debug_issue_rate failed")
            if best is None or out[0] < best[0]:
                best = list(out)
        cyc, ms, valu, lds = best
        lookups = 256.0 * (4 * waves) * 64 * (valu / 8.0) if mix in (6, 7, 14) else 256.0 * (4 * waves) * 64 * lds
        print("| %s | %d | %.2f | %.0f | %.2f |" % (L.bgth_debug_issue_rate_name(mix).decode(), waves, cyc / (waves * valu),
                                                  lookups / (ms * 1e-3) / 1e9, cyc / (4 * waves * lds) * 4 if lds else float("nan")))
