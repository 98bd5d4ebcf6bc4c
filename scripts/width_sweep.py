#!/usr/bin/env python3
"""Whole-cohort throughput by width on the shipped kernels (GPU box): one long scan per width with the automatic geometry,
the roofline fraction against the two ceilings of bench.py.  Writes gpurun_out/width_sweep.md / .json (the table of DESIGN.md).
usage: python scripts/width_sweep.py [samples,samples,...]"""
import datetime
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bgt_amd  # noqa: E402
import bench  # noqa: E402

widths = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2504, 5000, 10000, 17000, 25000, 32488, 35000, 50000, 100000]
peak = bench.lookup_peak(bgt_amd, 0)
rows = []
for n in widths:
    m = 2 * n
    sites = 1048576 if m <= 20000 else 524288 if m <= 70000 else 262144
    t0 = time.time()
    rle, lens = bgt_amd.synth_rows(m, 0, sites, 2)
    pbf = bgt_amd.HipPbf.from_rle(m, 13, rle, lens)
    del rle
    rd = bgt_amd.HipReader(pbf)
    bgt_amd.force_kernels(int(str(128 + int(os.environ.get("SWEEP_VARIANT", "0")))))   # 128: every scan builds its rows (no arena carried over)
    rd.scan(0, min(sites, 16384))
    best = 1e9
    for _ in range(3):
        rd.scan(0, sites)
        best = min(best, rd.timing()["scan_ms"])
    bgt_amd.force_kernels(0)
    g, p = rd.geometry(), rd.path()
    look = 2.0 * m * sites / (best * 1e-3) / 1e9
    kind = "directory path (producer + walk-only)" if p["directory_path"] else "plane-split" if p["plane_split"] else \
        ("pipelined" if 4 * g["rows_per_batch"] > g["threads"] // 64 and g["slices"] == 1 and m <= 50000 else "team")   # (a wave builds its plane-rows alone / teams of waves per plane-row)
    rows.append({"samples": n, "haplotypes": m, "sites": sites, "ms": best, "sites_per_s": sites / best * 1e3, "g_lookups_per_s": look,
                 "frac_of_ideal_mix": look / peak["ideal_mix_g_lookups_per_s"], "frac_of_own_statement": look / peak["g_lookups_per_s"],
                 "launch": "%d thr x %d col x %d slices, K %d" % (g["threads"], g["cols_per_thread"], g["slices"], g["rows_per_batch"]), "kernels": kind})
    print(rows[-1], "setup %.1fs" % (time.time() - t0), flush=True)
    rd.close(); pbf.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
stamp = datetime.date.today().isoformat()
md = ["Whole-cohort scans by width (scripts/width_sweep.py, %s, one MI355X, one-shot: every scan builds its rows; ceilings measured live: "
      "ideal mix %.2f T, own statement %.2f T lookups/s)" % (stamp, peak["ideal_mix_g_lookups_per_s"] / 1e3, peak["g_lookups_per_s"] / 1e3), "",
      "| samples (haplotypes) | sites | launch | kernels | ms | M sites/s | T lookups/s | of the ideal-mix ceiling | of the own statement |", "|---|---|---|---|---|---|---|---|---|"]
for r in rows:
    md.append("| %d (%d) | %d | %s | %s | %.2f | %.1f | %.2f | %.2f | %.2f |" % (r["samples"], r["haplotypes"], r["sites"], r["launch"], r["kernels"], r["ms"],
                                                                          r["sites_per_s"] / 1e6, r["g_lookups_per_s"] / 1e3, r["frac_of_ideal_mix"], r["frac_of_own_statement"]))
open(os.path.join(ROOT, "gpurun_out", "width_sweep.md"), "w").write("\n".join(md) + "\n")
json.dump({"date": stamp, "peak": peak, "rows": rows}, open(os.path.join(ROOT, "gpurun_out", "width_sweep.json"), "w"), indent=1)
print("\n".join(md))
