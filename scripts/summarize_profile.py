#!/usr/bin/env python3
"""Condense a scripts/profile.sh run (gpurun_out/prof_<tag>) into profiles/<tag>/ (committed evidence):
kernel_stats.csv (rocprofv3 --kernel-trace --stats), pmc_summary.json (mean per launch of the bench-size
scan kernel for every PMC pass), fetch_calibration.json and profiles/traffic_latest.json (HBM bytes per
launch = FETCH_SIZE scaled by the calibration + WRITE_SIZE)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else "c2"          # what bench.py was run with under rocprofv3
sites = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles", tag)
os.makedirs(dst, exist_ok=True)
def newest(pattern):
    """gpurun merges a call's files INTO gpurun_out/: a tag profiled twice holds both runs' files side by side (rocprofv3 names
    them by process id).  Per directory only the newest counts."""
    by_dir = {}
    for f in glob.glob(pattern):
        d = os.path.dirname(f)
        if d not in by_dir or os.path.getmtime(f) > os.path.getmtime(by_dir[d]):
            by_dir[d] = f
    return sorted(by_dir.values())


for f in newest(os.path.join(src, "trace", "*", "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, "kernel_stats.csv"))
stats = list(csv.DictReader(open(os.path.join(dst, "kernel_stats.csv"))))
# the bench-size launches: scripts/profile.sh runs 1 warm-up + 3 timed steps = 4 calls (other scan_kernel
# instances are the one-block passes that derive the synthetic cohort's checkpoints during set-up)
scan = [r for r in stats if "scan_kernel" in r["Name"] or "walk_kernel" in r["Name"] or "plane_kernel<" in r["Name"]]
# (the trace pass runs TRACE_STEPS + 5 calls, 25 by default; older runs 4: the instantiation called that often, else the most-called)
want = [r for r in scan if int(r["Calls"]) in (4, int(os.environ.get("TRACE_STEPS", "20")) + 5)]
main = max(want or scan, key=lambda r: (int(r["Calls"]), float(r["AverageNs"])))
kname = main["Name"]
out = {"kernel": kname, "calls": int(main["Calls"]), "avg_ms": float(main["AverageNs"]) / 1e6, "counters": {}}
# the directory path's producer (rows built once into the HBM arena), when the profiled scans ran it
prod = [r for r in stats if "dirbuild_" in r["Name"] and int(r["Calls"]) == int(main["Calls"])] or [r for r in stats if "dirbuild_" in r["Name"]]
pname = prod[0]["Name"] if prod else None
if prod:
    out["producer"] = {"kernel": pname, "calls": int(prod[0]["Calls"]), "avg_ms": float(prod[0]["AverageNs"]) / 1e6, "counters": {}}
for f in newest(os.path.join(src, "pmc_*", "*", "*counter_collection.csv")):
    agg, pagg = collections.defaultdict(list), collections.defaultdict(list)
    big = out["avg_ms"]
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"] == kname:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        elif pname and r["Kernel_Name"] == pname:
            pagg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in agg.items():
        if len(v) > 4:                                     # smaller launches of the same instantiation (set-up passes, checks): the
            v = sorted(v)[-4:]                             # four bench-size ones carry the largest counts
        out["counters"][c] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
    for c, v in pagg.items():
        if len(v) > 4:
            v = sorted(v)[-4:]
        out["producer"]["counters"][c] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
calib = {}
for w in (4, 16):
    fs = newest(os.path.join(src, "calib_w%d" % w, "*", "*counter_collection.csv"))
    vals = [float(r["Counter_Value"]) for f in fs for r in csv.DictReader(open(f))
            if "stream_read" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
    if vals:
        mean = sum(vals) / len(vals)
        calib["width%d" % w] = {"bytes_streamed": 1 << 30, "FETCH_SIZE_KiB": mean,
                                "bytes_per_counted_byte": (1 << 30) / (mean * 1024.0)}
json.dump(calib, open(os.path.join(dst, "fetch_calibration.json"), "w"), indent=1)
c = out["counters"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    # the scan kernels read with 4-byte loads, the walk-only kernel pulls its rows with 16-byte LDS-DMA pieces
    k = calib.get("width16" if "walk_kernel" in kname else "width4", {}).get("bytes_per_counted_byte", 1.0)
    fetch = c["FETCH_SIZE"]["mean_per_launch"] * 1024.0 * k
    write = c["WRITE_SIZE"]["mean_per_launch"] * 1024.0
    out["hbm_bytes_per_launch"] = {"fetch_corrected": fetch, "write": write, "total": fetch + write,
                                   "fetch_scale_from_calibration": k}
    meta = json.load(open(os.path.join(root, "gpurun_out", "bench_for_%s.json" % tag))) if os.path.exists(
        os.path.join(root, "gpurun_out", "bench_for_%s.json" % tag)) else {}
    traffic = {"tag": tag, "workload": workload, "sites": sites, "hbm_bytes_per_launch": fetch + write,
               "fetch_bytes": fetch, "write_bytes": write, "fetch_scale": k, "kernel": kname}
    pc = out.get("producer", {}).get("counters", {})
    if "FETCH_SIZE" in pc and "WRITE_SIZE" in pc:
        k4 = calib.get("width4", {}).get("bytes_per_counted_byte", 1.0)
        traffic["producer"] = {"kernel": pname, "fetch_bytes": pc["FETCH_SIZE"]["mean_per_launch"] * 1024.0 * k4,
                               "write_bytes": pc["WRITE_SIZE"]["mean_per_launch"] * 1024.0, "avg_ms": out["producer"]["avg_ms"]}
    json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    if workload == "c2" and sites == 1000000:        # the headline configuration: what bench.py reports as roofline.traffic
        json.dump(traffic, open(os.path.join(root, "profiles", "traffic_latest.json"), "w"), indent=1)
json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
