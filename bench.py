#!/usr/bin/env python3
"""bench.py -- sites/sec of the `bgt view -G -f'AC>0'` whole-cohort scan on MI355X.

One "step" = one pass of the hot path over the rank's whole shard: PBWT run-length decode of both bit
planes of every site, rank-tracking column reconstruction, AC/AN reduction (all on the GPU, through the
C ABI of libbgt_hip.so), the site filter AC>0 on the device, counts + pass flags delivered to the host.  Inputs (RLE
strings, row directory, checkpoints) are resident in HBM before the timed region starts.

  N = 1   headline: workload C2 of BASELINE.json, synthetic 10,000 samples (m = 20,000 haplotypes) x 1,000,000 sites;
          then `secondary` records at 100,000 samples (the north-star width): C3 (1,000,000 sites, every 20th sample =
          10,000 tracked columns) and one C4 shard (153 file blocks = 1,253,376 sites, whole cohort), each with its own
          roofline and a CPU baseline from the compiled reference on the first sites of the same database.
  N > 1   headline: the SAME per-GPU workload, weak scaling (the driver compares this value with the N = 1 value): rank r holds
          sites [r, r+1) x 1,000,000 of ONE C2-width database of N x 1,000,000 sites (site-range sharding, no data-path
          collective), then an all_gather over RCCL/xGMI of the per-shard counts and flags.  Every rank builds its shard
          from the identity order, the shards' final ranks are exchanged and each shard is re-based onto the composition
          of the earlier ones (bgth_pbf_rebase), so the gathered counts are those of the one database.  Rank 0 checks them
          after the timed region: the plane-popcount identity on every site of every shard and a CPU-oracle window that
          runs across the boundary between shard 0 and shard 1 (`parity_ok`).
          secondary "C4-sharded" (same run, same checks): BASELINE configs[3] itself, strong scaling -- the 1,221 file
          blocks of ONE database of 100,000 samples x 10,000,000 sites split over the N ranks (153 blocks per rank at
          N = 8).  `--workload c4` makes it the headline (its N = 1 point: `bench.py --workload c4`).
          On failure the failing rank prints one JSON line with the error, its rank and device.

Prints one JSON line (rank 0).

roofline: the scan kernel keeps the permutation in registers, so it moves ~0.1 % of the reference algorithm's bytes and
its bound is the issue side, not HBM (the north star's ">= 50 % of HBM bandwidth" does not apply: roofline.note).  `bound` =
"valu_issue"; `unit` = rank lookups per second (one lookup = one tracked column x one bit plane x one site: the LF-mapping
step); `achieved` = algorithmic lookups of the launch / its duration (HIP events).  `peak` (and `frac`) is the CODE-INDEPENDENT
ceiling: the minimal 8-instruction lookup (5 instructions of the 4-cycle class, 3 of the 2-cycle class) priced with class
rates measured live in this run, as if the classes added up and nothing else ever issued.  `peak_own_statement` /
`frac_of_own_statement` = what the product's own row-step statement (8 VALU + 1 ds_read_b64 per lookup, random LDS entries,
SALU counts) sustains alone on every SIMD at 4 waves per SIMD, also measured live (bgth_debug_issue_rate) -- round 4 showed
that this, not the class sum, is what the hardware gives a mixed instruction stream (profiles/r04_issue/README.md).
`algorithmic_equiv_gbs` keeps SURVEY 8d's figure (reference-algorithm bytes / kernel time); `counters` = rocprofv3 PMC numbers
of the same kernel replayed from profiles/ (marked so); `traffic` of the headline record is measured IN THE RUN: bench.py
re-executes itself under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` (inrun_counters).
`resident_end_to_end` = the metric's command line like for like with cpu_baseline (open, scan, filter, format, write every
passing line) with the image resident in a `bgt-server -u` host; `cli_end_to_end` = the same from a cold process.
`cpu_baseline` = the compiled reference (oracle/_ref/bgt) timed on this box's host cores on a bounded sample.
"""
import argparse
import datetime
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SAMPLES = {"c2": 10000, "c3": 100000, "c4": 100000, "small": 2504, "hrc": 32488}   # hrc: the width of the reference's published numbers (HRC r1)
SEEDS = {"c2": 2, "c3": 3, "c4": 4, "small": 1, "hrc": 7}
HBM_PEAK_GBS = 8000.0                      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "bgt")
MY_BIN = os.path.join(ROOT, "bgt_amd", "bin", "bgt")


GUIDE_VALU_CYCLES = 2.0                    # MI355X_MICROARCH.md: one VALU wave-instruction per 2 cycles per SIMD (SIMD-32 halves)
COMPACT_LIMIT = 4000                       # bytes of the final stdout line (the driver parses that line only)


def _num(x, digits=4):
    """A finite float rounded to `digits` significant digits, an int as it is, anything else None (strict JSON: no NaN / Infinity)."""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, int):
        return x
    try:
        f = float(x)
    except (TypeError, ValueError):
        return None
    if f != f or f in (float("inf"), float("-inf")):
        return None
    if f == 0.0:
        return 0.0
    from math import floor, log10
    return round(f, max(0, digits - 1 - int(floor(log10(abs(f))))))


def compact_roofline(r):
    """Numbers only.  `peak` / `frac` = the micro-architecture guide's plain VALU issue (one wave-instruction per 2 cycles and SIMD,
    8 instructions per lookup): a hardware number anyone can recompute from kernel_ms, lookups_per_launch and clock_ghz.  Secondary,
    named: the class sum measured live and the product's own statement alone on the chip.  HBM: measured traffic per launch beside
    the compulsory floor (strings + descriptors + checkpoints of the tracked columns + counts)."""
    if not r:
        return None
    c = {"bound": r.get("bound"), "achieved": _num(r.get("achieved")), "peak": _num(r.get("peak")), "unit": r.get("unit"),
         "frac": _num(r.get("frac")), "peak_is": "1024 SIMDs x 64 lanes x clock_ghz / (8 VALU x 2 cycles) (MI355X_MICROARCH.md)",
         "clock_ghz": _num(r.get("peak_clock_ghz")),
         "peak_class_sum": _num(r.get("peak_class_sum")), "frac_of_class_sum": _num(r.get("frac_of_class_sum")),
         "peak_own_statement": _num(r.get("peak_own_statement")), "frac_of_own_statement": _num(r.get("frac_of_own_statement")),
         "traffic": _num(r.get("traffic"), 6), "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x calibration + WRITE_SIZE)",
         "traffic_in_run": bool(r.get("traffic_in_run", {}).get("hbm_bytes_per_launch")) if isinstance(r.get("traffic_in_run"), dict) else False,
         "hbm_floor_bytes": _num(r.get("hbm_floor_bytes"), 6),
         "traffic_over_floor": _num(r["traffic"] / r["hbm_floor_bytes"], 3) if r.get("traffic") and r.get("hbm_floor_bytes") else None,
         "hbm_frac_measured": _num(r.get("hbm_frac_measured")), "hbm_peak_gbs": r.get("hbm_peak_gbs"),
         "kernel": r.get("kernel"), "kernel_ms": _num(r.get("kernel_ms")), "lookups_per_launch": _num(r.get("lookups_per_launch"), 6),
         "algorithmic_equiv_gbs": _num(r.get("algorithmic_equiv_gbs"))}
    cn = r.get("counters") or {}
    if cn.get("lds_busy") is not None:
        c["lds_busy_profiled"] = _num(cn.get("lds_busy"), 3)
        c["lds_conflict_frac_profiled"] = _num(cn.get("lds_conflict_frac"), 3)
    return c


def compact_record(out, detail_path=None):
    """The ONE line the driver reads (last line of stdout): < COMPACT_LIMIT bytes, strict JSON (allow_nan=False), numbers and short
    labels only.  Everything else of `out` goes to the detail file; the prose that used to sit in the record is DESIGN.md 4-5."""
    if out.get("value") is None or "error" in out:
        c = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "error", "failed_rank", "failed_local_rank") if k in out}
        c["error"] = str(c.get("error", ""))[:600]
        return json.dumps(c, allow_nan=False)
    cfg = out.get("config", {})
    geo = cfg.get("launch") or {}
    c = {"metric": out["metric"], "value": _num(out["value"], 6), "unit": out["unit"], "n_gpus": out["n_gpus"], "steps": out["steps"],
         "warmup": out["warmup"], "ms_per_step": _num(out["ms_per_step"], 5), "higher_is_better": True, "scaling": out.get("scaling") or "weak",
         "vs_baseline": None, "dtype": out.get("dtype"), "data": out.get("data"),
         "config": {"workload": cfg.get("workload"), "haplotypes": cfg.get("haplotypes"), "tracked_columns": cfg.get("tracked_columns"),
                    "sites_per_gpu": cfg.get("sites_per_gpu"), "sites_total": cfg.get("sites_total"), "sharding": cfg.get("sharding"),
                    "sites_passing_filter": cfg.get("sites_passing_filter"),
                    "launch": {k: geo.get(k) for k in ("threads", "cols_per_thread", "slices", "rows_per_batch", "workgroups") if k in geo}},
         "roofline": compact_roofline(out.get("roofline")), "parity_ok": out.get("parity_ok")}
    par = out.get("parity")
    if isinstance(par, dict):
        c["parity"] = {"popcount_identity_ok": par.get("popcount_identity_ok"),
                       "sites_checked_popcount_identity": par.get("sites_checked_popcount_identity"),
                       "oracle_window_matches": (par.get("oracle_window") or {}).get("matches"),
                       "oracle_window_rows": (par.get("oracle_window") or {}).get("rows")}
        frz = par.get("from_row_zero")
        if isinstance(frz, dict):
            c["parity"]["from_row_zero"] = {k: frz.get(k) for k in ("sites", "of_sites", "oracle_counts_match", "reference_stdout_identical")}
    cb = out.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = {"value": _num(cb.get("value")), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                             "sample": (cb.get("sample") or "")[:160],
                             "cli_stdout_identical_to_reference": cb.get("cli_stdout_identical_to_reference"),
                             "gpu_matches_cpu_on_sample": cb.get("gpu_matches_cpu_on_sample")}
        if isinstance(cb.get("all_cores"), dict) and cb["all_cores"].get("value"):
            c["cpu_baseline"]["all_cores"] = {"value": _num(cb["all_cores"]["value"]), "processes": cb["all_cores"].get("processes")}
        if "error" in cb:
            c["cpu_baseline"] = {"error": str(cb["error"])[:200]}
    for key in ("resident_end_to_end", "cli_end_to_end"):                       # the command line, like for like with cpu_baseline
        e = out.get(key)
        if isinstance(e, dict) and e.get("wall_s"):
            c[key] = {"wall_s": _num(e["wall_s"]), "sites_per_s": _num(e.get("sites_per_s")), "output_lines": e.get("output_lines")}
            if e.get("vs_cpu_baseline"):
                c[key]["vs_cpu_baseline"] = _num(e["vs_cpu_baseline"])
    ps = out.get("product_sharded")
    if isinstance(ps, dict):
        if "error" in ps:
            c["product_sharded"] = {"error": str(ps["error"])[:200], "devices": ps.get("devices")}
        else:
            bv = ps.get("bgt_view") or {}
            c["product_sharded"] = {"devices": ps.get("devices"), "ms_per_step": _num(ps.get("ms_per_step")), "sites_per_s": _num(ps.get("sites_per_s")),
                                    "per_shard_kernel_ms": [_num(x) for x in ps.get("per_shard_kernel_ms", [])], "gather_ms": _num(ps.get("gather_ms")),
                                    "parity_ok": ps.get("parity_ok"),
                                    "bgt_view_wall_s": {k: _num((bv.get(k) or {}).get("wall_s")) for k in ("one_device", "sharded")},
                                    "bgt_view_stdout_identical": bv.get("stdout_identical")}
    if out.get("n_gpus", 1) > 1:
        c["per_rank_kernel_ms"] = [_num(x) for x in out.get("per_rank_kernel_ms", [])]
        c["gather_ms"] = _num(out.get("gather_ms"))
        c["ranks"] = {k: out.get("ranks", {}).get(k) for k in ("world_size", "backend", "device_of_rank", "devices_visible")}
    sec = []
    for s in out.get("secondary", []):
        if "error" in s and "sites_per_s" not in s and "commands" not in s:
            sec.append({"name": s.get("name"), "error": str(s["error"])[:120]})
        elif "commands" in s:                                                    # published commands through both binaries
            for q in s["commands"]:
                e = {"name": "%s: %s" % (s.get("name"), q.get("command")), "sites": s.get("sites"), "cold_s": _num(q.get("this_repo_s")),
                     "resident_s": _num(q.get("resident_s")), "reference_s": _num(q.get("reference_s")),
                     "published_s": q.get("published_s"), "stdout_identical": q.get("stdout_identical")}
                sec.append({k: v for k, v in e.items() if v is not None})
        elif "wall_s" in s:                                                      # a command line record (C5)
            sec.append({"name": s.get("name"), "wall_s": _num(s["wall_s"]), "reference_s": _num(s.get("reference", {}).get("wall_s")),
                        "stdout_identical": s.get("reference", {}).get("stdout_identical"),
                        "sharded_stdout_identical": s.get("sharded", {}).get("stdout_identical")})
        else:
            rf = s.get("roofline") or {}
            e = {"name": s.get("name"), "sites_per_s": _num(s.get("sites_per_s", s.get("value"))), "ms_per_step": _num(s.get("ms_per_step")),
                 "kernel_ms": _num(s.get("kernel_ms", rf.get("kernel_ms"))), "frac": _num(rf.get("frac"), 3),
                 "parity_ok": s.get("parity_ok"),
                 "cpu_baseline_sites_per_s": _num((s.get("cpu_baseline") or {}).get("value"))}
            if s.get("arena_kept"):
                e["arena_kept_ms"] = _num(s["arena_kept"].get("kernel_ms"))
            if s.get("n_gpus", 1) > 1:                                           # the block-sharded database riding along (N > 1)
                sp = s.get("parity") or {}
                e.update({"scaling": s.get("scaling"), "haplotypes": s.get("config", {}).get("haplotypes"),
                          "sites_total": s.get("config", {}).get("sites_total"),
                          "sites_checked_popcount_identity": sp.get("sites_checked_popcount_identity"),
                          "per_rank_kernel_ms": [_num(x) for x in s.get("per_rank_kernel_ms", [])], "gather_ms": _num(s.get("gather_ms"))})
            sec.append(e)
    if sec:
        c["secondary"] = sec
    if out.get("parity_error"):
        c["parity_error"] = str(out["parity_error"])[:300]
    if detail_path:
        c["detail"] = detail_path
    line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    while len(line) > COMPACT_LIMIT and c.get("secondary"):                      # never over the limit: drop records from the end, say so
        c["secondary"].pop()
        c["secondary_truncated"] = True
        line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    return line


def emit(out, detail_arg=None):
    """Full record -> the detail file (--detail PATH; default bench_detail.json at the repo root, and in gpurun_out/ where that
    exists so that it travels back from the GPU box); compact line -> stdout, last."""
    detail = None
    try:
        txt = json.dumps(out, indent=1, default=str)
        targets = [detail_arg] if detail_arg else [os.path.join(d, "bench_detail.json") for d in (ROOT, os.path.join(ROOT, "gpurun_out"))
                                                   if os.path.isdir(d)]
        for t in targets:
            with open(t, "w") as f:
                f.write(txt)
            detail = os.path.basename(t)
    except Exception:
        pass
    sys.stdout.flush()
    print(compact_record(out, detail), flush=True)


def guide_peak(peak):
    """G lookups/s at the guide's plain VALU issue: 1024 SIMDs x 64 lanes x clock / (8 instructions x 2 cycles)"""
    return 1024.0 * 64.0 * peak["clock_ghz"] / (8.0 * GUIDE_VALU_CYCLES)


def lookup_peak(bgt_amd, device):
    """G rank-lookups/s the chip sustains running only the product's row step (live microbenchmark, ~40 ms)."""
    import ctypes as C
    L = bgt_amd.bench_lib()                                                       # measurement tools: not in the product library
    out = (C.c_double * 4)()
    iters, waves = 20000, 4
    if L.bgth_debug_issue_rate(device, 7, waves, iters, out) != 0:            # mix 7: step4 + ds_read_b64, random entries
        raise RuntimeError("bgth_debug_issue_rate failed")
    cycles, ms, valu = out[0], out[1], out[2]
    lookups = 256.0 * (4 * waves) * 64 * (valu / 8.0)                           # CUs x waves x lanes x lookups per wave
    # class rates for the code-independent ceiling: v_add_u32 (the 2-cycle class) and v_bcnt_u32_b32 (the 4-cycle class
    # of the VOP3 integer forms), dependency-free streams at 4 waves per SIMD
    cls = {}
    for name, mix in (("c2", 1), ("c4", 3)):
        o = (C.c_double * 4)()
        if L.bgth_debug_issue_rate(device, mix, waves, iters, o) != 0:
            raise RuntimeError("bgth_debug_issue_rate failed")
        cls[name] = o[0] / (waves * o[2])
    clock_hz = cycles / (ms * 1e-3)
    ideal_cycles = 5.0 * cls["c4"] + 3.0 * cls["c2"]                            # v_mad_i32_i24 v_lshlrev v_bcnt v_cmp v_cndmask | v_ashrrev v_sub v_add
    return {"g_lookups_per_s": lookups / (ms * 1e-3) / 1e9, "cycles_per_valu_instr": cycles / (waves * valu),
            "ideal_mix_g_lookups_per_s": 1024.0 * 64.0 * clock_hz / ideal_cycles / 1e9,
            "class_cycles": {"two_cycle_class_v_add_u32": cls["c2"], "four_cycle_class_v_bcnt_u32_b32": cls["c4"],
                             "minimal_step_cycles_per_lookup_wave": ideal_cycles},
            "valu_instr_per_cycle_per_simd": waves * valu / cycles, "clock_ghz": cycles / (ms * 1e6), "ms": ms,
            "source": "live: bgth_debug_issue_rate(mix 7 = the scan kernel's own 8-lookup statement incl. ds_read_b64 on "
                      "random entries and SALU counts, 256 CUs x 4 waves/SIMD, %d iterations)" % iters}


def replayed_counters(workload, sites, kernel_name):
    """rocprofv3 PMC numbers of the same kernel on the same workload, collected by scripts/profile.sh and committed under
    profiles/ -- NOT measured in this run."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for tag in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        tp, pp = os.path.join(pdir, tag, "traffic.json"), os.path.join(pdir, tag, "pmc_summary.json")
        if not (os.path.exists(tp) and os.path.exists(pp)):
            continue
        try:
            tj, pj = json.load(open(tp)), json.load(open(pp))
        except Exception:
            continue
        if tj.get("workload") == workload and tj.get("sites") == sites and kernel_name in pj.get("kernel", ""):
            best = (tag, tj, pj)                                                   # latest tag wins (sorted)
    if best is None:
        return None
    tag, tj, pj = best
    c = {k: v["mean_per_launch"] for k, v in pj.get("counters", {}).items()}
    out = {"replayed_from": "profiles/%s (rocprofv3 --pmc passes of scripts/profile.sh on this workload; not measured in this run)" % tag,
           "profiled_kernel_ms": pj.get("avg_ms"), "hbm_bytes_per_launch": tj.get("hbm_bytes_per_launch"),
           "valu_instr_per_launch": c.get("SQ_INSTS_VALU")}
    if tj.get("producer"):                                                         # directory path, one-shot: the producer's launch too
        pr = tj["producer"]
        out["producer"] = {"kernel": pr.get("kernel", "")[:60], "profiled_kernel_ms": pr.get("avg_ms"),
                           "hbm_bytes_per_launch": pr.get("fetch_bytes", 0) + pr.get("write_bytes", 0),
                           "hbm_write_gbs": pr.get("write_bytes", 0) / (pr.get("avg_ms", 1e9) * 1e-3) / 1e9}
        out["hbm_bytes_per_launch_walk_only"] = out["hbm_bytes_per_launch"]
        out["hbm_bytes_per_launch"] = out["hbm_bytes_per_launch"] + out["producer"]["hbm_bytes_per_launch"]
        out["profiled_kernel_ms_walk_only"] = out["profiled_kernel_ms"]
        out["profiled_kernel_ms"] = out["profiled_kernel_ms"] + (pr.get("avg_ms") or 0.0)
    if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_INSTS_VALU"):
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0                                           # summed over the 8 XCDs
        out["valu_instr_per_cycle_per_simd"] = c["SQ_INSTS_VALU"] / (1024.0 * cyc)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            out["lds_busy"] = c["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc)
    if c.get("SQ_LDS_IDX_ACTIVE") and c.get("SQ_LDS_BANK_CONFLICT"):
        out["lds_conflict_frac"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    return out


def hbm_floor(pbf, T, sites, n_groups=1):
    """Compulsory HBM bytes of one launch over `sites` rows: the run-length strings (padded to dwords, as they sit in HBM), the row
    descriptors, one rank checkpoint of the T tracked columns and both planes per sub-block, and the counts written."""
    sub = -(-sites // int(pbf.unit_rows))
    parts = {"strings": float(pbf.rle_bytes) * sites / max(1, pbf.n), "row_descriptors": 16.0 * sites,
             "checkpoints": sub * 2.0 * T * 4.0, "counts": 12.0 * sites * (1 + (n_groups if n_groups > 1 else 0))}
    parts["total"] = sum(parts.values())
    return parts


def make_roofline(peak, T, sites, k_ms, rle_bytes_per_site, geo, workload, counters_sites, path=None, floor=None):
    lookups = 2.0 * T * sites                                                      # tracked columns x 2 planes x sites
    achieved = lookups / (k_ms * 1e-3) / 1e9
    alg_bytes_per_site = 16.0 * T + rle_bytes_per_site + 12.0
    if path and path.get("plane_split") and not path.get("directory_path"):
        kname = "plane_kernel<%d" % geo["cols_per_thread"]                        # scan_plane.hip: a workgroup per (sub-block, plane)
    else:
        kname = ("walk_kernel<%d, %d" if path and path.get("directory_path") else "scan_kernel<%d, %d") % (geo["threads"], geo["cols_per_thread"])
    guide = guide_peak(peak)
    # the self-calibrated ceilings come from micro-kernels timed live: where something else shares the device (the dry run of N > 1
    # on one GPU) they can come out absurd -- then they are left out rather than printed
    sane = lambda x: x if x and 0.1 * guide < x < 1.5 * guide else None
    pcs, pown = sane(peak["ideal_mix_g_lookups_per_s"]), sane(peak["g_lookups_per_s"])
    r = {"bound": "valu_issue", "achieved": achieved, "peak": guide, "unit": "G rank-lookups/s",
         "frac": achieved / guide, "traffic": None,
         "kernel": kname + (", ..., %d threads>" % geo["threads"] if kname.startswith("plane_kernel") else ", ...>"), "kernel_ms": k_ms, "lookups_per_launch": lookups,
         "peak_source": "hardware number: MI355X_MICROARCH.md's VALU issue, one wave64 instruction per %.0f cycles and SIMD, x 1024 SIMDs x 64 "
                        "lanes x the clock measured live (%.3f GHz) / the 8 VALU instructions of the minimal lookup" % (GUIDE_VALU_CYCLES, peak["clock_ghz"]),
         "peak_clock_ghz": peak["clock_ghz"],
         "peak_class_sum": pcs, "frac_of_class_sum": achieved / pcs if pcs else None,
         "peak_class_sum_source": "self-calibrated, secondary: 5 x four-cycle-class + 3 x two-cycle-class cycles for the minimal 8-instruction "
                                  "lookup, class rates measured live as single-instruction streams (%.2f / %.2f cycles per wave-instruction at 4 "
                                  "waves per SIMD; profiles/r04_issue)"
                                  % (peak["class_cycles"]["four_cycle_class_v_bcnt_u32_b32"], peak["class_cycles"]["two_cycle_class_v_add_u32"]),
         "peak_own_statement": pown, "frac_of_own_statement": achieved / pown if pown else None,
         "peak_own_statement_source": peak["source"], "peak_own_statement_cycles_per_valu_instr": peak["cycles_per_valu_instr"],
         "algorithmic_bytes_per_site": alg_bytes_per_site,
         "algorithmic_equiv_gbs": alg_bytes_per_site * sites / (k_ms * 1e-3) / 1e9,
         "hbm_peak_gbs": HBM_PEAK_GBS,
         "note": "HBM is NOT the bound and the north star's '>= 50 % of HBM read bandwidth' does not apply to this design: the "
                 "permutation stays in registers, so HBM carries only the RLE strings, row descriptors and checkpoints (hbm_floor_bytes; "
                 "traffic / hbm_floor_bytes says what is re-read).  algorithmic_equiv_gbs prices the REFERENCE algorithm's 16*T bytes "
                 "per site (SURVEY 8d) at this kernel's speed -- above the HBM peak, i.e. not moving the permutation beats moving it at full "
                 "HBM speed; it bounds nothing.  The bound is VALU issue."}
    if floor:
        r["hbm_floor_bytes"] = floor["total"]
        r["hbm_floor_parts"] = floor
    rc = replayed_counters(workload, counters_sites, kname)
    if rc:
        r["counters"] = rc
        r["traffic"] = rc.get("hbm_bytes_per_launch")
        r["traffic_replayed_from"] = rc["replayed_from"]
        if rc.get("hbm_bytes_per_launch") and rc.get("profiled_kernel_ms"):
            r["hbm_frac_measured"] = rc["hbm_bytes_per_launch"] / (rc["profiled_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        if rc.get("valu_instr_per_cycle_per_simd"):
            r["valu_issue_frac_profiled"] = rc["valu_instr_per_cycle_per_simd"] / peak["valu_instr_per_cycle_per_simd"]   # (of the own statement's rate)
        if rc.get("valu_instr_per_launch"):
            # share of the kernel's VALU instructions that are the lookups' 8-instruction steps: the rest builds the rows'
            # rank directories, which a sparse selection of a wide cohort cannot amortise (C3: 10,000 of 200,000 columns)
            r["lookup_share_of_valu_instr"] = lookups * 8.0 / 64.0 / rc["valu_instr_per_launch"]
    return r



def plane_ones(np, rle, lens, chunk_strings=1 << 16):
    """Ones of every RLE string (rows x 2 planes) straight from the bytes (reference pbwt.c:12-21: len = (code & 15) <<
    4 (code >> 4), bit = byte & 1) -- independent of any permutation state."""
    b = np.arange(256, dtype=np.int64)
    tab = np.where(b & 1, ((b >> 1) & 15) << (4 * (b >> 5)), 0)
    lens64 = lens.astype(np.int64)
    off = np.concatenate([[0], np.cumsum(lens64)])
    out = np.zeros(lens.size, np.int64)
    for a in range(0, lens.size, chunk_strings):
        z = min(lens.size, a + chunk_strings)
        seg = tab[rle[off[a]:off[z]]]
        if seg.size:
            starts = off[a:z] - off[a]
            tot = np.add.reduceat(seg, np.minimum(starts, seg.size - 1))
            tot[lens64[a:z] == 0] = 0
            out[a:z] = tot
    return out.reshape(-1, 2)


def popcount_identity(np, counts, ones, m):
    """counts int32[n][3] = {AN, AC, AC<M>} of a whole-cohort scan against the strings' ones: n(1) + n(3) = ones(plane 0),
    n(2) + n(3) = ones(plane 1), n(2) = m - AN."""
    c = counts.reshape(-1, 3).astype(np.int64)
    return bool(np.array_equal(c[:, 1] + c[:, 2], ones[:, 0]) and np.array_equal((m - c[:, 0]) + c[:, 2], ones[:, 1]))


def oracle_window(bgt_amd, np, img, m, shift, seed, abs_row, img_row, n_rows, tmp, device, cols=None, group=None, n_groups=1):
    """CPU-oracle counts of cohort rows [abs_row, abs_row + n_rows): the strings are drawn again (the generator is
    addressable by row), built into a small image from the identity order, re-based onto the ranks the big image `img`
    holds before its row img_row (= the same cohort row), saved as a .pbf and decoded by the oracle from that 'S' record
    on, sequentially -- across every block / shard boundary inside the window."""
    import orc                                                    # CPU oracle: checker only
    # Subset decoding (reference pbwt.c:340-388) takes its order from an 'S' record only when it SEEKS more than a block
    # ahead (pbwt.c:349-372: nearer targets are reached by decoding forward) -- a reader at row 0 assumes the identity
    # order, as every real file has there.  With a column subset the window therefore gets a lead of two 2048-row blocks:
    # the oracle seeks past them, picks up the order from the third block's 'S' record and carries its own tracked ranks
    # from there on (it never reloads them while reading on).
    lead = 4096 if cols is not None else 0
    assert img_row >= lead and abs_row >= lead
    rle, lens = bgt_amd.synth_rows(m, abs_row - lead, lead + n_rows, seed)
    wshift = 11 if lead else max(shift, int(n_rows - 1).bit_length())   # whole cohort: ONE block, no 'S' record inside the window,
    small = bgt_amd.HipPbf.from_rle(m, wshift, rle, lens, device=device)   # so the oracle carries its own order across the boundaries
    small.rebase(img.ranks_at(img_row - lead))
    path = os.path.join(tmp, "window_%d.pbf" % abs_row)
    small.save(path)
    small.close()
    ora = orc.Pbf(open(path, "rb").read())
    os.remove(path)
    if cols is not None:
        ora.subset(cols)
    t0 = time.perf_counter()
    oc = ora.scan(lead, lead + n_rows, group=group, n_groups=n_groups)
    return oc.reshape(n_rows, -1, 3), time.perf_counter() - t0



def inrun_counters(n_samples, sites, seed, tmp, kernel_substr, producer_substr=None, child_env=None):
    """HBM traffic of the scan kernel measured IN THIS RUN: this script starts itself twice under `rocprofv3 --pmc` (one
    counter per pass, as MI355X_MICROARCH.md prescribes; no tracing domains) in a child mode that only builds the same
    cohort and scans it four times, and reads the counters of the bench-size launches from the CSVs.  FETCH_SIZE counts
    KiB at half rate on gfx950 for this kernel's 4-byte loads: the factor is the committed calibration on a 1 GiB stream
    (profiles/*/fetch_calibration.json, 2.000)."""
    import csv
    import glob
    import shutil
    if shutil.which("rocprofv3") is None:
        return {"error": "rocprofv3 not on PATH"}
    out = {"source": "in-run: bench.py re-executed under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on this box (child mode: same "
                     "cohort, four scans; the three largest launches of the kernel averaged)", "kernel": kernel_substr}
    scale = 2.0
    try:
        for tag in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
            fc = os.path.join(ROOT, "profiles", tag, "fetch_calibration.json")
            if os.path.exists(fc):
                scale = json.load(open(fc)).get("width4", {}).get("bytes_per_counted_byte", 2.0)
                out["fetch_scale"] = {"value": scale, "from": "profiles/%s/fetch_calibration.json" % tag}
                break
    except Exception:
        pass
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(tmp, "pmc_%s_%d" % (counter, n_samples))
        cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
               "--counters-child", "%d,%d,%d" % (n_samples, sites, seed)]
        try:
            subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600, check=True,
                           env=dict(os.environ, TMPDIR=tmp, **(child_env or {})), cwd=tmp)
        except Exception as e:
            return {"error": "rocprofv3 pass for %s failed: %s" % (counter, repr(e)[:120])}
        vals, pvals = [], []
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != counter:
                    continue
                if kernel_substr in r["Kernel_Name"]:
                    vals.append(float(r["Counter_Value"]))
                elif producer_substr and producer_substr in r["Kernel_Name"]:
                    pvals.append(float(r["Counter_Value"]))
        if not vals:
            return {"error": "no %s rows for %s" % (counter, kernel_substr)}
        top = sorted(vals)[-3:]
        out[counter + "_KiB_per_launch"] = sum(top) / len(top)
        if pvals:
            top = sorted(pvals)[-3:]
            out.setdefault("producer", {"kernel": producer_substr})[counter + "_KiB_per_launch"] = sum(top) / len(top)
    out["fetch_bytes"] = out["FETCH_SIZE_KiB_per_launch"] * 1024.0 * scale
    out["write_bytes"] = out["WRITE_SIZE_KiB_per_launch"] * 1024.0
    out["hbm_bytes_per_launch"] = out["fetch_bytes"] + out["write_bytes"]
    pr = out.get("producer")
    if pr and "FETCH_SIZE_KiB_per_launch" in pr and "WRITE_SIZE_KiB_per_launch" in pr:     # directory path, one-shot: + the producer
        pr["fetch_bytes"] = pr["FETCH_SIZE_KiB_per_launch"] * 1024.0 * scale
        pr["write_bytes"] = pr["WRITE_SIZE_KiB_per_launch"] * 1024.0
        out["hbm_bytes_per_launch_walk_only"] = out["hbm_bytes_per_launch"]
        out["hbm_bytes_per_launch"] += pr["fetch_bytes"] + pr["write_bytes"]
    return out


def counters_child(spec):
    """Child mode of inrun_counters: nothing but the cohort and four scans (no checks, no output)."""
    import bgt_amd
    n_samples, sites, seed = (int(x) for x in spec.split(","))
    m = 2 * n_samples
    rle, lens = bgt_amd.synth_rows(m, 0, sites, seed)
    pbf = bgt_amd.HipPbf.from_rle(m, 13, rle, lens)
    rd = bgt_amd.HipReader(pbf)
    if os.environ.get("BENCH_REBUILD_ROWS"):                     # directory path: every scan builds its rows, as the timed steps do
        bgt_amd.force_kernels(bgt_amd.hip.FORCE_REBUILD_ROWS)
    for _ in range(4):
        rd.scan(0, sites)

def reference_cli_baseline(n_samples, ns, seed, view_args, tmp, what, all_cores=False):
    """The compiled reference's `bgt view` on a database of the first `ns` sites of the cohort; also this repo's CLI on
    the same command (stdout compared)."""
    __import__("bgt_amd").build_host_shell()
    prefix = os.path.join(tmp, "full_%d_%d" % (n_samples, ns))         # (cli_end_to_end names the full database the same way)
    if not os.path.exists(prefix + ".pbf"):
        subprocess.check_call([MY_BIN, "synth", prefix, str(n_samples), str(ns), str(seed)])
    cmd = ["view"] + view_args + [prefix]
    t0 = time.perf_counter()
    ref_out = subprocess.run([REF_BIN] + cmd, stdout=subprocess.PIPE, check=True).stdout
    t_ref = time.perf_counter() - t0
    t0 = time.perf_counter()
    my_out = subprocess.run([MY_BIN] + cmd, stdout=subprocess.PIPE, check=True).stdout
    t_mine = time.perf_counter() - t0
    same = hashlib.md5(ref_out).hexdigest() == hashlib.md5(my_out).hexdigest()
    out = {"value": ns / t_ref, "unit": "sites/s", "cores": 1, "kind": "reference",
           "sample": "reference `bgt view %s` (oracle/_ref/bgt, gcc -O2, one thread, %.1f s wall incl. open) on a %d-sample x "
                     "%d-site database = first sites of %s" % (" ".join(view_args), t_ref, n_samples, ns, what),
           "cli_stdout_identical_to_reference": same, "cli_stdout_bytes": len(ref_out),
           "this_repo_cli_same_command_s": round(t_mine, 2)}
    if all_cores:
        # the reference is single-threaded; the whole box = one process per 8192-site block range (disjoint -r
        # ranges), as many at a time as there are cores
        try:
            n_blk = (ns + 8191) // 8192
            procs = min(n_blk, os.cpu_count() or 1)
            t0 = time.perf_counter()
            running = []
            for k in range(n_blk):
                reg = "11:%d-%d" % (1000 + 10 * k * 8192, 1000 + 10 * min(ns, (k + 1) * 8192) - 1)
                running.append(subprocess.Popen([REF_BIN, "view"] + view_args + ["-r", reg, prefix], stdout=subprocess.DEVNULL))
                if len(running) >= procs:
                    running.pop(0).wait()
            for pr in running:
                pr.wait()
            t_all = time.perf_counter() - t0
            out["all_cores"] = {"value": ns / t_all, "unit": "sites/s", "processes": procs,
                                "sample": "%d reference processes over disjoint 8192-site regions, %.1f s wall" % (n_blk, t_all)}
        except Exception as e:
            out["all_cores"] = {"error": repr(e)[:120]}
    return out, same


def cli_end_to_end(n_samples, sites, seed, tmp):
    """The metric's own command line on the FULL configuration through this repo's `bgt view`: process start, HIP
    runtime, .pbf map + parse + upload + sub-checkpoints, site table, one device scan, filter, VCF text of every passing
    site -- the wall time a user sees (output to /dev/null; the stdout of a shorter database is compared with the
    reference binary in cpu_baseline)."""
    __import__("bgt_amd").build_host_shell()
    prefix = os.path.join(tmp, "full_%d_%d" % (n_samples, sites))
    t0 = time.perf_counter()
    if not os.path.exists(prefix + ".pbf"):                            # (the CPU baseline at full length has written it already)
        subprocess.check_call([MY_BIN, "synth", prefix, str(n_samples), str(sites), str(seed)])
    t_synth = time.perf_counter() - t0
    best, stages, lines = None, None, 0
    for _ in range(3):
        t0 = time.perf_counter()
        pr = subprocess.run([MY_BIN, "view", "-G", "-f", "AC>0", prefix], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                            env=dict(os.environ, BGT_TRACE="1", BGTH_TRACE="1"), check=True)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best = dt
            lines = pr.stdout.count(b"\n")
            stages = {}
            for ln in pr.stderr.decode().splitlines():
                mt = re.match(r"^\[bgth? trace\]\s+(.*?)\s+([0-9.]+) ms\b", ln)     # (other trace lines carry no time; an epoch stamp may follow)
                if mt:
                    stages[mt.group(1).strip()] = round(stages.get(mt.group(1).strip(), 0.0) + float(mt.group(2)), 2)
    return {"command": "bgt view -G -f 'AC>0' <prefix> (stdout to a pipe)", "sites": sites, "samples": n_samples,
            "wall_s": round(best, 3), "sites_per_s": sites / best, "output_lines": lines, "stages_ms": stages,
            "database_written_in_s": round(t_synth, 2),
            "note": "best of 3 process runs; stages_ms from BGT_TRACE / BGTH_TRACE (library stages are nested inside "
                    "'prepare'; refills summed); the HIP runtime start-up alone is 60-220 ms of every process"}


def resident_end_to_end(prefix, sites, cpu_baseline_sites_per_s):
    """The metric's command line LIKE FOR LIKE with cpu_baseline (a process that opens the database, scans, filters, formats and
    writes every passing site line) but with the image resident: `bgt-server -u SOCKET <prefix>` holds the database in HBM,
    `BGT_SERVER=SOCKET bgt view -G -f 'AC>0' <prefix>` hands it the query with its own stdout (a pipe here) and leaves with its
    status.  Process start to last byte, best of 5; the line count must be the local run's."""
    srv_bin = os.path.join(ROOT, "bgt_amd", "bin", "bgt-server")
    sock = os.path.join(os.path.dirname(prefix), "bgt.sock")
    srv = subprocess.Popen([srv_bin, "-u", sock, prefix], stderr=subprocess.DEVNULL)
    try:
        t0 = time.perf_counter()
        while not os.path.exists(sock):
            if srv.poll() is not None or time.perf_counter() - t0 > 120:
                raise RuntimeError("bgt-server -u did not come up")
            time.sleep(0.02)
        env = dict(os.environ, BGT_SERVER=sock)
        best, lines, small = None, 0, None
        for _ in range(6):
            t0 = time.perf_counter()
            o = subprocess.run([MY_BIN, "view", "-G", "-f", "AC>0", prefix], stdout=subprocess.PIPE, env=env, check=True).stdout
            dt = time.perf_counter() - t0
            if best is None or dt < best:
                best, lines = dt, o.count(b"\n")
        mid = 1000 + 10 * (sites // 2)
        for _ in range(5):                                        # a small region query through the same path
            t0 = time.perf_counter()
            o = subprocess.run([MY_BIN, "view", "-G", "-C", "-r", "11:%d-%d" % (mid, mid + 1179), prefix], stdout=subprocess.PIPE, env=env, check=True).stdout
            dt = time.perf_counter() - t0
            small = dt if small is None or dt < small else small
        rec = {"command": "BGT_SERVER=<socket> bgt view -G -f 'AC>0' <prefix> (stdout to a pipe; bgt-server -u <socket> holds the image in HBM)",
               "wall_s": round(best, 4), "sites_per_s": sites / best, "output_lines": lines,
               "region_query_118_sites_ms": round(small * 1e3, 2),
               "note": "scan + device filter + format + write of every passing line, the client process included; best of 6"}
        if cpu_baseline_sites_per_s:
            rec["vs_cpu_baseline"] = sites / best / cpu_baseline_sites_per_s
        return rec
    finally:
        srv.terminate()
        try:
            srv.wait(timeout=30)
        except Exception:
            srv.kill()


def hrc_cli_record(tmp, n_samples=32488, sites=142000, seed=7):
    """The four commands the reference publishes its numbers on (README.md:276-281, HRC r1: 32,488 samples, ~142,000 sites of
    chr11:10-20 Mb) on a synthetic cohort of that width AND length, through both binaries: stdout compared; wall time of one
    cold process each (this repo's includes the HIP start and the image build), of the same command line answered by a resident
    `bgt-server -u` (best of 3), and of the reference process -- beside the seconds the reference's README publishes."""
    __import__("bgt_amd").build_host_shell()
    prefix = os.path.join(tmp, "hrc_%d_%d" % (n_samples, sites))
    if not os.path.exists(prefix + ".pbf"):
        subprocess.check_call([MY_BIN, "synth", prefix, str(n_samples), str(sites), str(seed)])
    rec = {"name": "HRC-cli", "sites": sites,
           "workload": "HRC r1 shape at the published length: %d samples (%d haplotypes) x %d sites, the reference's four published "
                       "commands (README.md:276-281)" % (n_samples, 2 * n_samples, sites), "commands": []}
    srv_bin = os.path.join(ROOT, "bgt_amd", "bin", "bgt-server")
    sock = os.path.join(tmp, "hrc.sock")
    srv = subprocess.Popen([srv_bin, "-u", sock, prefix], stderr=subprocess.DEVNULL)
    try:
        t0 = time.perf_counter()
        while not os.path.exists(sock):
            if srv.poll() is not None or time.perf_counter() - t0 > 120:
                raise RuntimeError("bgt-server -u did not come up")
            time.sleep(0.02)
        env_res = dict(os.environ, BGT_SERVER=sock)
        for label, va, published in (("view -G", ["-G"], 11.0), ("view -GC", ["-G", "-C"], 30.0),
                                     ("view -GC -s (2,499 of 32,488)", ["-G", "-C", "-s", "idx%13==0"], 4.0),
                                     ("view -G -s A -s B", ["-G", "-s", "idx<2500", "-s", "idx>=30000"], 8.0)):
            t0 = time.perf_counter()
            mine = subprocess.run([MY_BIN, "view"] + va + [prefix], stdout=subprocess.PIPE, check=True).stdout
            t_cold = time.perf_counter() - t0
            t_res, res_same = None, True
            for _ in range(3):
                t0 = time.perf_counter()
                o = subprocess.run([MY_BIN, "view"] + va + [prefix], stdout=subprocess.PIPE, check=True, env=env_res).stdout
                dt = time.perf_counter() - t0
                t_res = dt if t_res is None or dt < t_res else t_res
                res_same = res_same and hashlib.md5(o).hexdigest() == hashlib.md5(mine).hexdigest()
            c = {"command": label, "this_repo_s": round(t_cold, 3), "resident_s": round(t_res, 4), "published_s": published,
                 "stdout_bytes": len(mine), "resident_stdout_identical": res_same}
            if not res_same:
                rec["parity_error"] = "`bgt view %s` through the resident host differs from the local run" % " ".join(va)
            if os.path.exists(REF_BIN):
                t0 = time.perf_counter()
                ref = subprocess.run([REF_BIN, "view"] + va + [prefix], stdout=subprocess.PIPE, check=True).stdout
                c["reference_s"] = round(time.perf_counter() - t0, 3)
                c["reference_sites_per_s"] = sites / c["reference_s"]
                c["stdout_identical"] = hashlib.md5(ref).hexdigest() == hashlib.md5(mine).hexdigest() and res_same
                if not c["stdout_identical"]:
                    rec["parity_error"] = "`bgt view %s` differs from the reference at the HRC width" % " ".join(va)
            rec["commands"].append(c)
    finally:
        srv.terminate()
        try:
            srv.wait(timeout=30)
        except Exception:
            srv.kill()
    return rec


def c5_record(tmp, n_samples=50000, sites=65536):
    """configs[4] (C5) in its product form on one device: two databases (seeds 5 and 6: the same positions, independent
    alleles), two sample groups across both, `-f 'AC1>0&&AC2==0'` -- this repo's CLI (one device image per database, and
    with BGT_GPUS every database dealt over four shards) and the compiled reference on the same files, stdout compared."""
    __import__("bgt_amd").build_host_shell()
    dbs = []
    for tag, seed in (("c5a", 5), ("c5b", 6)):
        prefix = os.path.join(tmp, "%s_%d_%d" % (tag, n_samples, sites))
        subprocess.check_call([MY_BIN, "synth", prefix, str(n_samples), str(sites), str(seed)])
        dbs.append(prefix)
    cmd = ["view", "-G", "-s", 'pop=="A"', "-s", 'pop=="B"', "-f", "AC1>0&&AC2==0"] + dbs

    def timed(binary, env=None, repeats=1):
        best, sig = None, None
        for _ in range(repeats):
            t0 = time.perf_counter()
            o = subprocess.run([binary] + cmd, stdout=subprocess.PIPE, check=True, env=env).stdout
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
            sig = (hashlib.md5(o).hexdigest(), len(o))
        return best, sig

    t_mine, sig_mine = timed(MY_BIN, repeats=2)
    t_sh, sig_sh = timed(MY_BIN, env=dict(os.environ, BGT_GPUS="0,0,0,0"), repeats=2)
    rec = {"name": "C5-cli", "workload": "C5 shape: two databases x %d samples x %d sites, two groups across both, "
                                         "-G -f'AC1>0&&AC2==0' through `bgt view`" % (n_samples, sites),
           "command": "bgt view -G -s 'pop==\"A\"' -s 'pop==\"B\"' -f 'AC1>0&&AC2==0' a b",
           "wall_s": round(t_mine, 3), "merged_sites_per_s": 2 * sites / t_mine,
           "sharded": {"BGT_GPUS": "0,0,0,0 (four shards per database, all on this box's one device)", "wall_s": round(t_sh, 3),
                       "stdout_identical": sig_sh == sig_mine},
           "stdout_bytes": sig_mine[1]}
    if os.path.exists(REF_BIN):
        t_ref, sig_ref = timed(REF_BIN)
        rec["reference"] = {"wall_s": round(t_ref, 2), "cores": 1, "stdout_identical": sig_ref == sig_mine,
                            "speedup": t_ref / t_mine}
        if sig_ref != sig_mine or sig_sh != sig_mine:
            rec["parity_error"] = "`bgt view` stdout over two databases differs (reference / sharded / single)"
    elif sig_sh != sig_mine:
        rec["parity_error"] = "`bgt view` stdout over two databases differs between the sharded and the single image"
    return rec


def product_sharded_record(torch, bgt_amd, np, devices, n_samples, sites, seed, steps, tmp, view_cmds=True):
    """The PRODUCT's multi-GPU path, which `bgt view` takes under BGT_GPUS and which the torch ranks of this script do not:
    ONE process opens one database file as block-aligned site-range shards, one per listed device (bgth_pbf_open_sharded: a
    partial image, stream and host thread per shard), and bgth_reader_scan_device gathers the shards' counts on the first
    device over RCCL (ncclSend / ncclRecv in one group; shards of one device: copies).  Timed: `steps` gathered whole-cohort
    scans of all `sites` rows, wall clock around enqueue .. root-stream synchronise; per shard the kernel time by HIP events.
    Checked: the gathered counts against a single-device image of the same file and the plane-popcount identity on every site.
    Then the command line: `BGT_GPUS=<devices> bgt view -G -f'AC>0'` against the same command on one device."""
    __import__("bgt_amd").build_host_shell()
    m = 2 * n_samples
    prefix = os.path.join(tmp, "full_%d_%d" % (n_samples, sites))
    if not os.path.exists(prefix + ".pbf"):
        subprocess.check_call([MY_BIN, "synth", prefix, str(n_samples), str(sites), str(seed)])
    rec = {"devices": list(devices), "database": "%d samples x %d sites (one .pbf, %d shards)" % (n_samples, sites, len(devices)),
           "path": "bgth_pbf_open_sharded + bgth_reader_scan_device (RCCL send/recv gather on device %d)" % devices[0]}
    t0 = time.perf_counter()
    pbf = bgt_amd.HipPbf.open_sharded(prefix + ".pbf", list(devices))
    rd = bgt_amd.HipReader(pbf)
    rec["open_s"] = round(time.perf_counter() - t0, 2)
    with torch.cuda.device(devices[0]):
        d = torch.empty((sites, 1, 3), dtype=torch.int32, device="cuda:%d" % devices[0])
        wall = []
        with bgt_amd.forced_kernels(bgt_amd.hip.FORCE_REBUILD_ROWS):          # (one-shot figures, like the ranks')
            for k in range(steps + 1):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                rd.scan_device(0, sites, d.data_ptr())                          # stream = NULL: returns when the gathered counts are there
                wall.append((time.perf_counter() - t0) * 1e3)
        wall = wall[1:]                                                          # (the first one warms up)
        shard_ms = [rd.shard_timing(i)["scan_ms"] for i in range(len(devices))]
        got = d.cpu().numpy()
    rec.update({"steps": steps, "ms_per_step": sum(wall) / len(wall), "best_ms": min(wall), "sites_per_s": sites / (sum(wall) / len(wall) * 1e-3),
                "per_shard_kernel_ms": [round(x, 3) for x in shard_ms],
                "gather_ms": max(0.0, min(wall) - max(shard_ms)),               # what the step costs beyond its slowest shard's kernel
                "kernel_path": rd.path()})
    rd.close()
    pbf.close()
    one = bgt_amd.HipPbf.open(prefix + ".pbf", device=devices[0])
    rd1 = bgt_amd.HipReader(one)
    want = rd1.scan(0, sites)
    rd1.close()
    one.close()
    rle, lens = bgt_amd.synth_rows(m, 0, sites, seed)
    ident = popcount_identity(np, got, plane_ones(np, rle, lens)[:sites], m)
    del rle, lens
    rec["parity_ok"] = bool(np.array_equal(got, want) and ident)
    rec["parity"] = {"gathered_equals_single_image": bool(np.array_equal(got, want)), "popcount_identity_ok": ident, "sites": sites}
    if not rec["parity_ok"]:
        rec["parity_error"] = "product_sharded: gathered counts differ: %s" % json.dumps(rec["parity"])
    if view_cmds:
        cmd = [MY_BIN, "view", "-G", "-f", "AC>0", prefix]
        runs = {}
        # (BGT_GPUS: a list of one needs its comma -- "N" alone means devices 0..N-1)
        for tag, env in (("one_device", {"BGT_GPUS": "%d," % devices[0]}), ("sharded", {"BGT_GPUS": ",".join(str(x) for x in devices)})):
            best, sig = None, None
            for _ in range(2):
                t0 = time.perf_counter()
                o = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
                dt = time.perf_counter() - t0
                if o.returncode != 0:
                    runs[tag] = {"error": o.stderr.decode()[-200:]}
                    break
                best = dt if best is None or dt < best else best
                sig = (hashlib.md5(o.stdout).hexdigest(), len(o.stdout))
            else:
                runs[tag] = {"BGT_GPUS": env["BGT_GPUS"], "wall_s": round(best, 3), "stdout_md5": sig[0], "stdout_bytes": sig[1]}
        same = "error" not in runs.get("one_device", {"error": 1}) and "error" not in runs.get("sharded", {"error": 1}) and \
            runs["one_device"]["stdout_md5"] == runs["sharded"]["stdout_md5"]
        rec["bgt_view"] = {"command": "bgt view -G -f 'AC>0' <prefix>", **runs, "stdout_identical": bool(same)}
        if not same:
            rec["parity_error"] = "product_sharded: `BGT_GPUS=... bgt view` differs from the one-device run: %s" % json.dumps(runs)[:300]
    return rec


def product_child(spec):
    """Child mode of the product_sharded record: a process of its own -- a first run on eight real devices that hangs (RCCL
    initialisation beside torch's, a device that does not answer) is killed by its parent after a deadline and costs the bench
    line one record, not the line.  Prints one JSON object."""
    import numpy as np
    import torch
    import bgt_amd
    devs, n_samples, sites, seed, steps, tmp = spec.split(":")
    try:
        rec = product_sharded_record(torch, bgt_amd, np, [int(x) for x in devs.split(",")], int(n_samples), int(sites), int(seed), int(steps), tmp)
    except Exception as e:
        import traceback
        rec = {"error": repr(e)[:300], "traceback_tail": [t[:160] for t in traceback.format_exc().splitlines()[-4:]], "devices": devs}
    sys.stdout.flush()
    print("PRODUCT_SHARDED_JSON " + json.dumps(rec, default=str), flush=True)


def product_sharded_in_a_child(devs, n_samples, sites, seed, steps, tmp, deadline_s=420):
    spec = ":".join([",".join(str(x) for x in devs), str(n_samples), str(sites), str(seed), str(steps), tmp])
    try:
        pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--product-child", spec], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                            timeout=deadline_s, env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")})
    except subprocess.TimeoutExpired:
        return {"error": "no answer within %d s: the child was killed (the ranks' headline above is unaffected)" % deadline_s, "devices": list(devs)}
    for ln in reversed(pr.stdout.decode(errors="replace").splitlines()):
        if ln.startswith("PRODUCT_SHARDED_JSON "):
            return json.loads(ln[len("PRODUCT_SHARDED_JSON "):])
    return {"error": "the child left with status %d and no record: %s" % (pr.returncode, pr.stderr.decode(errors="replace")[-300:]), "devices": list(devs)}


def server_record(prefix, n_sites):
    """The resident query server (bgt_amd/bin/bgt-server, the reference's bgt-server.go restated in C) on the database
    cli_end_to_end wrote: images in HBM once, then per-query latency over HTTP next to one reference `bgt view` process per
    query; the record lines of every answer are compared with the reference's."""
    import socket
    import urllib.request
    srv_bin = os.path.join(ROOT, "bgt_amd", "bin", "bgt-server")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    t0 = time.perf_counter()
    srv = subprocess.Popen([srv_bin, "-p", str(port), "-m", "4000000000", prefix], stderr=subprocess.DEVNULL)
    try:
        while True:
            try:
                socket.create_connection(("127.0.0.1", port), timeout=1).close()
                break
            except OSError:
                if srv.poll() is not None or time.perf_counter() - t0 > 120:
                    raise RuntimeError("bgt-server did not come up")
                time.sleep(0.02)
        ready = time.perf_counter() - t0
        mid = 1000 + 10 * (n_sites // 2)
        out = {"name": "server", "program": "bgt-server -p PORT -m 4000000000 <prefix> (images resident in HBM)",
               "ready_after_s": round(ready, 2), "queries": []}
        for label, q, va in (
                ("100 sites of a region, AC/AN", "C&r=11:%d-%d" % (mid, mid + 999), ["-G", "-C", "-r", "11:%d-%d" % (mid, mid + 999)]),
                ("10,000 sites of a region, -f'AC>0'", "f=AC%%3E0&r=11:%d-%d" % (mid, mid + 99999), ["-G", "-f", "AC>0", "-r", "11:%d-%d" % (mid, mid + 99999)]),
                ("genotypes of 20 samples over 1,000 sites", "g&s=idx%%3C20&r=11:%d-%d" % (mid, mid + 9999), ["-C", "-s", "idx<20", "-r", "11:%d-%d" % (mid, mid + 9999)])):     # (the server sets -C whenever s is given)
            best, body = None, b""
            for _ in range(5):
                t = time.perf_counter()
                body = urllib.request.urlopen("http://127.0.0.1:%d/?%s" % (port, q), timeout=600).read()
                dt = time.perf_counter() - t
                best = dt if best is None or dt < best else best
            rec = {"query": label, "server_ms": round(best * 1e3, 2), "lines": body.count(b"\n")}
            if os.path.exists(REF_BIN):
                t = time.perf_counter()
                ref = subprocess.run([REF_BIN, "view"] + va + [prefix], stdout=subprocess.PIPE, check=True).stdout
                rec["reference_view_process_ms"] = round((time.perf_counter() - t) * 1e3, 2)
                recs = lambda b: [ln for ln in b.split(b"\n") if ln and not ln.startswith(b"#")]
                rec["records_identical_to_reference"] = recs(body) == recs(ref)
                if not rec["records_identical_to_reference"]:
                    out["parity_error"] = "bgt-server records differ from reference `bgt view` (%s)" % label
            out["queries"].append(rec)
        return out
    finally:
        srv.terminate()
        srv.wait(timeout=30)


class Pipeline:
    """scan -> device filter -> (all_gather) -> pinned host copy, double buffered: the copy of step i overlaps step i+1."""

    def __init__(self, torch, bgt_amd, rd, row0, row1, dev, local, world, rank, dist):
        self.torch, self.rd, self.row0, self.row1, self.world, self.rank, self.dist = torch, rd, row0, row1, world, rank, dist
        n = self.n = row1 - row0
        self.flt = bgt_amd.HipFilter("AC>0", n_groups=1, device=local)            # -f'AC>0', evaluated on the device
        self.main = torch.cuda.current_stream()
        self.side = torch.cuda.Stream(device=dev)
        self.counts = [torch.empty((n, 1, 3), dtype=torch.int32, device=dev) for _ in range(2)]
        self.flags = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(2)]
        self.n_pass_d = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(2)]
        if world > 1:
            self.g_counts = [torch.empty((world * n, 1, 3), dtype=torch.int32, device=dev) for _ in range(2)]
            self.g_flags = [torch.empty(world * n, dtype=torch.uint8, device=dev) for _ in range(2)]
        else:
            self.g_counts, self.g_flags = self.counts, self.flags
        if rank == 0:
            self.host = [torch.empty((world * n, 1, 3), dtype=torch.int32).pin_memory() for _ in range(2)]
            self.host_flags = [torch.empty(world * n, dtype=torch.uint8).pin_memory() for _ in range(2)]
            self.host_n_pass = [torch.zeros(1, dtype=torch.int64).pin_memory() for _ in range(2)]
        self.ready = [torch.cuda.Event() for _ in range(2)]
        self.gather_ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        self.copied = [torch.cuda.Event() for _ in range(2)]
        for e in self.copied:
            e.record(self.main)
        self.done = 0

    def step(self):
        torch = self.torch
        b = self.done & 1
        self.done += 1
        self.main.wait_event(self.copied[b])                                     # buffer b has left the device
        self.rd.scan_device(self.row0, self.row1, self.counts[b].data_ptr(), stream=self.main.cuda_stream)
        self.n_pass_d[b].zero_()
        self.flt.apply_device(self.counts[b].data_ptr(), self.n, 3, self.flags[b].data_ptr(), self.n_pass_d[b].data_ptr(),
                              self.main.cuda_stream)
        if self.world > 1:                                                       # per-shard AN/AC + flags over xGMI
            self.gather_ev[0].record(self.main)
            if self.dist.get_backend() == "gloo":                        # dry run of the multi-rank path without RCCL (tests)
                self.main.synchronize()
                gc, gf, npass = self.g_counts[b].cpu(), self.g_flags[b].cpu(), self.n_pass_d[b].cpu()
                self.dist.all_gather_into_tensor(gc, self.counts[b].cpu())
                self.dist.all_gather_into_tensor(gf, self.flags[b].cpu())
                self.dist.all_reduce(npass)
                self.g_counts[b].copy_(gc); self.g_flags[b].copy_(gf); self.n_pass_d[b].copy_(npass)
            else:
                self.dist.all_gather_into_tensor(self.g_counts[b], self.counts[b])
                self.dist.all_gather_into_tensor(self.g_flags[b], self.flags[b])
                self.dist.all_reduce(self.n_pass_d[b])
            self.gather_ev[1].record(self.main)
        self.ready[b].record(self.main)
        if self.rank == 0:
            with torch.cuda.stream(self.side):
                self.side.wait_event(self.ready[b])
                self.host[b].copy_(self.g_counts[b], non_blocking=True)
                self.host_flags[b].copy_(self.g_flags[b], non_blocking=True)
                self.host_n_pass[b].copy_(self.n_pass_d[b], non_blocking=True)
                self.copied[b].record(self.side)
        return b                                                                 # (nothing here waits: steps are enqueued back to back)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def run(self, steps, warmup):
        for _ in range(warmup):
            self.step()
        self.barrier()
        t0 = time.perf_counter()
        last = 0
        for _ in range(steps):
            last = self.step()
        self.barrier()                                                           # includes the side stream: all results on the host
        dt = time.perf_counter() - t0
        k_ms = self.rd.timing()["scan_ms"]                                       # HIP events around the scan kernel of the last step
        self.gather_ms = self.gather_ev[0].elapsed_time(self.gather_ev[1]) if self.world > 1 else 0.0   # (last step)
        return dt, k_ms, last


def secondary_record(torch, bgt_amd, np, peak, name, what, n_samples, sites, seed, every, steps, warmup, dev, local, tmp,
                     cpu_sites, counters_workload, inrun=True):
    m = 2 * n_samples
    t0 = time.time()
    rle, lens = bgt_amd.synth_rows(m, 0, sites, seed)
    t_gen = time.time() - t0
    t0 = time.time()
    pbf = bgt_amd.HipPbf.from_rle(m, 13, rle, lens, device=local)
    t_load = time.time() - t0
    rle_bytes_per_site = rle.size / sites
    ones = plane_ones(np, rle, lens) if every <= 1 else None
    del rle
    rd = bgt_amd.HipReader(pbf)
    view_args = ["-G", "-f", "AC>0"]
    if every > 1:
        sel = np.arange(0, n_samples, every)
        rd.select(np.stack([2 * sel, 2 * sel + 1], 1).reshape(-1))
        view_args = ["-G", "-f", "AC>0", "-s", "idx%%%d==0" % every]
    T = rd.width
    pipe = Pipeline(torch, bgt_amd, rd, 0, sites, dev, local, 1, 0, None)
    # ONE-SHOT figures first: every step builds its rows again (the directory path would otherwise walk the arena the
    # previous step left; FORCE_REBUILD_ROWS forbids that) -- then the same steps with the arena kept, labelled so
    kept = None
    with bgt_amd.forced_kernels(bgt_amd.hip.FORCE_REBUILD_ROWS):
        dt, k_ms, last = pipe.run(steps, warmup)
    geo, path = rd.geometry(), rd.path()
    if path["directory_path"]:
        dt2, k2, _ = pipe.run(steps, 1)
        kept = {"sites_per_s": sites / (dt2 / steps), "ms_per_step": dt2 / steps * 1e3, "kernel_ms": k2,
                "note": "second and later passes of a reader over the same rows: the directory arena (%.1f GB) still holds "
                        "them, only the walk-only kernel runs" % (sites * 2.0 * (((m + 31) // 32 + 2) & ~1) * 8 / 1e9)}
    rec = {"workload": what, "haplotypes": m, "tracked_columns": T, "sites": sites,
           "sites_per_s": sites / (dt / steps), "ms_per_step": dt / steps * 1e3, "kernel_ms": k_ms, "steps": steps,
           "sites_passing_filter": int(pipe.host_n_pass[last].item()), "launch": geo, "kernel_path": path,
           "rle_bytes_per_site": round(rle_bytes_per_site, 1), "hbm_resident_bytes": pbf.hbm_bytes,
           "setup": {"generate_s": round(t_gen, 1), "upload_and_checkpoints_s": round(t_load, 1)},
           "roofline": make_roofline(peak, T, sites, k_ms, rle_bytes_per_site, geo, counters_workload, sites, path, hbm_floor(pbf, T, sites))}
    if kept:
        kept["roofline_frac"] = 2.0 * T * sites / (kept["kernel_ms"] * 1e-3) / 1e9 / guide_peak(peak)
        rec["arena_kept"] = kept
    # ---- the timed step's output: (1) whole cohort: the plane-popcount identity on EVERY site; (2) a CPU-oracle window
    # across a mid-file block boundary (the oracle starts from the order the image holds 2048 rows before the boundary and
    # carries its own order across it)
    host = pipe.host[last].numpy()
    par = {}
    if ones is not None:
        par["popcount_identity_ok"] = popcount_identity(np, host[:sites], ones[:sites], m)
        par["sites_checked_popcount_identity"] = sites
    mid = (sites // 2) // 8192 * 8192
    if mid >= 8192:
        cols = np.stack([2 * sel, 2 * sel + 1], 1).reshape(-1) if every > 1 else None
        oc, t_or = oracle_window(bgt_amd, np, pbf, m, 13, seed, mid - 2048, mid - 2048, 2048 + 512, tmp, local, cols)
        par["oracle_window"] = {"rows": [mid - 2048, mid + 512], "block_boundary_at_row": mid,
                                "matches": bool(np.array_equal(oc, host[mid - 2048: mid + 512])), "oracle_s": round(t_or, 2)}
    par["parity_ok"] = bool(par.get("popcount_identity_ok", True) and par.get("oracle_window", {"matches": True})["matches"])
    rec["parity"], rec["parity_ok"] = par, par["parity_ok"]
    if not par["parity_ok"]:
        rec["parity_error"] = "timed output fails the on-box checks: %s" % json.dumps(par)
    if os.path.exists(REF_BIN) and cpu_sites > 0:
        try:
            base, same = reference_cli_baseline(n_samples, cpu_sites, seed, view_args, tmp, what)
            # the timed GPU output against the reference's stdout on those sites: passing sites = data lines
            rec["cpu_baseline"] = base
            rec["speedup_vs_reference_1core"] = rec["sites_per_s"] / base["value"]
            if not same:
                rec["parity_error"] = "`bgt view` stdout differs from the reference binary"
        except Exception as e:
            rec["cpu_baseline"] = {"error": repr(e)[:200]}
    del pipe
    rd.close()
    pbf.close()
    if path["directory_path"] and every <= 1 and inrun:
        # HBM bytes of the one-shot scan measured in this run: walk-only kernel + producer (the arena rebuilt by every scan);
        # after this process has given its image and arena back, so that the child's scan is one pass like the timed one
        kn = "walk_kernel<%d, %d" % (geo["threads"], geo["cols_per_thread"])
        ic = inrun_counters(n_samples, sites, seed, tmp, kn, "dirbuild_kernel", {"BENCH_REBUILD_ROWS": "1"})
        rec["roofline"]["traffic_in_run"] = ic
        if "hbm_bytes_per_launch" in ic:
            rec["roofline"]["traffic"] = ic["hbm_bytes_per_launch"]
            rec["roofline"]["traffic_source"] = ic["source"]
            rec["roofline"]["hbm_frac_measured"] = ic["hbm_bytes_per_launch"] / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            if ic.get("producer", {}).get("write_bytes") and path.get("producer_ms"):
                rec["roofline"]["producer_hbm_write_gbs"] = ic["producer"]["write_bytes"] / (path["producer_ms"] * 1e-3) / 1e9
    return rec


def sharded_run(args, ctx, workload, steps, warmup, sites_arg):
    """One timed run of the hot path over this rank's shard: synthetic rows -> image in HBM -> (N > 1: chained onto the shards
    before it) -> `steps` x (scan + device filter + gather + copy to rank 0's pinned memory) -> rank 0's on-box parity checks
    of the gathered result.  Returns a namespace with the record (rank 0: .out) and what the N = 1 follow-ups use."""
    torch, bgt_amd, np, dist, rank, world, local, dev, block_shards = (ctx.torch, ctx.bgt_amd, ctx.np, ctx.dist, ctx.rank, ctx.world,
                                                                        ctx.local, ctx.dev, ctx.block_shards)
    n_samples = SAMPLES[workload]
    m = 2 * n_samples
    shift = 13
    seed = args.seed or SEEDS[workload]
    strong = workload == "c4"
    if strong:                                                # configs[3]: one database, block-aligned shards
        total = sites_arg or 10000000
        shards = block_shards(total, shift, world)
        row_lo, row_hi = shards[rank]
        sites = shards[0][1] - shards[0][0]                   # every rank's buffers are sized for the longest shard
    else:
        sites = sites_arg or 1000000
        total = world * sites
        row_lo, row_hi = rank * sites, (rank + 1) * sites

    # ---- synthetic shard: file rows [row_lo, row_hi) of cohort (seed, m), drawn on the host cores
    t0 = time.time()
    my_rows = row_hi - row_lo
    if strong and my_rows < sites:                            # the last shard is shorter: pad with the cohort's next rows so
        my_rows = sites                                       # that all_gather sees equal sizes (they are cut off below)
    rle, lens = bgt_amd.synth_rows(m, row_lo, my_rows, seed)
    t_gen = time.time() - t0
    t0 = time.time()
    pbf = bgt_amd.HipPbf.from_rle(m, shift, rle, lens, device=local)   # upload + checkpoints on the GPU
    t_load = time.time() - t0
    rle_bytes_per_site = rle.size / my_rows
    t_chain = 0.0
    if world > 1:
        # ONE database: shard r starts from the order the rows before it leave behind = final(r-1) o ... o final(0),
        # every final(i) being what rank i's shard does to the identity order (ranks by column; one gather per plane)
        t0 = time.time()
        mine = torch.from_numpy(pbf.final_ranks())                # (the last, padded shard's is used by nobody)
        finals = [torch.empty_like(mine) for _ in range(world)]
        if args.backend == "nccl":
            g = [f.to(dev) for f in finals]
            dist.all_gather(g, mine.to(dev))
            finals = [f.cpu() for f in g]
        else:
            dist.all_gather(finals, mine)
        start = np.stack([np.arange(m, dtype=np.int32)] * 2)
        for i in range(rank):
            f = finals[i].numpy()
            start = np.stack([f[0][start[0]], f[1][start[1]]])
        if rank:
            pbf.rebase(start)
        t_chain = time.time() - t0
    rd = bgt_amd.HipReader(pbf)
    if args.every > 1:                                          # sample subset (-s): fewer tracked columns, same rows
        sel = np.arange(0, n_samples, args.every)
        rd.select(np.stack([2 * sel, 2 * sel + 1], 1).reshape(-1))
    rd.tune(args.threads, args.cpt, args.batch)
    T = rd.width

    if rank == 0 and ctx.peak is None:
        ctx.peak = lookup_peak(bgt_amd, local)
    peak = ctx.peak
    pipe = Pipeline(torch, bgt_amd, rd, 0, sites, dev, local, world, rank, dist)
    # Every timed step does ALL the work of a pass: on the directory path (wide cohorts: C4) the rows are built again by every
    # step instead of being walked from the arena the step before left behind (FORCE_REBUILD_ROWS; no effect on the other
    # kernels).  The arena-kept rate is reported by the C4-shard secondary record, labelled.
    one_shot_forced = True
    with bgt_amd.forced_kernels(bgt_amd.hip.FORCE_REBUILD_ROWS):
        dt, k_ms, last = pipe.run(steps, warmup)
    n_pass = int(pipe.host_n_pass[last].item()) if rank == 0 else 0
    if rank == 0:
        host = pipe.host[last]
        host_flags = pipe.host_flags[last]
        if not strong:
            assert n_pass == int(host_flags.numpy().sum())
        else:                                                   # the last shard is padded to the longest: count its real rows only
            hf = host_flags.numpy()
            n_pass = int(sum(hf[r * sites: r * sites + (shards[r][1] - shards[r][0])].sum() for r in range(world)))
    rank_kernel_ms, parity = [k_ms], None
    if world > 1:
        cdev = dev if args.backend == "nccl" else "cpu"
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        ks = [torch.zeros(1, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(ks, torch.tensor([k_ms], dtype=torch.float64, device=cdev))
        rank_kernel_ms = [float(k.item()) for k in ks]
        dv = [torch.zeros(1, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(dv, torch.tensor([float(torch.cuda.current_device())], dtype=torch.float64, device=cdev))
        rank_devices = [int(x.item()) for x in dv]
        # ---- parity of the gathered result, after the timed region.  (1) every rank counts the ones of its strings; rank 0
        # holds the gathered counts and checks the plane-popcount identity on every site of every shard
        real = row_hi - row_lo
        ones = torch.zeros((sites, 2), dtype=torch.int64)
        if args.every <= 1:
            ones[:real] = torch.from_numpy(plane_ones(np, rle, lens)[:real])
        all_ones = [torch.empty_like(ones, device=cdev) for _ in range(world)]
        dist.all_gather(all_ones, ones.to(cdev))
        if rank == 0:
            parity = {"sites_checked_popcount_identity": 0, "popcount_identity_ok": None}
            if args.every <= 1:
                ok = True
                for r in range(world):
                    n_r = (shards[r][1] - shards[r][0]) if strong else sites
                    ok = ok and popcount_identity(np, host.numpy()[r * sites: r * sites + n_r], all_ones[r].cpu().numpy()[:n_r], m)
                    parity["sites_checked_popcount_identity"] += n_r
                parity["popcount_identity_ok"] = ok
            # (2) a CPU-oracle window across the boundary between shard 0 and shard 1: the last rows of rank 0's shard and the
            # first rows of rank 1's, decoded sequentially from the order rank 0's image holds before the window
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            with tempfile.TemporaryDirectory() as wtmp:
                edge = row_hi - row_lo                                   # rank 0: rows [0, edge) of its image
                back = edge - max(0, (edge - (2048 if m > 20000 else 8192)) // 2048 * 2048)   # (starts on a checkpoint row)
                ahead = min(sites, 512 if m > 20000 else 2048)
                cols = np.stack([2 * sel, 2 * sel + 1], 1).reshape(-1) if args.every > 1 else None
                oc, t_or = oracle_window(bgt_amd, np, pbf, m, shift, seed, row_hi - back, edge - back, back + ahead, wtmp, local, cols)
                got = np.concatenate([host.numpy()[edge - back: edge], host.numpy()[sites: sites + ahead]])
                parity["oracle_window"] = {"rows": [int(row_hi - back), int(row_hi + ahead)], "shard_boundary_at_row": int(row_hi),
                                           "matches": bool(np.array_equal(oc, got)), "oracle_s": round(t_or, 2)}
            parity["parity_ok"] = bool(parity["oracle_window"]["matches"] and parity["popcount_identity_ok"] is not False)

    ms_per_step = dt / steps * 1e3
    value = total / (dt / steps)
    geo = rd.geometry()

    out = None
    if rank == 0:
        if workload == "c2":
            wl = "C2: synthetic %d samples x %d sites per GPU, whole cohort, -G -f'AC>0'" % (n_samples, sites)
        elif strong:
            wl = "C4: synthetic %d samples x %d sites, whole cohort, -G -f'AC>0', %d file blocks sharded over %d GPU(s)" % (
                n_samples, total, (total + 8191) // 8192, world)
        else:
            wl = "%s: %d samples x %d sites per GPU%s" % (workload, n_samples, sites,
                                                          ", every %d-th sample selected" % args.every if args.every > 1 else "")
        out = {
            "metric": "sites/sec `bgt view -G -f'AC>0'` whole-cohort scan",
            "value": value, "unit": "sites/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": ("strong" if strong else "weak") if world > 1 else None,      # (one GPU: nothing scales)
            "vs_baseline": None,
            "vs_baseline_note": "BASELINE.json publishes no number for this metric; the like-for-like ratio against the reference process "
                                "on this box is resident_end_to_end.vs_cpu_baseline",
            "value_scope": "kernel pipeline: scan + device filter + copy of counts and flags to the host, inputs resident in HBM -- no "
                           "VCF text; the command line end to end is resident_end_to_end (image resident) and cli_end_to_end (cold process)",
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": wl,
                       "haplotypes": m, "tracked_columns": T, "sites_per_gpu": sites, "sites_total": total,
                       "sharding": ("block-aligned site ranges x%d (bgt_amd.shard.block_shards) + all_gather(counts)" % world if strong else
                                    "site-range x%d + all_gather(counts)" % world) if world > 1 else "single GPU",
                       "rle_bytes_per_site": round(rle_bytes_per_site, 1), "sites_passing_filter": n_pass,
                       "filter": "AC>0 evaluated on the device (bgth_filter_apply_device); counts + flags copied "
                                 "to pinned host memory, the copy of step i overlapping the scan of step i+1",
                       "launch": geo},
            "roofline": make_roofline(peak, T, sites, k_ms, rle_bytes_per_site, geo, workload, sites, rd.path(), hbm_floor(pbf, T, sites)),
            "setup": {"generate_s": round(t_gen, 2), "upload_and_checkpoints_s": round(t_load, 2),
                      "hbm_resident_bytes": pbf.hbm_bytes},
        }
        if world > 1:
            out["parity_ok"] = parity["parity_ok"]
            out["parity"] = parity
            out["per_rank_kernel_ms"] = rank_kernel_ms
            out["ranks"] = {"world_size": world, "backend": dist.get_backend(), "device_of_rank": rank_devices,
                            "devices_visible": torch.cuda.device_count()}
            out["gather_ms"] = pipe.gather_ms
            out["setup"]["chain_shards_s"] = round(t_chain, 2)
            out["config"]["one_database"] = ("every shard built from the identity order, then re-based onto the composition of "
                                             "the earlier shards' final ranks (bgth_pbf_final_ranks / bgth_pbf_rebase)")
            out["config"]["kernel_path"] = rd.path()
            out["config"]["rows_rebuilt_by_every_step"] = bool(one_shot_forced)       # (no directory arena carried from step to step)
            if not parity["parity_ok"]:
                out["parity_error"] = "gathered counts fail the on-box checks: %s" % json.dumps(parity)
            if strong:
                out["config"]["n1_point"] = "python bench.py --workload c4 (the same database on one GPU)"

    R = types.SimpleNamespace(out=out, rle=rle, lens=lens, pbf=pbf, rd=rd, pipe=pipe, sites=sites, total=total, m=m, shift=shift, seed=seed,
                              row_lo=row_lo, n_samples=n_samples, geo=geo, k_ms=k_ms, T=T,
                              host=pipe.host[last] if rank == 0 else None, host_flags=pipe.host_flags[last] if rank == 0 else None)

    def release():
        R.pipe = R.host = R.host_flags = R.rle = R.lens = None
        R.rd.close()
        R.pbf.close()
        torch.cuda.empty_cache()
    R.release = release
    return R


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None, choices=sorted(SAMPLES),
                    help="default: c2 on one GPU, c4 (BASELINE configs[3], strong scaling over one database) on several")
    ap.add_argument("--sites", type=int, default=0, help="sites per GPU (default 1,000,000; c4: all 10,000,000 split over the GPUs)")
    ap.add_argument("--every", type=int, default=0, help="select every N-th sample only (C3: --workload c3 --every 20)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=1000000,
                    help="sites for the CPU baseline and the from-row-0 parity check (0 = skip; default: C2's full length -- the compiled "
                         "reference and the CPU oracle each decode the whole database from the identity order of row 0, ~23 s each)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the 100,000-sample secondary records (N = 1)")
    ap.add_argument("--secondary-steps", type=int, default=3)
    ap.add_argument("--secondary-sites", type=int, default=0, help="N > 1: total sites of the C4-sharded secondary record (default 10,000,000)")
    ap.add_argument("--no-product-sharded", action="store_true",
                    help="skip the product's own multi-GPU path (one process, bgth_pbf_open_sharded over --gpus devices, RCCL gather)")
    ap.add_argument("--product-devices", default=None,
                    help="device list of the product_sharded record (default: 0..gpus-1; e.g. 0,0 = two shards on one device)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo = dry run through host copies)")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--cpt", type=int, default=0)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--no-counters", action="store_true", help="skip the in-run rocprofv3 --pmc passes (HBM traffic of the scan kernel)")
    ap.add_argument("--detail", default=None, help="where the full record goes (default: bench_detail.json beside bench.py); stdout carries "
                                                   "one compact line")
    ap.add_argument("--counters-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--product-child", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.counters_child:
        counters_child(args.counters_child)
        return
    if args.product_child:
        product_child(args.product_child)
        return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus != 1 and world == 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run" % args.gpus)

    import numpy as np
    import torch
    import bgt_amd
    from bgt_amd.shard import block_shards

    if os.environ.get("BENCH_ALL_RANKS_ON_DEVICE0"):         # dry run of N > 1 on a one-GPU box (with --backend gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            # ("nccl" is RCCL on ROCm.)  A rank that fails alone leaves the others in a collective: five minutes, not the default ten+
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=5))
        else:
            dist.init_process_group(args.backend, timeout=datetime.timedelta(minutes=5))

    ctx = types.SimpleNamespace(torch=torch, bgt_amd=bgt_amd, np=np, dist=dist, rank=rank, world=world, local=local, dev=dev,
                                block_shards=block_shards, peak=None)
    # One GPU: C2, the configuration the metric is quoted on.  Several GPUs: the SAME per-GPU workload, weak scaling -- rank r
    # holds sites [r, r+1) x 1,000,000 of ONE database of N x 1,000,000 sites (a per-N value the N = 1 value compares with) --
    # and, as a secondary record of the same run, BASELINE configs[3]: the 10,000,000-site x 100,000-sample database
    # block-sharded over the N GPUs (strong scaling; `--workload c4` makes it the headline).
    headline_wl = args.workload or "c2"
    R = sharded_run(args, ctx, headline_wl, args.steps, args.warmup, args.sites)
    out = R.out
    (rle, lens, pbf, rd, pipe, sites, total, m, shift, seed, row_lo, n_samples, geo, k_ms, peak, T) = (
        R.rle, R.lens, R.pbf, R.rd, R.pipe, R.sites, R.total, R.m, R.shift, R.seed, R.row_lo, R.n_samples, R.geo, R.k_ms, ctx.peak, R.T)
    host, host_flags = R.host, R.host_flags
    args.workload = headline_wl
    if world > 1 and headline_wl == "c2" and not args.no_secondary and not args.every:
        del pipe, rd, pbf, rle, lens, host, host_flags
        R.release()
        try:
            R4 = sharded_run(args, ctx, "c4", args.secondary_steps, 1, args.secondary_sites)
            if rank == 0:
                rec = R4.out
                rec["name"] = "C4-sharded"
                for k in ("metric", "unit", "higher_is_better", "vs_baseline", "vs_baseline_note", "value_scope", "dtype", "data"):
                    rec.pop(k, None)
                out["secondary"] = [rec]
                if rec.get("parity_error"):
                    out["parity_error"] = "C4-sharded: " + rec["parity_error"]
            R4.release()
        except Exception as e:                                    # the headline stands; the record says what happened
            if rank == 0:
                out["secondary"] = [{"name": "C4-sharded", "error": repr(e)[:400]}]
        rle = lens = pbf = rd = pipe = host = host_flags = None

    # ---- CPU baseline + on-box parity check (rank 0, N=1 only) on a bounded sample of the same cohort:
    # the first `cpu_sample` sites.  The COMPILED REFERENCE (oracle/_ref/bgt, built from /root/reference in the build
    # container and shipped with the repo) runs the metric's own command line on a database this repo writes; its
    # stdout is also compared with this repo's `bgt view`.  Always: the CPU oracle (port) on the same rows, compared
    # with the counts the GPU delivered in the last timed step.
    with tempfile.TemporaryDirectory() as tmp:
        if rank == 0 and world == 1 and args.cpu_sample > 0 and args.every <= 1:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import orc                                            # CPU oracle: checker / baseline only
            ns = min(sites, args.cpu_sample if m <= 20000 else 4096)
            nstr = 2 * ns
            nbytes = int(lens[:nstr].sum(dtype=np.int64))
            sample = bgt_amd.HipPbf.from_rle(m, shift, rle[:nbytes], lens[:nstr], device=local)
            path = os.path.join(tmp, "sample.pbf")
            sample.save(path)
            data = open(path, "rb").read()
            sample.close()
            p = orc.Pbf(data)
            t0 = time.perf_counter()
            oc = p.scan(0, ns)
            n_pass_cpu = int((oc[:, 1] > 0).sum())
            t_cpu = time.perf_counter() - t0
            same = bool(np.array_equal(oc.reshape(ns, 1, 3), host.numpy()[:ns])) and \
                bool(np.array_equal(oc[:, 1] > 0, host_flags.numpy()[:ns] != 0))
            port = {"value": ns / t_cpu, "unit": "sites/s", "cores": 1, "kind": "port",
                    "sample": "first %d sites of the same cohort (oracle/liborc.so: decode both planes + AC/AN + "
                              "AC>0, one thread, %.1f s)" % (ns, t_cpu),
                    "gpu_matches_cpu_on_sample": same, "sites_passing_filter": n_pass_cpu}
            out["cpu_baseline"] = port
            if not same:
                out["parity_error"] = "GPU counts differ from the CPU oracle on the sample"
            # every site of the timed output against the strings' ones, and an oracle window across a mid-file block boundary
            par = {"popcount_identity_ok": popcount_identity(np, host.numpy()[:sites], plane_ones(np, rle, lens)[:sites], m),
                   "sites_checked_popcount_identity": sites,
                   # the timed launch's own counts against ONE sequential oracle pass from the identity order of row 0
                   "from_row_zero": {"sites": ns, "of_sites": sites, "oracle_counts_match": same}}
            mid = (sites // 2) // (1 << shift) * (1 << shift)
            if mid >= (1 << shift):
                back, ahead = (1 << shift) if m <= 20000 else 2048, 1024 if m <= 20000 else 512
                oc, t_or = oracle_window(bgt_amd, np, pbf, m, shift, seed, row_lo + mid - back, mid - back, back + ahead, tmp, local)
                par["oracle_window"] = {"rows": [mid - back, mid + ahead], "block_boundary_at_row": mid,
                                        "matches": bool(np.array_equal(oc, host.numpy()[mid - back: mid + ahead])), "oracle_s": round(t_or, 2)}
            par["parity_ok"] = bool(same and par["popcount_identity_ok"] and par.get("oracle_window", {"matches": True})["matches"])
            out["parity"], out["parity_ok"] = par, par["parity_ok"]
            if not par["parity_ok"] and "parity_error" not in out:
                out["parity_error"] = "timed output fails the on-box checks: %s" % json.dumps(par)
            if os.path.exists(REF_BIN):
                try:
                    base, cli_same = reference_cli_baseline(n_samples, ns, seed, ["-G", "-f", "AC>0"], tmp,
                                                            "the same cohort", all_cores=True)
                    base["gpu_matches_cpu_on_sample"] = same
                    base["port"] = port
                    par["from_row_zero"]["reference_stdout_identical"] = cli_same      # the reference binary over the same ns sites
                    out["cpu_baseline"] = base
                    if not cli_same:
                        out["parity_error"] = "`bgt view` stdout differs from the reference binary"
                except Exception as e:                        # keep the port numbers, say why
                    out["cpu_baseline"]["reference_leg_error"] = repr(e)[:200]
        if rank == 0 and world == 1 and not args.no_counters and not args.every and args.cpu_sample > 0:
            # HBM bytes of the timed kernel, measured in this run (not replayed from profiles/)
            kn = ("walk_kernel<%d, %d" if rd.path()["directory_path"] else "scan_kernel<%d, %d") % (geo["threads"], geo["cols_per_thread"])
            ic = inrun_counters(n_samples, sites, seed, tmp, kn)
            out["roofline"]["traffic_in_run"] = ic
            if "hbm_bytes_per_launch" in ic:
                out["roofline"]["traffic"] = ic["hbm_bytes_per_launch"]
                out["roofline"]["traffic_source"] = ic["source"]
                out["roofline"]["hbm_frac_measured"] = ic["hbm_bytes_per_launch"] / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        if rank == 0 and world == 1 and args.workload == "c2" and args.cpu_sample > 0 and not args.every:
            try:
                out["cli_end_to_end"] = cli_end_to_end(n_samples, sites, seed, tmp)
            except Exception as e:
                out["cli_end_to_end"] = {"error": repr(e)[:200]}
            try:
                out["resident_end_to_end"] = resident_end_to_end(os.path.join(tmp, "full_%d_%d" % (n_samples, sites)), sites,
                                                                 out.get("cpu_baseline", {}).get("value"))
            except Exception as e:
                out["resident_end_to_end"] = {"error": repr(e)[:200]}
            try:
                out["server"] = server_record(os.path.join(tmp, "full_%d_%d" % (n_samples, sites)), sites)
                if out["server"].get("parity_error"):
                    out["parity_error"] = out["server"]["parity_error"]
            except Exception as e:
                out["server"] = {"error": repr(e)[:200]}
        # ---- secondary records at the north-star width (100,000 samples), N = 1 only
        if rank == 0 and world == 1 and args.workload == "c2" and not args.no_secondary and not args.every:
            del pipe
            rd.close()
            pbf.close()
            del rle, lens
            out["secondary"] = []
            if os.path.join(ROOT, "tests") not in sys.path:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
            for name, what, s_samples, s_sites, s_seed, s_every, cw in (
                    ("HRC-GC", "HRC r1 shape (the reference's published numbers, README.md:276-281): synthetic 32488 samples (64,976 haplotypes) x "
                               "142000 sites, whole cohort, counts + -f'AC>0' (`view -GC`)", 32488, 142000, 7, 0, "hrc"),
                    ("HRC-GC-subset", "HRC r1 shape: 32488 samples x 142000 sites, every 13th sample (2,499 samples: `view -GC -s`)",
                     32488, 142000, 7, 13, "hrcsub"),
                    ("C3", "C3: synthetic 100000 samples x 1000000 sites, every 20th sample (5,000 samples, 10,000 tracked columns), -G -f'AC>0'",
                     100000, 1000000, 3, 20, "c3"),
                    ("C4-shard", "C4 shard: synthetic 100000 samples x 1253376 sites (153 file blocks = one GPU's share of the 10,000,000-site "
                                 "configuration), whole cohort, -G -f'AC>0'", 100000, 153 * 8192, 4, 0, "c4shard")):
                try:
                    rec = secondary_record(torch, bgt_amd, np, peak, name, what, s_samples, s_sites, s_seed, s_every,
                                           # (short scans: three steps are at the mercy of one host hiccup)
                                           args.secondary_steps if s_sites > 500000 else max(args.secondary_steps, 20), 1, dev, local, tmp,
                                           8192 + 2048 if s_samples > 50000 else 16384, cw,   # (past the second 'S' record)
                                           inrun=not args.no_counters and s_samples > 50000)
                    rec["name"] = name
                    out["secondary"].append(rec)
                    if rec.get("parity_error"):
                        out["parity_error"] = name + ": " + rec["parity_error"]
                except Exception as e:
                    out["secondary"].append({"name": name, "error": repr(e)[:300]})
            try:
                rec = hrc_cli_record(tmp)
                out["secondary"].append(rec)
                if rec.get("parity_error"):
                    out["parity_error"] = "HRC-cli: " + rec["parity_error"]
            except Exception as e:
                out["secondary"].append({"name": "HRC-cli", "error": repr(e)[:300]})
            try:
                rec = c5_record(tmp)
                out["secondary"].append(rec)
                if rec.get("parity_error"):
                    out["parity_error"] = "C5-cli: " + rec["parity_error"]
            except Exception as e:
                out["secondary"].append({"name": "C5-cli", "error": repr(e)[:300]})
    # ---- the PRODUCT's multi-GPU path: after the ranks are done -- they have released their images and left
    # the process group -- rank 0 ALONE opens one database over all the devices through the C ABI and times its RCCL gather.
    if world > 1:
        try:
            R.release()
        except Exception:
            pass
        torch.cuda.empty_cache()
        dist.barrier()
        dist.destroy_process_group()
    devs = [int(x) for x in args.product_devices.split(",")] if args.product_devices else (list(range(world)) if world > 1 else None)
    if devs and os.environ.get("BENCH_ALL_RANKS_ON_DEVICE0") and not args.product_devices:
        devs = [0] * len(devs)                                    # (the dry run of N > 1 on a one-GPU box: N shards on its one device)
    if rank == 0 and devs and not args.no_product_sharded and headline_wl == "c2":
        with tempfile.TemporaryDirectory() as ptmp:
            # (in a process of its own, with a deadline: whatever happens there, the ranks' headline line is printed)
            prec = product_sharded_in_a_child(devs, SAMPLES["c2"], (args.sites or 1000000) * max(1, world if world > 1 else 1), SEEDS["c2"],
                                              max(3, min(args.steps, 10)), ptmp)
            out["product_sharded"] = prec
            if prec.get("parity_error"):
                out["parity_error"] = prec["parity_error"]
    if rank == 0:
        emit(out, args.detail)
    try:                                                              # (readers before their images, explicitly: not left to the
        pipe = rd = pbf = None                                        #  order the garbage collector finalises a cycle in at exit)
        R.release()
    except Exception:
        pass


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as exc:                                   # one line a first multi-GPU run can be diagnosed from
        import traceback
        print(json.dumps({"metric": "sites/sec `bgt view -G -f'AC>0'` whole-cohort scan", "value": None, "unit": "sites/s",
                          "n_gpus": int(os.environ.get("WORLD_SIZE", "1")), "error": repr(exc)[:600],
                          "failed_rank": int(os.environ.get("RANK", "0")), "failed_local_rank": int(os.environ.get("LOCAL_RANK", "0")),
                          "traceback_tail": [t[:160] for t in traceback.format_exc().splitlines()[-6:]]}, allow_nan=False), flush=True)
        raise
