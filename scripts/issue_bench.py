#!/usr/bin/env python3
"""Runs the VALU issue experiments of issue_bench.hip (generated at build time by bgt_amd/csrc/gen/gen_issue_bench.py) on the GPU
box and writes gpurun_out/issue_bench.json + a text table (copy both to profiles/r04_issue/).
Usage: python scripts/issue_bench.py [iters] [comma-separated name prefixes]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bgt_amd  # noqa: E402


def main():
    L = bgt_amd.bench_lib()
    L.bgth_issue_bench_count.restype = C.c_int
    L.bgth_issue_bench_name.restype = C.c_char_p
    L.bgth_issue_bench_name.argtypes = [C.c_int]
    L.bgth_issue_bench_note.restype = C.c_char_p
    L.bgth_issue_bench_note.argtypes = [C.c_int]
    L.bgth_issue_bench_run.restype = C.c_int
    L.bgth_issue_bench_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = []
    lines = ["%-118s %7s %7s %7s %7s   (cycles per VALU wave-instruction and SIMD at 1 / 2 / 4 / 8 waves per SIMD)" % ("experiment", "w=1", "w=2", "w=4", "w=8")]
    for i in range(L.bgth_issue_bench_count()):
        name = L.bgth_issue_bench_name(i).decode()
        if flt and not any(name.startswith(f) for f in flt.split(",")):
            continue
        rec = {"id": i, "name": name, "note": L.bgth_issue_bench_note(i).decode()}
        cells = []
        for w in (1, 2, 4, 8):
            out = (C.c_double * 5)()
            if w == 8 and name.startswith("7"):          # > 64 VGPRs: two workgroups do not share a CU
                cells.append("   -   ")
                continue
            if L.bgth_issue_bench_run(0, i, w, iters, out) != 0:
                rec["w%d" % w] = None
                cells.append("   -   ")
                continue
            cyc, ms, n = out[0], out[1], out[2]
            if out[4]:
                raise SystemExit("%s: %d waves left the row -- the step is wrong" % (name, out[4]))
            if out[3]:                                   # two streams on one SIMD: cycles of each kind of wave
                rec["w%d_streamY" % w] = out[3] / (w * n)
                cyc = max(cyc, out[3])
                rec["w%d_streamX" % w] = out[0] / (w * n)
            rec["w%d" % w] = cyc / (w * n)
            rec["ghz_w%d" % w] = cyc / (ms * 1e6) if ms else None
            rec["valu_per_iter"] = n / iters
            cells.append("%7.2f" % rec["w%d" % w] if not out[3] else "%3.1f|%3.1f" % (rec["w%d_streamX" % w], rec["w%d_streamY" % w]))
        rows.append(rec)
        lines.append("%-118s %s   %s" % (name, " ".join(cells), rec["note"]))
        print(lines[-1], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump({"device": "MI355X (gfx950)", "iters": iters, "rows": rows}, open(os.path.join(ROOT, "gpurun_out", "issue_bench.json"), "w"), indent=1)
    open(os.path.join(ROOT, "gpurun_out", "issue_bench.txt"), "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
