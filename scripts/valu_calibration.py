#!/usr/bin/env python3
"""Issue-rate calibration behind bench.py's roofline (run on the GPU box): wave-instructions per cycle and SIMD of the
scan kernel's row-step instruction mix, of single instructions, and of the LDS gathers, at 1 / 2 / 4 waves per SIMD.
Writes gpurun_out/valu_calibration.json (copy it to profiles/<round>/ to have it judged)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bgt_amd  # noqa: E402


def main():
    L = bgt_amd.bench_lib()
    L.bgth_debug_issue_rate.restype = C.c_int
    L.bgth_debug_issue_rate.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.bgth_debug_issue_rate_name.restype = C.c_char_p
    L.bgth_debug_issue_rate_name.argtypes = [C.c_int]
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    rows = []
    mix = 0
    while True:
        name = L.bgth_debug_issue_rate_name(mix)
        if not name:
            break
        for w in (1, 2, 4):
            out = (C.c_double * 4)()
            if L.bgth_debug_issue_rate(0, mix, w, iters, out) != 0:
                raise SystemExit(bgt_amd.last_error())
            cyc, ms, valu, lds = out[0], out[1], out[2], out[3]
            rec = {"mix": mix, "name": name.decode(), "waves_per_simd": w, "iters": iters, "cycles_slowest_wave": cyc, "ms": ms,
                   "valu_per_wave": valu, "lds_per_wave": lds,
                   "valu_per_cycle_per_simd": w * valu / cyc if cyc else None,          # w waves share one SIMD
                   "cycles_per_valu": cyc / (w * valu) if valu else None,
                   "lds_cycles_per_instr_per_cu": cyc / (4 * w * lds) if lds else None,  # 4 w waves share one LDS
                   "ghz_implied": cyc / (ms * 1e6) if ms else None}
            rows.append(rec)
            print("%-52s w=%d  cyc %12.0f  %7.3f ms  %.2f GHz  VALU/cyc/SIMD %s  cyc/VALU %s  LDS cyc/instr/CU %s" % (
                rec["name"], w, cyc, ms, rec["ghz_implied"] or 0,
                "%.3f" % rec["valu_per_cycle_per_simd"] if valu else "  -  ",
                "%.2f" % rec["cycles_per_valu"] if valu else " - ",
                "%.2f" % rec["lds_cycles_per_instr_per_cu"] if lds else " - "), flush=True)
        mix += 1
    # instruction classes
    L.bgth_debug_op_rate.restype = C.c_int
    L.bgth_debug_op_rate.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    L.bgth_debug_op_rate_name.restype = C.c_char_p
    L.bgth_debug_op_rate_name.argtypes = [C.c_int]
    ops = []
    op = 0
    while True:
        name = L.bgth_debug_op_rate_name(op)
        if not name:
            break
        rec = {"op": name.decode()}
        for w in (2, 4):
            out = (C.c_double * 3)()
            if L.bgth_debug_op_rate(0, op, w, iters, out) != 0:
                raise SystemExit(bgt_amd.last_error())
            rec["cycles_per_instr_w%d" % w] = out[0] / (w * out[2])
        ops.append(rec)
        print("%-28s cycles per wave-instruction and SIMD: %.2f (2 waves/SIMD)  %.2f (4 waves/SIMD)" %
              (rec["op"], rec["cycles_per_instr_w2"], rec["cycles_per_instr_w4"]), flush=True)
        op += 1
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump({"device": "MI355X (gfx950)", "note": "one workgroup per CU, 256 CUs; cycles = s_memtime of the slowest wave",
               "runs": rows, "instruction_classes": ops}, open(os.path.join(ROOT, "gpurun_out", "valu_calibration.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
