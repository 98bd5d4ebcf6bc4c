import ctypes as C, sys
sys.path.insert(0, '.')
import bgt_amd
L = bgt_amd.bench_lib()
L.bgth_debug_issue_rate.restype = C.c_int
L.bgth_debug_issue_rate.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
L.bgth_debug_issue_rate_name.restype = C.c_char_p
L.bgth_debug_issue_rate_name.argtypes = [C.c_int]
for mix in (7, 11, 12, 13, 2):
    for w in (1, 2, 4):
        out = (C.c_double * 4)()
        L.bgth_debug_issue_rate(0, mix, w, 10000, out)
        print("%-60s w=%d cycles/VALU %.2f" % (L.bgth_debug_issue_rate_name(mix).decode(), w, out[0] / (w * out[2])), flush=True)
