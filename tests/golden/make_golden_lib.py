"""ctypes access to the COMPILED reference (oracle/_ref/libbgt_ref.so): writer and scenario replay.
Used by make_golden.py (fixture generation) and by the live oracle-vs-reference test."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")

if os.path.exists("/root/reference/pbwt.c"):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
ref = C.CDLL(os.path.join(REF, "libbgt_ref.so"))
u8p = C.POINTER(C.c_uint8)
ref.pbf_open_w.restype = C.c_void_p
ref.pbf_open_w.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
ref.pbf_open_r.restype = C.c_void_p
ref.pbf_open_r.argtypes = [C.c_char_p]
ref.pbf_close.argtypes = [C.c_void_p]
ref.pbf_write.argtypes = [C.c_void_p, C.POINTER(u8p)]
ref.pbf_read.restype = C.POINTER(u8p)
ref.pbf_read.argtypes = [C.c_void_p]
ref.pbf_seek.argtypes = [C.c_void_p, C.c_uint64]
ref.pbf_subset.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]


def ref_write_pbf(path, mat, shift, g=2):
    rows, m = mat.shape
    w = ref.pbf_open_w(path.encode(), m, g, shift)
    bufs = [np.zeros(m, np.uint8) for _ in range(g)]
    planes = (u8p * g)(*[b.ctypes.data_as(u8p) for b in bufs])
    for r in range(rows):
        for k in range(g):
            bufs[k][:] = (mat[r] >> k) & 1
        ref.pbf_write(w, planes)
    ref.pbf_close(w)


def ref_replay(path, m, g, ops):
    """Run a scenario on the reference reader; returns uint8 array (n_rows_read, g, width)."""
    p = ref.pbf_open_r(path.encode())
    width = m
    out = []
    for op in ops:
        if op[0] == "subset":
            cols = (C.c_int * len(op[1]))(*op[1])
            ref.pbf_subset(p, len(op[1]), cols)
            width = len(op[1]) if 0 < len(op[1]) < m else m
        elif op[0] == "seek":
            ref.pbf_seek(p, op[1])
        elif op[0] == "read":
            for _ in range(op[1]):
                a = ref.pbf_read(p)
                if not a:
                    break
                out.append(np.stack([np.ctypeslib.as_array(a[k], (width,)).copy() for k in range(g)]))
    ref.pbf_close(p)
    return np.stack(out) if out else np.zeros((0, g, width), np.uint8)


