"""ctypes binding of the CPU oracle (oracle/liborc.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; nothing under
bgt_amd/ does.  The library is (re)built on demand with `make -C oracle liborc.so` (plain gcc).
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = os.path.join(ROOT, "oracle", "liborc.so")


def _build():
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("orc_pbwt.c", "orc_scan.c", "orc.h")]
    if (not os.path.exists(_LIB)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liborc.so"])


_build()
lib = C.CDLL(_LIB)

u8p = C.POINTER(C.c_uint8)
i32p = C.POINTER(C.c_int32)
u32p = C.POINTER(C.c_uint32)

lib.orc_rle_len.restype = C.c_uint32
lib.orc_rle_len.argtypes = [C.c_uint8]
lib.orc_rle_put_run.restype = C.c_int
lib.orc_rle_put_run.argtypes = [u8p, C.c_uint32, C.c_int]
lib.orc_rle_encode.restype = C.c_int
lib.orc_rle_encode.argtypes = [C.c_int, u8p, u8p]
lib.orc_rle_count_ones.restype = C.c_int64
lib.orc_rle_count_ones.argtypes = [u8p, C.c_int]
lib.orc_pbf_open.restype = C.c_void_p
lib.orc_pbf_open.argtypes = [u8p, C.c_size_t]
lib.orc_pbf_close.argtypes = [C.c_void_p]
for f in ("orc_pbf_m", "orc_pbf_g", "orc_pbf_shift", "orc_pbf_subset_width"):
    getattr(lib, f).restype = C.c_int
    getattr(lib, f).argtypes = [C.c_void_p]
lib.orc_pbf_n.restype = C.c_int64
lib.orc_pbf_n.argtypes = [C.c_void_p]
lib.orc_pbf_tell.restype = C.c_int64
lib.orc_pbf_tell.argtypes = [C.c_void_p]
lib.orc_pbf_subset.restype = C.c_int
lib.orc_pbf_subset.argtypes = [C.c_void_p, C.c_int, i32p]
lib.orc_pbf_seek.restype = C.c_int
lib.orc_pbf_seek.argtypes = [C.c_void_p, C.c_int64]
lib.orc_pbf_read.restype = C.POINTER(u8p)
lib.orc_pbf_read.argtypes = [C.c_void_p]
lib.orc_pbf_perm.restype = i32p
lib.orc_pbf_perm.argtypes = [C.c_void_p, C.c_int]
lib.orc_pbw_new.restype = C.c_void_p
lib.orc_pbw_new.argtypes = [C.c_int, C.c_int, C.c_int]
lib.orc_pbw_row.restype = C.c_int
lib.orc_pbw_row.argtypes = [C.c_void_p, C.POINTER(u8p)]
lib.orc_pbw_finish.restype = C.c_size_t
lib.orc_pbw_finish.argtypes = [C.c_void_p, C.POINTER(u8p)]
lib.orc_allele_counts.argtypes = [C.c_int, u8p, u8p, u32p, C.c_int, i32p]
lib.orc_scan.restype = C.c_int64
lib.orc_scan.argtypes = [C.c_void_p, C.c_int64, C.c_int64, u32p, C.c_int, i32p, u8p]

_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def _p(a, t):
    return a.ctypes.data_as(t)


def rle_len(byte):
    return lib.orc_rle_len(byte)


def rle_put_run(length, bit):
    buf = np.zeros(8, np.uint8)
    n = lib.orc_rle_put_run(_p(buf, u8p), length, bit)
    return bytes(buf[:n])


def rle_encode(bits):
    bits = np.ascontiguousarray(bits, np.uint8)
    out = np.zeros(bits.size + 1, np.uint8)
    n = lib.orc_rle_encode(bits.size, _p(bits, u8p), _p(out, u8p))
    return bytes(out[:n])


def encode_pbf(mat, g=2, shift=13):
    """mat: (rows, m) array of codes; plane k = bit k of the code. Returns PBF bytes (oracle writer)."""
    mat = np.ascontiguousarray(mat)
    rows, m = mat.shape
    w = lib.orc_pbw_new(m, g, shift)
    planes = (u8p * g)()
    bufs = [np.zeros(m, np.uint8) for _ in range(g)]
    for k in range(g):
        planes[k] = _p(bufs[k], u8p)
    for r in range(rows):
        for k in range(g):
            bufs[k][:] = (mat[r] >> k) & 1
        lib.orc_pbw_row(w, planes)
    out = u8p()
    n = lib.orc_pbw_finish(w, C.byref(out))
    data = C.string_at(out, n)
    _libc.free(out)
    return data


class Pbf:
    """Oracle PBF reader over a bytes object (mirrors pbf_open_r/pbf_subset/pbf_seek/pbf_read)."""

    def __init__(self, data):
        self._buf = np.frombuffer(data, np.uint8).copy()
        self.h = lib.orc_pbf_open(_p(self._buf, u8p), self._buf.size)
        if not self.h:
            raise ValueError("not a PBF image")
        self.m = lib.orc_pbf_m(self.h)
        self.g = lib.orc_pbf_g(self.h)
        self.shift = lib.orc_pbf_shift(self.h)
        self.n = lib.orc_pbf_n(self.h)

    def close(self):
        if self.h:
            lib.orc_pbf_close(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def subset(self, cols):
        cols = np.ascontiguousarray(cols, np.int32)
        self._cols = cols
        return lib.orc_pbf_subset(self.h, cols.size, _p(cols, i32p))

    def seek(self, row):
        return lib.orc_pbf_seek(self.h, row)

    def tell(self):
        return lib.orc_pbf_tell(self.h)

    def width(self):
        return lib.orc_pbf_subset_width(self.h)

    def read(self):
        r = lib.orc_pbf_read(self.h)
        if not r:
            return None
        w = self.width()
        return np.stack([np.ctypeslib.as_array(r[k], (w,)).copy() for k in range(self.g)])

    def perm(self, plane):
        return np.ctypeslib.as_array(lib.orc_pbf_perm(self.h, plane), (self.m,)).copy()

    def scan(self, row0, row1, group=None, n_groups=1, want_gt=False):
        g_out = n_groups if n_groups > 1 else 0
        counts = np.zeros((row1 - row0, 3 * (1 + g_out)), np.int32)
        w = self.width()
        gt = np.zeros((row1 - row0, (w + 3) // 4), np.uint8) if want_gt else None
        grp = None
        if group is not None:
            grp = np.ascontiguousarray(group, np.uint32)
        ret = lib.orc_scan(self.h, row0, row1, _p(grp, u32p) if grp is not None else None, n_groups,
                           _p(counts, i32p), _p(gt, u8p) if want_gt else None)
        if ret < 0:
            raise RuntimeError("orc_scan failed: %d" % ret)
        return (counts, gt) if want_gt else counts


def allele_counts(a0, a1, group=None, n_groups=1):
    a0 = np.ascontiguousarray(a0, np.uint8)
    a1 = np.ascontiguousarray(a1, np.uint8)
    g_out = n_groups if n_groups > 1 else 0
    out = np.zeros(3 * (1 + g_out), np.int32)
    grp = np.ascontiguousarray(group, np.uint32) if group is not None else None
    lib.orc_allele_counts(a0.size, _p(a0, u8p), _p(a1, u8p), _p(grp, u32p) if grp is not None else None,
                          n_groups, _p(out, i32p))
    return out
