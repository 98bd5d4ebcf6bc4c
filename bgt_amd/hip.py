"""ctypes mirror of include/bgt_hip.h.  No fallback: a missing library or device raises."""
import ctypes as C
import weakref
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("BGT_AMD_LIB") or os.path.join(_HERE, "lib", "libbgt_hip.so")   # (profiling builds: see csrc/Makefile)

i32p = C.POINTER(C.c_int32)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)


def library_path():
    return _LIB_PATH


def _digest(paths):
    """sha1 over the names and bytes of the given source files."""
    import hashlib
    h = hashlib.sha1()
    for f in sorted(paths):
        h.update(os.path.relpath(f, _HERE).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _built(stamp, digest, outputs):
    """A build is current when its outputs exist and the stamp written after it names these very sources.  By CONTENT, not by
    modification time: a snapshot of the tree on another machine (the GPU box) carries this machine's mtimes -- with the clocks
    a few minutes apart every `make` there found its prerequisites "newer" and recompiled (GPU tests: 80 s in the first test that
    built, ~10 s in every later one that called make)."""
    try:
        return all(os.path.exists(o) for o in outputs) and open(stamp).read().strip() == digest
    except OSError:
        return False


def _csrc_sources():
    import glob
    src = os.path.join(_HERE, "csrc")
    inc = os.path.join(os.path.dirname(_HERE), "include")
    return (glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.cpp")) + glob.glob(os.path.join(src, "*.h")) +
            glob.glob(os.path.join(src, "gen", "*.py")) + [os.path.join(src, "Makefile")] + glob.glob(os.path.join(inc, "*.h")))


def _build_settings():
    """The make variables a build depends on beside its sources: a library built for another ARCH / by another HIPCC / as the
    profiling build is not `current` for this one."""
    return "".join("%s=%s\n" % (k, os.environ.get(k, "")) for k in ("ARCH", "HIPCC", "ABLATE", "CXXFLAGS"))


def build_library(force=False):
    """Compile the gfx950 library in-tree (hipcc cross-compiles without a GPU).  Nothing happens when the libraries on disk were
    built from exactly these sources (lib/.csrc.stamp)."""
    src = os.path.join(_HERE, "csrc")
    stamp = os.path.join(_HERE, "lib", ".csrc.stamp")
    import hashlib
    digest = hashlib.sha1((_digest(_csrc_sources()) + "\n" + _build_settings()).encode()).hexdigest()
    if force and os.path.exists(_LIB_PATH):
        os.remove(_LIB_PATH)
    default_lib = os.path.join(_HERE, "lib", "libbgt_hip.so")
    if not force and _built(stamp, digest, [default_lib, os.path.join(_HERE, "lib", "libbgt_hip_bench.so")]):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", src])
    with open(stamp, "w") as f:
        f.write(digest + "\n")
    return _LIB_PATH


_BENCH_LIB = None


def bench_lib():
    """libbgt_hip_bench.so (include/bgt_hip_bench.h): the issue-rate calibration kernels behind bench.py's roofline --
    measurement tools, kept out of the product library."""
    global _BENCH_LIB
    if _BENCH_LIB is None:
        _hip_runtime_first()
        path = os.path.join(_HERE, "lib", "libbgt_hip_bench.so")
        if not os.path.exists(path):
            raise RuntimeError("bgt_amd: %s is missing -- build it with `make -C bgt_amd/csrc`" % path)
        L = C.CDLL(path)
        L.bgth_debug_issue_rate.restype = C.c_int
        L.bgth_debug_issue_rate.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.bgth_debug_issue_rate_name.restype = C.c_char_p
        L.bgth_debug_issue_rate_name.argtypes = [C.c_int]
        L.bgth_debug_op_rate.restype = C.c_int
        L.bgth_debug_op_rate.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.bgth_debug_op_rate_name.restype = C.c_char_p
        L.bgth_debug_op_rate_name.argtypes = [C.c_int]
        _BENCH_LIB = L
    return _BENCH_LIB


def build_host_shell():
    """Compile the C host shell (bgt_amd/host -> lib/libbgt.so, bin/bgt, bin/bgt-server); needs lib/libbgt_hip.so.  Nothing happens
    when what is on disk was built from exactly these sources and this device library (lib/.host.stamp)."""
    import glob
    host = os.path.join(_HERE, "host")
    inc = os.path.join(os.path.dirname(_HERE), "include")
    stamp = os.path.join(_HERE, "lib", ".host.stamp")
    srcs = (glob.glob(os.path.join(host, "*.c")) + glob.glob(os.path.join(host, "*.h")) + [os.path.join(host, "Makefile")] +
            glob.glob(os.path.join(inc, "*.h")))
    digest = _digest(srcs) + ":" + _digest(_csrc_sources())          # (libbgt.so links the device library)
    outs = [os.path.join(_HERE, "lib", "libbgt.so"), os.path.join(_HERE, "bin", "bgt"), os.path.join(_HERE, "bin", "bgt-server")]
    if _built(stamp, digest, outs):
        return outs[0]
    subprocess.check_call(["make", "-s", "-C", host])
    with open(stamp, "w") as f:
        f.write(digest + "\n")
    return outs[0]


def _hip_runtime_first():
    """torch ships its own libamdhip64.so (DT_NEEDED without the version suffix, so the dynamic loader does not
    match it with /opt/rocm's libamdhip64.so.7).  If this library pulled in the system runtime first and torch
    its own afterwards, the process would hold two HIP runtimes and the second one finds no device.  Loading
    torch first makes both resolve to the same runtime.  Processes that never import torch are unaffected."""
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def _load():
    _hip_runtime_first()
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError("bgt_amd: %s is missing -- build it with `make -C bgt_amd/csrc` "
                           "(there is no CPU fallback)" % _LIB_PATH)
    L = C.CDLL(_LIB_PATH)
    L.bgth_last_error.restype = C.c_char_p
    L.bgth_version.restype = C.c_char_p
    L.bgth_device_count.restype = C.c_int
    L.bgth_pbf_open.restype = C.c_void_p
    L.bgth_pbf_open.argtypes = [C.c_char_p, C.c_int]
    L.bgth_pbf_open_mem.restype = C.c_void_p
    L.bgth_pbf_open_mem.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    L.bgth_pbf_from_rle.restype = C.c_void_p
    L.bgth_pbf_from_rle.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
    L.bgth_pbf_save.restype = C.c_int64
    L.bgth_pbf_save.argtypes = [C.c_void_p, C.c_char_p]
    L.bgth_pbf_close.argtypes = [C.c_void_p]
    for f in ("bgth_pbf_get_m", "bgth_pbf_get_g", "bgth_pbf_get_shift"):
        getattr(L, f).restype = C.c_int
        getattr(L, f).argtypes = [C.c_void_p]
    for f in ("bgth_pbf_get_n", "bgth_pbf_hbm_bytes", "bgth_pbf_rle_bytes"):
        getattr(L, f).restype = C.c_int64
        getattr(L, f).argtypes = [C.c_void_p]
    L.bgth_reader_create.restype = C.c_void_p
    L.bgth_reader_create.argtypes = [C.c_void_p]
    L.bgth_reader_destroy.argtypes = [C.c_void_p]
    L.bgth_reader_select.restype = C.c_int
    L.bgth_reader_select.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.bgth_reader_width.restype = C.c_int
    L.bgth_reader_width.argtypes = [C.c_void_p]
    L.bgth_reader_scan.restype = C.c_int64
    L.bgth_reader_scan.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    L.bgth_reader_scan_device.restype = C.c_int64
    L.bgth_reader_scan_device.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p]
    L.bgth_reader_slot_words.restype = C.c_int
    L.bgth_reader_slot_words.argtypes = [C.c_void_p]
    L.bgth_reader_slot_map.restype = C.c_int
    L.bgth_reader_slot_map.argtypes = [C.c_void_p, C.c_void_p]
    L.bgth_reader_seek.restype = C.c_int
    L.bgth_reader_seek.argtypes = [C.c_void_p, C.c_int64]
    L.bgth_reader_read.restype = C.POINTER(u8p)
    L.bgth_reader_read.argtypes = [C.c_void_p]
    L.bgth_reader_last_counts.restype = i32p
    L.bgth_reader_last_counts.argtypes = [C.c_void_p]
    L.bgth_reader_last_timing.restype = C.c_int
    L.bgth_reader_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.bgth_reader_last_geometry.restype = C.c_int
    L.bgth_reader_last_geometry.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.bgth_pbf_final_ranks.restype = C.c_int
    L.bgth_pbf_final_ranks.argtypes = [C.c_void_p, C.c_void_p]
    L.bgth_pbf_ranks_at.restype = C.c_int
    L.bgth_pbf_ranks_at.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    L.bgth_pbf_rebase.restype = C.c_int
    L.bgth_pbf_rebase.argtypes = [C.c_void_p, C.c_void_p]
    L.bgth_reader_last_path.restype = C.c_int
    L.bgth_reader_last_path.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.bgth_force_kernels.restype = None
    L.bgth_force_kernels.argtypes = [C.c_uint]
    L.bgth_reader_tune.restype = C.c_int
    L.bgth_reader_tune.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    return L


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def last_error():
    return lib().bgth_last_error().decode()


# bgth_force_kernels (include/bgt_hip.h: BGTH_FORCE_*): kernel families forced for tests / one-shot benchmark steps
FORCE_NO_TOGGLE_ARRAY, FORCE_NO_EMPTY_PLANE_SHORTCUT, FORCE_EMPTY_PLANE_SHORTCUT, FORCE_COLUMN_ORDER = 1, 2, 4, 8
FORCE_DIRECTORY_PATH, FORCE_NO_DIRECTORY_PATH, FORCE_REBUILD_ROWS = 32, 64, 128
FORCE_SEQUENTIAL_CHECKPOINTS, FORCE_RCCL_TO_SELF, FORCE_NO_PLANE_SPLIT, FORCE_PLANE_SPLIT = 512, 1024, 2048, 4096
FORCE_THREE_PLANE_BUFFERS = 8192
_forced = 0


def force_kernels(flags=0):
    """Process-wide: OR of FORCE_* (0 = automatic).  Returns the previous value."""
    global _forced
    prev, _forced = _forced, int(flags)
    lib().bgth_force_kernels(_forced)
    return prev


class forced_kernels:
    """with forced_kernels(FORCE_DIRECTORY_PATH): ...   -- restores the previous setting on exit."""

    def __init__(self, flags):
        self.flags = flags

    def __enter__(self):
        self.prev = force_kernels(self.flags)
        return self

    def __exit__(self, *exc):
        force_kernels(self.prev)
        return False


def device_count():
    return lib().bgth_device_count()


class HipPbf:
    """A .pbf image resident in HBM (bgth_pbf_t)."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError(last_error() or "bgth_pbf_open failed")
        self.h = handle
        self._readers = weakref.WeakSet()       # readers (and encoders of nothing): destroyed BEFORE their image, whatever order the
        L = lib()                               # garbage collector finalises a cycle in (bgth_reader_destroy reads its image)
        self.m = L.bgth_pbf_get_m(handle)
        self.g = L.bgth_pbf_get_g(handle)
        self.shift = L.bgth_pbf_get_shift(handle)
        self.n = L.bgth_pbf_get_n(handle)

    @classmethod
    def open(cls, path, device=0):
        return cls(lib().bgth_pbf_open(os.fsencode(path), device))

    @classmethod
    def open_rows(cls, path, row0, row1, device=0):
        """Partial image: only the file blocks covering rows [row0,row1)."""
        L = lib()
        L.bgth_pbf_open_rows.restype = C.c_void_p
        L.bgth_pbf_open_rows.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_int]
        return cls(L.bgth_pbf_open_rows(path.encode(), row0, row1, device))

    @classmethod
    def open_sharded(cls, path, devices):
        """One database over several devices (or several shards on one): block-aligned site ranges, one partial image each."""
        L = lib()
        L.bgth_pbf_open_sharded.restype = C.c_void_p
        L.bgth_pbf_open_sharded.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
        d = np.ascontiguousarray(devices, np.int32)
        return cls(L.bgth_pbf_open_sharded(os.fsencode(path), d.size, d.ctypes.data))

    @classmethod
    def from_bytes(cls, data, device=0):
        buf = np.frombuffer(data, np.uint8)
        return cls(lib().bgth_pbf_open_mem(buf.ctypes.data, buf.size, device))

    @classmethod
    def from_rle(cls, m, shift, rle, lens, device=0):
        """rle: uint8 array of concatenated strings (row-major, plane-minor); lens: uint32 per string."""
        rle = np.ascontiguousarray(rle, np.uint8)
        lens = np.ascontiguousarray(lens, np.uint32)
        assert lens.size % 2 == 0 and int(lens.sum(dtype=np.int64)) == rle.size
        return cls(lib().bgth_pbf_from_rle(m, 2, shift, lens.size // 2, rle.ctypes.data, lens.ctypes.data, device))

    def final_ranks(self):
        """int32[2][m]: rank of every column after the last row (images built by from_rle)."""
        out = np.empty((2, self.m), np.int32)
        if lib().bgth_pbf_final_ranks(self.h, out.ctypes.data) != 0:
            raise RuntimeError(last_error())
        return out

    def ranks_at(self, row):
        """int32[2][m]: rank of every column before `row` (a checkpoint row of the image)."""
        out = np.empty((2, self.m), np.int32)
        if lib().bgth_pbf_ranks_at(self.h, row, out.ctypes.data) != 0:
            raise RuntimeError(last_error())
        return out

    def rebase(self, start_ranks):
        """Make the image start from start_ranks (int32[2][m], rank of every column before its first row) instead of the
        identity order -- how the site-range shards of one database are opened side by side."""
        st = np.ascontiguousarray(start_ranks, np.int32)
        assert st.shape == (2, self.m)
        if lib().bgth_pbf_rebase(self.h, st.ctypes.data) != 0:
            raise RuntimeError(last_error())

    def save(self, path):
        n = lib().bgth_pbf_save(self.h, os.fsencode(path))
        if n < 0:
            raise RuntimeError(last_error())
        return n

    @property
    def n_shards(self):
        L = lib()
        L.bgth_pbf_n_shards.restype = C.c_int
        L.bgth_pbf_n_shards.argtypes = [C.c_void_p]
        return L.bgth_pbf_n_shards(self.h)

    @property
    def hbm_bytes(self):
        return lib().bgth_pbf_hbm_bytes(self.h)

    @property
    def rle_bytes(self):
        return lib().bgth_pbf_rle_bytes(self.h)

    @property
    def unit_rows(self):
        """rows per sub-block = the spacing of the image's rank checkpoints = the unit of work of a launch"""
        L = lib()
        L.bgth_pbf_unit_rows.restype = C.c_int64
        L.bgth_pbf_unit_rows.argtypes = [C.c_void_p]
        return L.bgth_pbf_unit_rows(self.h)

    def close(self):
        if self.h:
            for r in list(self._readers):
                r.close()
            lib().bgth_pbf_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipReader:
    """Per-reader device state (bgth_reader_t)."""

    def __init__(self, pbf):
        self.pbf = pbf
        self.h = lib().bgth_reader_create(pbf.h)
        if not self.h:
            raise RuntimeError(last_error())
        self.n_groups = 1
        pbf._readers.add(self)

    def close(self):
        if self.h:
            if self.pbf.h:                       # (an image closed first has taken its readers with it)
                lib().bgth_reader_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def select(self, cols=None, group=None, n_groups=1):
        c = g = None
        n = 0
        if cols is not None:
            c = np.ascontiguousarray(cols, np.int32)
            n = c.size
        if group is not None:
            g = np.ascontiguousarray(group, np.uint32)
        rc = lib().bgth_reader_select(self.h, n, c.ctypes.data if c is not None else None,
                                      g.ctypes.data if g is not None else None, n_groups)
        if rc < 0:
            raise RuntimeError(last_error())
        self.n_groups = n_groups if group is not None else 1

    @property
    def width(self):
        return lib().bgth_reader_width(self.h)

    @property
    def count_entries(self):
        return 1 + (self.n_groups if self.n_groups > 1 else 0)

    def tune(self, threads=0, cols_per_thread=0, rows_per_batch=0):
        lib().bgth_reader_tune(self.h, threads, cols_per_thread, rows_per_batch)

    def scan(self, row0, row1, want_gt=False):
        rows = row1 - row0
        counts = np.zeros((rows, self.count_entries, 3), np.int32)
        gt = np.zeros((rows, (self.width + 3) // 4), np.uint8) if want_gt else None
        n = lib().bgth_reader_scan(self.h, row0, row1, counts.ctypes.data, gt.ctypes.data if want_gt else None)
        if n < 0:
            raise RuntimeError(last_error())
        return (counts, gt) if want_gt else counts

    def scan_device(self, row0, row1, d_counts_ptr, d_h0_ptr=None, d_h1_ptr=None, stream=None):
        n = lib().bgth_reader_scan_device(self.h, row0, row1, d_counts_ptr, d_h0_ptr, d_h1_ptr, stream)
        if n < 0:
            raise RuntimeError(last_error())
        return n

    @property
    def slot_words(self):
        return lib().bgth_reader_slot_words(self.h)

    def slot_map(self):
        out = np.zeros(self.width, np.int32)
        lib().bgth_reader_slot_map(self.h, out.ctypes.data)
        return out

    def seek(self, row):
        if lib().bgth_reader_seek(self.h, row) < 0:
            raise RuntimeError(last_error())

    WANT_PLANES, WANT_GT8, WANT_GTTEXT, WANT_BITS = 1, 2, 4, 8

    def config(self, want=1, max_rows_ahead=0):
        """What the pull interface materialises per row besides the counts (mask of WANT_*)."""
        L = lib()
        L.bgth_reader_config.restype = C.c_int
        L.bgth_reader_config.argtypes = [C.c_void_p, C.c_int, C.c_int64]
        if L.bgth_reader_config(self.h, want, max_rows_ahead) < 0:
            raise RuntimeError(last_error())
        self._want = want

    def read(self):
        """Next row: the byte planes (uint8[2][width]; uint8[g][width] for an image of more than two planes), or True when planes
        are not configured; None at the end."""
        r = lib().bgth_reader_read(self.h)
        if not r:
            return None
        if not (getattr(self, "_want", 1) & 1):
            return True
        w = self.width
        return np.stack([np.ctypeslib.as_array(r[k], (w,)).copy() for k in range(max(2, self.pbf.g))])

    def last_gt8(self):
        L = lib()
        L.bgth_reader_last_gt8.restype = C.POINTER(C.c_int8)
        L.bgth_reader_last_gt8.argtypes = [C.c_void_p]
        p = L.bgth_reader_last_gt8(self.h)
        return np.ctypeslib.as_array(p, (self.width,)).copy() if p else None

    def last_gt_text(self):
        L = lib()
        L.bgth_reader_last_gt_text.restype = C.POINTER(C.c_char)
        L.bgth_reader_last_gt_text.argtypes = [C.c_void_p]
        p = L.bgth_reader_last_gt_text(self.h)
        return C.string_at(p, 2 * self.width) if p else None

    def fold_last(self, code=-1, bit=-1):
        """Allele-set reductions over the row just read (reader configured with WANT_BITS; reference bgt.c:859-876):
        carriers[s] += sample s has a haplotype of `code`; hap[i] |= 1 << bit for haplotypes of code 1."""
        L = lib()
        L.bgth_reader_fold_last.restype = C.c_int
        L.bgth_reader_fold_last.argtypes = [C.c_void_p, C.c_int, C.c_int]
        if L.bgth_reader_fold_last(self.h, code, bit) < 0:
            raise RuntimeError(last_error())

    def take_folds(self):
        """(carriers int32[width/2], hap uint64[width]) accumulated since the selection / the last take."""
        L = lib()
        L.bgth_reader_take_folds.restype = C.c_int
        L.bgth_reader_take_folds.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        car = np.zeros(self.width // 2, np.int32)
        hap = np.zeros(self.width, np.uint64)
        if L.bgth_reader_take_folds(self.h, car.ctypes.data, hap.ctypes.data) < 0:
            raise RuntimeError(last_error())
        return car, hap

    def last_counts(self):
        p = lib().bgth_reader_last_counts(self.h)
        return np.ctypeslib.as_array(p, (self.count_entries, 3)).copy()

    def timing(self):
        t = (C.c_float * 3)()
        lib().bgth_reader_last_timing(self.h, t)
        return {"scan_ms": t[0], "finalize_ms": t[1], "total_ms": t[2]}

    def shard_timing(self, shard):
        t = (C.c_float * 3)()
        L = lib()
        L.bgth_reader_shard_timing.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        if L.bgth_reader_shard_timing(self.h, shard, t) < 0:
            raise IndexError("no shard %d" % shard)
        return {"scan_ms": t[0], "finalize_ms": t[1], "total_ms": t[2]}

    def path(self):
        t = (C.c_float * 4)()
        lib().bgth_reader_last_path(self.h, t)
        return {"directory_path": (int(t[0]) & 3) == 1, "plane_split": (int(t[0]) & 3) == 2,
                "passes": int(t[1]), "producer_launches": int(t[2]), "producer_ms": t[3]}

    def geometry(self):
        g = (C.c_int * 6)()
        lib().bgth_reader_last_geometry(self.h, g)
        return dict(zip(("threads", "cols_per_thread", "slices", "rows_per_batch", "lds_bytes", "workgroups"), g))


# ---------------------------------------------------------------------------------------------------
# synthetic cohorts (include/bgt_synth.h)
# ---------------------------------------------------------------------------------------------------
class HipEncoder:
    """The device writer (bgth_encoder_t): rows of 2-bit codes -> the bytes of a .pbf, identical to what the
    reference writer produces (pbf_open_w / pbf_write / pbf_close, ref pbwt.c:199-311)."""

    def __init__(self, m, g=2, shift=13, device=0):
        L = lib()
        L.bgth_encoder_open.restype = C.c_void_p
        L.bgth_encoder_open.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int]
        L.bgth_encoder_write.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.bgth_encoder_write_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.bgth_encoder_take.restype = C.c_int64
        L.bgth_encoder_take.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.bgth_encoder_finish.restype = C.c_int64
        L.bgth_encoder_finish.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.bgth_encoder_free_image.argtypes = [C.c_void_p]
        L.bgth_encoder_close.argtypes = [C.c_void_p]
        L.bgth_encoder_kernel_ms.restype = C.c_double
        L.bgth_encoder_kernel_ms.argtypes = [C.c_void_p]
        L.bgth_encoder_last_error.restype = C.c_char_p
        self.m = m
        self.h = L.bgth_encoder_open(m, g, shift, device)
        if not self.h:
            raise RuntimeError(L.bgth_encoder_last_error().decode() or "bgth_encoder_open failed")

    def write(self, codes):
        """codes: (rows, m) uint8, bit k = plane k."""
        codes = np.ascontiguousarray(codes, np.uint8)
        assert codes.ndim == 2 and codes.shape[1] == self.m
        if lib().bgth_encoder_write(self.h, codes.ctypes.data, codes.shape[0]) < 0:
            raise RuntimeError(lib().bgth_encoder_last_error().decode())

    def write_packed(self, packed):
        """packed: (rows, (m + 3) // 4) uint8, four 2-bit codes per byte -- the genotype rows HipReader.scan(want_gt=True) returns."""
        packed = np.ascontiguousarray(packed, np.uint8)
        assert packed.ndim == 2 and packed.shape[1] == (self.m + 3) // 4
        if lib().bgth_encoder_write_packed(self.h, packed.ctypes.data, packed.shape[0]) < 0:
            raise RuntimeError(lib().bgth_encoder_last_error().decode())

    def take(self):
        """The bytes of the file produced so far; the encoder forgets them (stream them to the file)."""
        out = C.c_void_p()
        n = lib().bgth_encoder_take(self.h, C.byref(out))
        if n < 0:
            raise RuntimeError(lib().bgth_encoder_last_error().decode())
        data = C.string_at(out, n)
        lib().bgth_encoder_free_image(out)
        return data

    def finish(self):
        """The rest of the image (everything not taken yet, and the footer) as bytes."""
        out = C.c_void_p()
        n = lib().bgth_encoder_finish(self.h, C.byref(out))
        if n < 0:
            raise RuntimeError(lib().bgth_encoder_last_error().decode())
        data = C.string_at(out, n)
        lib().bgth_encoder_free_image(out)
        return data

    @property
    def kernel_ms(self):
        return lib().bgth_encoder_kernel_ms(self.h)

    def close(self):
        if self.h:
            lib().bgth_encoder_close(self.h)
            self.h = None

    def __del__(self):
        self.close()


def shard_ranges(n_rows, shift, n_shards):
    """[(row0, row1)] per shard as the C ABI deals them out (bgth_shard_ranges); needs no device."""
    L = lib()
    L.bgth_shard_ranges.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_void_p]
    out = np.zeros(2 * n_shards, np.int64)
    L.bgth_shard_ranges(n_rows, shift, n_shards, out.ctypes.data)
    return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(n_shards)]


def synth_rows(m, row0, n_rows, seed, n_threads=0):
    """Draw rows [row0,row0+n_rows) of cohort (seed, m) in the PBWT domain. Returns (rle uint8[], len uint32[2n])."""
    L = lib()
    L.bgth_synth_rows.restype = C.c_void_p
    L.bgth_synth_rows.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_uint64, C.c_int]
    L.bgth_synth_rle.restype = C.c_void_p
    L.bgth_synth_rle.argtypes = [C.c_void_p]
    L.bgth_synth_len.restype = C.c_void_p
    L.bgth_synth_len.argtypes = [C.c_void_p]
    L.bgth_synth_bytes.restype = C.c_int64
    L.bgth_synth_bytes.argtypes = [C.c_void_p]
    L.bgth_synth_free.argtypes = [C.c_void_p]
    h = L.bgth_synth_rows(m, row0, n_rows, seed, n_threads)
    if not h:
        raise RuntimeError("bgth_synth_rows failed")
    try:
        nb = L.bgth_synth_bytes(h)
        rle = np.ctypeslib.as_array(C.cast(L.bgth_synth_rle(h), u8p), (max(nb, 1),))[:nb].copy()
        lens = np.ctypeslib.as_array(C.cast(L.bgth_synth_len(h), u32p), (2 * n_rows,)).copy()
    finally:
        L.bgth_synth_free(h)
    return rle, lens


# ---------------------------------------------------------------------------------------------------
# site filter on the device (bgth_filter_*): expression parsed by the host shell's parser (libbgt.so)
# ---------------------------------------------------------------------------------------------------
_HOST_LIB_PATH = os.path.join(_HERE, "lib", "libbgt.so")
_host_lib = None


def host_lib():
    """libbgt.so: the reader API / expression parser of the host shell (bgt_amd/host)."""
    global _host_lib
    if _host_lib is None:
        _hip_runtime_first()
        if not os.path.exists(_HOST_LIB_PATH):                  # plain C, seconds to build: do it rather than fail
            try:
                build_host_shell()
            except (OSError, subprocess.CalledProcessError) as e:
                raise RuntimeError("bgt_amd: %s is missing and `make -C bgt_amd/host` failed: %s" % (_HOST_LIB_PATH, e))
        H = C.CDLL(_HOST_LIB_PATH)
        H.ke_parse.restype = C.c_void_p
        H.ke_parse.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
        H.ke_destroy.argtypes = [C.c_void_p]
        H.ke_export.restype = C.c_int
        H.ke_export.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _host_lib = H
    return _host_lib


def count_slot(name, n_groups=1):
    """Index of variable `name` in one site's counts vector int32[1+Gx][3] (AN, AC, AN<g>, AC<g>); -1 = unbound."""
    for prefix, field in (("AN", 0), ("AC", 1)):
        if name == prefix:
            return field
        if name.startswith(prefix) and name[2:].isdigit() and not name[2:].startswith("0"):
            g = int(name[2:])
            if n_groups > 1 and 1 <= g <= n_groups:
                return 3 * g + field
    return -1


class HipFilter:
    """A `-f` site filter compiled for the device: flags[site] = expression(counts[site]) != 0."""
    MAX_ITEMS = 48

    def __init__(self, expr, n_groups=1, device=0):
        H, L = host_lib(), lib()
        err = C.c_int(0)
        ke = H.ke_parse(expr.encode(), C.byref(err))
        if not ke or err.value:
            raise ValueError("cannot parse filter %r (error 0x%x)" % (expr, err.value))
        try:
            n = self.MAX_ITEMS
            op = np.zeros(n, np.int32); iv = np.zeros(n, np.int64); rv = np.zeros(n, np.float64)
            names = (C.c_char_p * n)()
            k = H.ke_export(ke, n, op.ctypes.data, iv.ctypes.data, rv.ctypes.data, names)
            if k <= 0:
                raise ValueError("filter %r cannot run on the device (strings, functions or > %d items)" % (expr, n))
            slot = np.full(n, -1, np.int32)
            for i in range(k):
                if op[i] == 2:
                    slot[i] = count_slot(names[i].decode(), n_groups)
        finally:
            H.ke_destroy(ke)
        L.bgth_filter_create.restype = C.c_void_p
        L.bgth_filter_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.bgth_filter_destroy.argtypes = [C.c_void_p]
        L.bgth_filter_apply_device.restype = C.c_int
        L.bgth_filter_apply_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                               C.c_void_p]
        self.expr, self.n_items = expr, k
        self.h = L.bgth_filter_create(device, k, op.ctypes.data, iv.ctypes.data, rv.ctypes.data, slot.ctypes.data)
        if not self.h:
            raise RuntimeError(last_error())

    def apply_device(self, d_counts_ptr, n_rows, ints_per_row, d_flags_ptr, d_n_pass_ptr, stream=None):
        if lib().bgth_filter_apply_device(self.h, d_counts_ptr, n_rows, ints_per_row, d_flags_ptr, d_n_pass_ptr,
                                          stream) != 0:
            raise RuntimeError(last_error())

    def close(self):
        if getattr(self, "h", None):
            lib().bgth_filter_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
