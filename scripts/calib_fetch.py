import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bgt_amd
L = bgt_amd.lib()
L.bgth_debug_stream_read.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.c_int]
w = int(sys.argv[1]) if len(sys.argv) > 1 else 4
print("rc", L.bgth_debug_stream_read(0, 1 << 30, w, 3), "width", w, "bytes", 1 << 30)
