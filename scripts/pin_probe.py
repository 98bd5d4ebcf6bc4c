import ctypes as C, time, threading
hip = C.CDLL("libamdhip64.so")
hip.hipSetDevice(0)
p = C.c_void_p(); hip.hipMalloc(C.byref(p), 1 << 30)
def t_alloc(mb):
    q = C.c_void_p(); t0 = time.perf_counter(); rc = hip.hipHostMalloc(C.byref(q), mb << 20, 0); t1 = time.perf_counter()
    return q, (t1 - t0) * 1e3, rc
for mb in (5, 19, 64, 300):
    q, ms, rc = t_alloc(mb)
    t0 = time.perf_counter(); hip.hipMemcpy(p, q, min(mb, 1024) << 20, 1); t1 = time.perf_counter()
    t2 = time.perf_counter(); hip.hipMemcpy(p, q, min(mb, 1024) << 20, 1); t3 = time.perf_counter()
    print("hipHostMalloc %d MB: %.1f ms (rc %d); copy %.1f GB/s, again %.1f GB/s" % (mb, ms, rc, mb / 1024 / (t1 - t0), mb / 1024 / (t3 - t2)))
    hip.hipHostFree(q)
# 16 threads each allocating 19 MB
res = []
def w():
    q, ms, rc = t_alloc(19); res.append(ms)
t0 = time.perf_counter(); th = [threading.Thread(target=w) for _ in range(16)]; [t.start() for t in th]; [t.join() for t in th]
print("16 threads x 19 MB: wall %.1f ms, per call %s" % ((time.perf_counter() - t0) * 1e3, [round(x) for x in res]))
import numpy as np
a = np.ones(300 << 20, np.uint8)
t0 = time.perf_counter(); hip.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), a.size, 1); t1 = time.perf_counter()
print("pageable 300 MB: %.1f GB/s" % (0.3 / (t1 - t0)))
