#!/usr/bin/env python3
"""dev (GPU box): what page-locking costs -- 80 arrays of 8 MB one after the other, from 16 threads at once, and as 2 of 320 MB."""
import ctypes as C, os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import kmdiff_amd as K
L = K._native.lib()
L.kmd_malloc_host.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; L.kmd_free_host.argtypes = [C.c_void_p]
K.DeviceBuffer.from_host(__import__("numpy").zeros(16, dtype="u1"))      # runtime up
def alloc(n, size):
    ps = []
    for _ in range(n):
        p = C.c_void_p(); assert L.kmd_malloc_host(C.byref(p), size) == 0; ps.append(p)
    return ps
def free(ps):
    for p in ps: L.kmd_free_host(p)
for rep in range(2):
    t0 = time.perf_counter(); a = alloc(80, 8 << 20); t1 = time.perf_counter(); free(a); t2 = time.perf_counter()
    print("80 x 8 MB, one thread: alloc %.3f s, free %.3f s" % (t1 - t0, t2 - t1))
    res = [None] * 16
    def work(i): res[i] = alloc(5, 8 << 20)
    t0 = time.perf_counter(); th = [threading.Thread(target=work, args=(i,)) for i in range(16)]; [t.start() for t in th]; [t.join() for t in th]; t1 = time.perf_counter()
    for r in res: free(r)
    t2 = time.perf_counter()
    print("80 x 8 MB, 16 threads: alloc %.3f s, free %.3f s" % (t1 - t0, t2 - t1))
    t0 = time.perf_counter(); a = alloc(2, 320 << 20); t1 = time.perf_counter(); free(a); t2 = time.perf_counter()
    print("2 x 320 MB: alloc %.3f s, free %.3f s" % (t1 - t0, t2 - t1))
    t0 = time.perf_counter(); a = alloc(1, 640 << 20); t1 = time.perf_counter(); free(a); t2 = time.perf_counter()
    print("1 x 640 MB: alloc %.3f s, free %.3f s" % (t1 - t0, t2 - t1))
