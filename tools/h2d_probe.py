#!/usr/bin/env python3
"""Host -> device copy rate from page-locked memory (kmd_malloc_host + kmd_memcpy_h2d), with the
NUMA placement of the GPU and of this process printed next to it."""
import ctypes as C
import glob
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmdiff_amd as K
from kmdiff_amd import _native

L = _native.lib()
for f in sorted(glob.glob("/sys/class/drm/card*/device/numa_node")):
    print(f, open(f).read().strip())
for f in sorted(glob.glob("/sys/devices/system/node/node*/cpulist")):
    print(f, open(f).read().strip())
print("affinity:", len(os.sched_getaffinity(0)), "cpus")
n = 1 << 30
h = C.c_void_p()
t0 = time.time()
_native.check(L.kmd_malloc_host(C.byref(h), n), "kmd_malloc_host")
print("page-locking 1 GiB: %.3f s" % (time.time() - t0))
C.memset(h, 1, n)
import torch
t = torch.empty(n, dtype=torch.uint8, device="cuda")
for sz in (1 << 30, 16 << 20, 1 << 20):
    reps = max(1, (1 << 30) // sz)
    _native.check(L.kmd_memcpy_h2d(C.c_void_p(t.data_ptr()), h, sz, None), "h2d")
    t0 = time.time()
    for i in range(reps):
        _native.check(L.kmd_memcpy_h2d(C.c_void_p(t.data_ptr() + i * sz), C.c_void_p(h.value + i * sz), sz, None), "h2d")
    dt = time.time() - t0
    print("H2D %4d MiB x %4d: %.1f GB/s" % (sz >> 20, reps, reps * sz / dt / 1e9))
L.kmd_free_host(h)
