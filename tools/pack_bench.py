#!/usr/bin/env python3
"""Host-side packing rate of kmd_pack_stream (no GPU needed): ns per record on one core, synthetic sorted k-mers with the
delta widths of a partition's sample stream (22-26 bits) and counts below 255 but for one in a thousand."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kmdiff_amd import _native

lib = _native.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
rng = np.random.default_rng(3)
for bits in (24, 40):
    kmers = np.cumsum(rng.integers(1, 1 << bits, n, dtype=np.uint64)).astype(np.uint64)
    counts = rng.integers(1, 60, n, dtype=np.uint32)
    counts[rng.integers(0, n, n // 1000)] = 70000
    out = np.zeros(n * 13 + 4096, dtype=np.uint8)
    offs = np.zeros(n // 256 + 2, dtype=np.uint32)
    lib.kmd_pack_stream.restype = C.c_size_t
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        got = lib.kmd_pack_stream(C.c_void_p(kmers.ctypes.data), C.c_void_p(counts.ctypes.data), C.c_size_t(n), C.c_void_p(out.ctypes.data),
                                  C.c_size_t(out.nbytes), C.c_void_p(offs.ctypes.data))
        best = min(best, time.perf_counter() - t0)
    assert got > 0
    print("kmd_pack_stream: %d records, deltas of <= %d bits: %.2f ns per record, %.2f bytes per record (checksum %d)"
          % (n, bits, best / n * 1e9, got / n, int(out[:got].astype(np.uint64).sum())))
