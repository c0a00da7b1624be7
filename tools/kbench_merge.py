#!/usr/bin/env python3
"""K2 micro-benchmark: kmd_merge_partition on streams cut out of a synthetic matrix."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmdiff_amd as K

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=4_000_000)
ap.add_argument("--nc", type=int, default=20)
ap.add_argument("--nk", type=int, default=20)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--layout", default="tiled")
ap.add_argument("--limbs", type=int, default=1, help="2: two-limb k-mers (32 < k <= 64)")
ap.add_argument("--sparse", type=float, default=1.0, help="keep each record with this probability (rows of few records: private k-mers)")
ap.add_argument("--keys", default="even", help="even: the generator's evenly spaced k-mers | random: uniform random (Poisson-sized buckets)")
a = ap.parse_args()
S = a.nc + a.nk
lib = K._native.lib()
mat = K.synth_matrix(0x6B6D64696666, 0, a.rows, a.nc, a.nk, 4, K.LAYOUT_ROWS)
host = mat.to_host()
lo = mat.kmers_to_host()[0]
if a.keys == "random":
    lo = np.unique(np.random.default_rng(5).integers(0, 1 << 62, int(a.rows * 1.02), dtype=np.uint64))[:a.rows]
    assert len(lo) == a.rows
if a.keys == "clustered":   # 2000 dense clusters (consecutive values) scattered over the range: what minimizer partitions may look like
    rng = np.random.default_rng(6)
    starts = np.sort(rng.integers(0, 1 << 61, 2000, dtype=np.uint64))
    per = a.rows // 2000 + 1
    lo = np.unique((starts[:, None] + np.arange(per, dtype=np.uint64)[None, :] * np.uint64(3)).ravel())[:a.rows]
    assert len(lo) == a.rows
hi = None
if a.limbs == 2:          # the same order as 128-bit keys: hi = top bits, lo = the rest moved to the top of the low limb
    hi = lo >> np.uint64(20)
    lo = (lo & np.uint64(0xFFFFF)) << np.uint64(44)
offs = np.zeros(S + 1, dtype=np.uint64)
ks, cs, hs = [], [], []
any_kept = np.zeros(a.rows, dtype=bool)
rng_sp = np.random.default_rng(11)
for s in range(S):
    sel = host[:, s] > 0
    if a.sparse < 1.0:
        sel &= rng_sp.random(a.rows) < a.sparse
    any_kept |= sel
    ks.append(lo[sel]); cs.append(host[sel, s]); offs[s + 1] = offs[s] + int(sel.sum())
    if hi is not None:
        hs.append(hi[sel])
kmers = np.concatenate(ks); counts = np.concatenate(cs).astype(np.uint32)
n = len(kmers)
dk, dc = K.DeviceBuffer.from_host(kmers), K.DeviceBuffer.from_host(counts)
dh = K.DeviceBuffer.from_host(np.concatenate(hs)) if hi is not None else None
LAY = {"tiled": K.LAYOUT_TILED, "rows": K.LAYOUT_ROWS, "soa": K.LAYOUT_SOA}[a.layout]
out = K.CountMatrix(a.rows, S, 4, LAY, with_kmers=True, kmer_limbs=a.limbs)
nr = C.c_uint64(0)
ts = []
for _ in range(a.iters + 1):
    lib.kmd_stream_sync(None)
    t0 = time.perf_counter()
    K._native.check(lib.kmd_merge_partition(S, dk.ptr, dh.ptr if dh else None, dc.ptr, offs.ctypes.data, 4, LAY, out.ld, a.rows,
                                            out.counts.ptr, out.kmer_lo.ptr, out.kmer_hi.ptr if dh else None, C.byref(nr), None))
    ts.append(time.perf_counter() - t0)
t = min(ts[1:])
assert nr.value == int(any_kept.sum()) or os.environ.get("KMD_NO_CHECK")
out.n_rows = a.rows
assert os.environ.get("KMD_NO_CHECK") or a.sparse < 1.0 or (out.to_host()[:1000] == host[:1000]).all()
inb = n * (12 if a.limbs == 1 else 20)
print("merge limbs=%d keys=%s S=%d rows=%d records=%d  %.2f ms  %.3e records/s  %.3e rows/s  input %.1f GB/s (%d B/record)"
      % (a.limbs, a.keys, S, a.rows, n, t * 1e3, n / t, a.rows / t, inb / t / 1e9, 12 if a.limbs == 1 else 20))
