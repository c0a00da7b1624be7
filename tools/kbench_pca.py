#!/usr/bin/env python3
"""K5 micro-benchmark: PCA front end on one synthetic partition (sampling pass, Gram matrix, Jacobi)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmdiff_amd as K

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=39_062_500)
ap.add_argument("--nc", type=int, default=20)
ap.add_argument("--nk", type=int, default=20)
ap.add_argument("--rate", type=float, default=0.001)
a = ap.parse_args()
S = a.nc + a.nk
lib = K._native.lib()
mat = K.synth_matrix(0x6B6D64696666, 0, a.rows, a.nc, a.nk, 4, K.LAYOUT_TILED)


def timed(f, n=3):
    best = 1e9
    for _ in range(n):
        lib.kmd_stream_sync(None)
        t0 = time.perf_counter()
        r = f()
        lib.kmd_stream_sync(None)
        best = min(best, time.perf_counter() - t0)
    return best, r


def sample_once():
    p = K.PopulationPCA(S, a.rate, seed=1, capacity=max(1 << 16, int(a.rows * a.rate * 2)))
    p.sample(mat)
    return p


ts, pca = timed(sample_once)
n = pca.count()
tg, xtx = timed(pca.gram)
te, (evec, evals) = timed(lambda: K.pca_eigen(xtx, min(10, S)))
print("pca S=%d rows=%d rate=%g sampled=%d  sample %.3f ms (%.2e rows/s)  gram %.3f ms  eigen %.3f ms  lambda1=%.4f"
      % (S, a.rows, a.rate, n, ts * 1e3, a.rows / ts, tg * 1e3, te * 1e3, evals[0]))
