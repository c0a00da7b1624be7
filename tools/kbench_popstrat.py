#!/usr/bin/env python3
"""Stage-2 micro-benchmark (config 5 shape): survivors/s of kmd_popstrat_apply."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmdiff_amd as K

ap = argparse.ArgumentParser()
ap.add_argument("--nc", type=int, default=100)
ap.add_argument("--nk", type=int, default=100)
ap.add_argument("--rows", type=int, default=4_000_000)
ap.add_argument("--npc", type=int, default=2)
ap.add_argument("--thr", type=float, default=1e-3)
a = ap.parse_args()
S = a.nc + a.nk
rng = np.random.default_rng(5)
Z = rng.normal(0, 0.1, size=(S, 10))
mat = K.synth_matrix(0x6B6D64696666, 0, a.rows, a.nc, a.nk, 4, K.LAYOUT_TILED)
tot = K.column_sums(mat)
model = K.PoissonLikelihood(a.nc, a.nk, tot[:a.nc], tot[a.nc:], 10000)
acc = K.SurvivorAccumulator(a.rows // 10)
K.diff_observer(model, acc, a.thr).process(mat)
ns = acc.finish()
pop = K.pop_strat_corrector(a.nc, a.nk, tot[:a.nc], tot[a.nc:], a.npc, Z)
counts = K.gather_counts(mat, acc.bufs["row"], ns)
pop.apply(counts, ns)
e0, e1 = K.Event(), K.Event()
e0.record()
for _ in range(3):
    p = pop.apply(counts, ns)
e1.record()
ms = e0.elapsed_ms(e1) / 3
print("popstrat S=%d F=%d survivors=%d  %.3f ms  %.3e survivors/s  (p<1e-6: %d)" %
      (S, pop.n_features, ns, ms, ns / (ms * 1e-3), int((p < 1e-6).sum())))
