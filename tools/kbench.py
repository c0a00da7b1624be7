#!/usr/bin/env python3
"""Kernel micro-benchmark: time kmd_poisson_filter on one resident synthetic partition.
KMD_LIB=<path to a variant .so> selects the build under test (tools/sweep.sh)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmdiff_amd as K


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=39_062_500)
    ap.add_argument("--nc", type=int, default=20)
    ap.add_argument("--nk", type=int, default=20)
    ap.add_argument("--count-bytes", type=int, default=4)
    ap.add_argument("--layout", default="soa")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--thr", type=float, default=5e-7)
    ap.add_argument("--tag", default="")
    ap.add_argument("--tile-rows", type=int, default=0, help="T of the tiled layout (default 4096)")
    a = ap.parse_args()
    if a.tile_rows:
        import kmdiff_amd.hip as H
        H.TILED_BLOCK_ROWS = a.tile_rows
    layout = {"soa": K.LAYOUT_SOA, "rows": K.LAYOUT_ROWS, "tiled": K.LAYOUT_TILED}[a.layout]
    mat = K.synth_matrix(0x6B6D64696666, 0, a.rows, a.nc, a.nk, a.count_bytes, layout)
    tot = K.column_sums(mat)
    model = K.PoissonLikelihood(a.nc, a.nk, tot[:a.nc], tot[a.nc:], 10000)
    acc = K.SurvivorAccumulator(max(1 << 20, a.rows // 100))
    obs = K.diff_observer(model, acc, a.thr)
    for _ in range(2):
        obs.process(mat)
    acc.read_counters()
    acc.counters.zero()
    ms = []
    for _ in range(3):           # back-to-back launches inside one event pair: no host gaps
        e0, e1 = K.Event(), K.Event()
        e0.record()
        for _ in range(a.iters):
            obs.process(mat)
        e1.record()
        ms.append(e0.elapsed_ms(e1) / a.iters)
    c = acc.read_counters()
    if os.environ.get("KMD_TIMING"):
        t = acc.counters.to_host(np.uint64, 16)
        print("timing: load %d cyc  math %d cyc  tiles %d  kernel %d cyc  wall %d ticks(100MHz) -> sclk %.0f MHz, per tile load %.0f math %.0f"
              % (t[8], t[9], t[10], t[11], t[12], t[11] / max(t[12], 1) * 100, t[8] / max(t[10], 1), t[9] / max(t[10], 1)))
    ms = np.array(ms)
    bpr = 8 + (a.nc + a.nk) * a.count_bytes
    best, med = ms.min(), np.median(ms)
    print("%-28s rows=%d S=%d cb=%d %s  median %.3f ms  best %.3f ms  %.0f GB/s (alg %dB/row)  %.3e rows/s  cand=%d sig=%d"
          % (a.tag or os.environ.get("KMD_LIB", "default")[-28:], a.rows, a.nc + a.nk, a.count_bytes, a.layout, med, best,
             a.rows * bpr / (med * 1e-3) / 1e9, bpr, a.rows / (med * 1e-3), int(c[4]) // (3 * a.iters), int(c[1]) // (3 * a.iters)),
          flush=True)


if __name__ == "__main__":
    main()
