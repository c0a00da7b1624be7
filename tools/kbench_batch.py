#!/usr/bin/env python3
"""kmd_merge_filter_batch: P copies of one 20v20 partition's streams (distinct device buffers) through the batch entry
point, against P single calls."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmdiff_amd as K
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=4_000_000)
ap.add_argument("--parts", type=int, default=12)
ap.add_argument("--iters", type=int, default=4)
ap.add_argument("--device", action="store_true", help="streams built on the device (kmd_synth_streams), four distinct partitions taken in turn: whole configs[2] partitions with --rows 39062500")
a = ap.parse_args()
lib = K._native.lib()
NC = NK = 20
if a.device:
    made = [K.synth_streams(0x6B6D64696666, p, a.rows, NC, NK) for p in range(min(4, a.parts))]
    tot = sum(t for _, t in made)
    sets = [made[i % len(made)][0] for i in range(a.parts)]
else:
    mat = K.synth_matrix(0x6B6D64696666, 0, a.rows, NC, NK, 4, K.LAYOUT_ROWS)
    host, lo = mat.to_host(), mat.kmers_to_host()[0]
    del mat
    streams = [(lo[host[:, s] > 0], host[host[:, s] > 0, s]) for s in range(NC + NK)]
    tot = host.sum(axis=0, dtype=np.uint64)
    sets = [K.StreamSet(streams) for _ in range(a.parts)]
model = K.PoissonLikelihood(NC, NK, tot[:NC], tot[NC:], 10000)
accs = [K.SurvivorAccumulator(max(1 << 18, a.rows // 100)) for _ in range(a.parts)]
obs = [K.diff_observer(model, acc, 5e-7) for acc in accs]
for o, ss in zip(obs, sets):
    K.merge_filter(ss, o)
lib.kmd_stream_sync(None)
best_s = best_b = 1e9
for _ in range(a.iters):
    t0 = time.perf_counter()
    for o, ss in zip(obs, sets):
        K.merge_filter(ss, o)
    lib.kmd_stream_sync(None)
    best_s = min(best_s, (time.perf_counter() - t0) / a.parts)
K.merge_filter_batch(sets, obs)
for _ in range(a.iters):
    t0 = time.perf_counter()
    rows = K.merge_filter_batch(sets, obs)
    best_b = min(best_b, (time.perf_counter() - t0) / a.parts)
assert rows == [a.rows] * a.parts
n = sets[0].total
print("batch S=40 records=%d rows=%d parts=%d  single calls %.3f ms per partition (%.0f GB/s)  kmd_merge_filter_batch %.3f ms per partition (%.3e rows/s, %.0f GB/s of 12 B/record)"
      % (n, a.rows, a.parts, best_s * 1e3, 12e-9 * n / best_s, best_b * 1e3, a.rows / best_b, 12e-9 * n / best_b))
