#!/usr/bin/env python3
"""dev: per-call times of kmd_merge_filter on one configs[2] partition, back to back (what bench.py's pipeline leg averages)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmdiff_amd as K
lib = K._native.lib()
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 39062500
ss, tot = K.synth_streams(0x6B6D64696666, 0, rows, 20, 20)
model = K.PoissonLikelihood(20, 20, tot[:20], tot[20:], 10000)
acc = K.SurvivorAccumulator(max(1 << 16, rows // 100))
obs = K.diff_observer(model, acc, 5e-7)
K.merge_filter(ss, obs)
acc.counters.zero()
for rep in range(3):
    ts = []
    e = [K.Event() for _ in range(13)]
    e[0].record()
    t0 = time.perf_counter()
    for i in range(12):
        K.merge_filter(ss, obs)
        e[i + 1].record()
        ts.append(time.perf_counter())
    lib.kmd_stream_sync(None)
    print("events ms:", " ".join("%.3f" % e[i].elapsed_ms(e[i + 1]) for i in range(12)), "| host ms:", " ".join("%.3f" % ((b - a) * 1e3) for a, b in zip([t0] + ts[:-1], ts)))
