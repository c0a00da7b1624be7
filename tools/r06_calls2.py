#!/usr/bin/env python3
"""dev: as tools/r06_calls.py, in bench.py's setting -- four partitions' streams resident, a page-locked copy behind it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmdiff_amd as K
lib = K._native.lib()
rows = 39062500
sets = []
tot = None
for p in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    ss_p, tot_p = K.synth_streams(0x6B6D64696666, p, rows, 20, 20)
    sets.append(ss_p); tot = tot_p if tot is None else tot + tot_p
ss = sets[0]
model = K.PoissonLikelihood(20, 20, tot[:20], tot[20:], 10000)
acc = K.SurvivorAccumulator(max(1 << 16, rows // 100))
obs = K.diff_observer(model, acc, 5e-7)
for rep in range(3):
    acc.counters.zero()
    e = [K.Event() for _ in range(17)]
    e[0].record()
    for i in range(16):
        K.merge_filter(ss, obs)
        e[i + 1].record()
    lib.kmd_stream_sync(None)
    print("events ms:", " ".join("%.3f" % e[i].elapsed_ms(e[i + 1]) for i in range(16)), flush=True)
    time.sleep(0.5)
