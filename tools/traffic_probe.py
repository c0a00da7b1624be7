#!/usr/bin/env python3
"""Run under `rocprofv3 --pmc FETCH_SIZE` (and again with WRITE_SIZE): launches the read
probes of a known byte count (4/8/16-byte loads) and then the filter kernel on one C3
partition, so that the PMC bytes of the filter kernel can be corrected with the calibration
factor of its own load width (MI355X_MICROARCH.md, HBM section)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmdiff_amd as K

lib = K._native.lib()
ROWS = 39_062_500
nbytes = 4 << 30
buf = K.DeviceBuffer(nbytes)
lib.kmd_memset(buf.ptr, 1, nbytes, None)
sink = K.DeviceBuffer(8).zero()
for w in (4, 8, 16, 64 + 8, 64 + 16):          # + 64: non-temporal (what k_filter_soa issues)
    for _ in range(2):
        K._native.check(lib.kmd_read_probe(buf.ptr, nbytes, w, sink.ptr, None))
lib.kmd_stream_sync(None)
buf.free()
LAYOUT = {"tiled": K.LAYOUT_TILED, "soa": K.LAYOUT_SOA, "rows": K.LAYOUT_ROWS}[os.environ.get("KMD_LAYOUT", "tiled")]
mat = K.synth_matrix(0x6B6D64696666, 0, ROWS, 20, 20, 4, LAYOUT)
tot = K.column_sums(mat)
model = K.PoissonLikelihood(20, 20, tot[:20], tot[20:], 10000)
acc = K.SurvivorAccumulator(1 << 20)
obs = K.diff_observer(model, acc, 5e-7)
for _ in range(3):
    obs.process(mat)
print("counters", acc.read_counters()[:6], "probe_bytes", nbytes, "rows", ROWS)
