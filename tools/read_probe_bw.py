import sys, os
sys.path.insert(0, os.getcwd())
import kmdiff_amd as K
lib = K._native.lib()
nb = 6 << 30
bufs = [K.DeviceBuffer(nb) for _ in range(4)]
for b in bufs: lib.kmd_memset(b.ptr, 1, nb, None)
sink = K.DeviceBuffer(8).zero()
for w in (8, 16, 72, 80):
    for nbuf in (1, 4):
        for _ in range(2): lib.kmd_read_probe(bufs[0].ptr, nb, w, sink.ptr, None)
        e0, e1 = K.Event(), K.Event()
        e0.record()
        for i in range(8): K._native.check(lib.kmd_read_probe(bufs[i % nbuf].ptr, nb, w, sink.ptr, None))
        e1.record()
        ms = e0.elapsed_ms(e1) / 8
        print("read probe width=%d%s buffers=%d  %.3f ms  %.0f GB/s" % (w & 63, " nt" if w & 64 else "", nbuf, ms, nb / ms / 1e6))
