#!/usr/bin/env python3
"""Streams -> survivors on one partition: K2 (k-way merge of the per-sample streams into the tiled
matrix) followed by K1 (Poisson test + threshold + compaction), both on data resident in HBM."""
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
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--keys", default="random")
ap.add_argument("--sparse", type=float, default=1.0, help="keep each record with this probability (rows of few records)")
ap.add_argument("--fused-only", action="store_true", help="only kmd_merge_filter (no matrix path beside it, no cross-check)")
ap.add_argument("--limbs", type=int, default=1, help="2: two-limb k-mers (32 < k <= 64); fused path only")
ap.add_argument("--overlap", type=int, default=0, help="also: this many host threads, each with a stream of its own, running the fused call on the same partition at once")
ap.add_argument("--triples", action="store_true", help="also time kmd_merge_sums + kmd_poisson_filter_sums")
ap.add_argument("--device", action="store_true", help="streams built on the device (kmd_synth_streams): whole configs[2] partitions (--rows 39062500) without a host copy; fused path only")
ap.add_argument("--partition", type=int, default=0)
ap.add_argument("--profile", type=int, default=0, help="--device: the rows' presence profile (1: mixed -- every second row in one or two samples, the others in 95 %%)")
a = ap.parse_args()
S = a.nc + a.nk
lib = K._native.lib()
if a.device:
    ss, tot = K.synth_streams(0x6B6D64696666, a.partition, a.rows, a.nc, a.nk, kmer_limbs=a.limbs, profile=a.profile)
    model = K.PoissonLikelihood(a.nc, a.nk, tot[:a.nc], tot[a.nc:], 10000)
    acc = K.SurvivorAccumulator(max(1 << 16, a.rows // 100), kmer_limbs=a.limbs)
    obs = K.diff_observer(model, acc, 5e-7)
    best_f = 1e9
    for _ in range(a.iters + 1):
        acc.counters.zero()
        lib.kmd_stream_sync(None)
        t0 = time.perf_counter()
        rows_f = K.merge_filter(ss, obs)
        lib.kmd_stream_sync(None)
        best_f = min(best_f, time.perf_counter() - t0)
    cf = acc.read_counters()
    assert rows_f == int(cf[0]) == a.rows
    # the pass that gives the survivors' p-values the reference's last bit (kmd_pvalues_refine), on this partition's sink
    e0, e1 = K.Event(), K.Event()
    ns, best_r = int(cf[1]), 1e9
    for _ in range(4):
        e0.record()
        K._native.check(lib.kmd_pvalues_refine(model.handle, ns, acc.bufs["mean_control"].ptr, acc.bufs["mean_case"].ptr, acc.bufs["pvalue"].ptr, None))
        e1.record()
        lib.kmd_stream_sync(None)
        best_r = min(best_r, e0.elapsed_ms(e1))
    print("kmd_pvalues_refine on the %d survivors: %.3f ms" % (ns, best_r))
    bpr = 12 if a.limbs == 1 else 20
    print("pipeline device-built S=%d records=%d rows=%d  fused merge+test (kmd_merge_filter) %.3f ms  %.3e rows/s  %.3e records/s  %.0f GB/s of %d B/record  sig=%d  candidates=%d"
          % (S, ss.total, rows_f, best_f * 1e3, rows_f / best_f, ss.total / best_f, bpr * 1e-9 * ss.total / best_f, bpr, int(cf[1]), int(cf[4])))
    sys.exit(0)
mat = K.synth_matrix(0x6B6D64696666, 0, a.rows, a.nc, a.nk, 4, K.LAYOUT_ROWS)
host = mat.to_host()
lo = mat.kmers_to_host()[0]
if a.keys == "random":
    lo = np.unique(np.random.default_rng(5).integers(0, 1 << 62, int(a.rows * 1.02), dtype=np.uint64))[:a.rows]
elif a.keys == "clustered":   # 2000 dense clusters scattered over the range (as tools/kbench_merge.py)
    starts = np.sort(np.random.default_rng(6).integers(0, 1 << 61, 2000, dtype=np.uint64))
    per = a.rows // 2000 + 1
    lo = np.unique((starts[:, None] + np.arange(per, dtype=np.uint64)[None, :] * np.uint64(3)).ravel())[:a.rows]
    assert len(lo) == a.rows
else:
    assert a.keys == "even", a.keys
offs = np.zeros(S + 1, dtype=np.uint64)
ks, cs = [], []
rng_sp = np.random.default_rng(11)
any_kept = np.zeros(a.rows, dtype=bool)
for s in range(S):
    sel = host[:, s] > 0
    if a.sparse < 1.0:
        sel &= rng_sp.random(a.rows) < a.sparse
        host[~sel, s] = 0
    any_kept |= sel
    ks.append(lo[sel]); cs.append(host[sel, s]); offs[s + 1] = offs[s] + int(sel.sum())
kmers = np.concatenate(ks); counts = np.concatenate(cs).astype(np.uint32)
n = len(kmers)
dk, dc = K.DeviceBuffer.from_host(kmers), K.DeviceBuffer.from_host(counts)
tot = host.sum(axis=0, dtype=np.uint64)
model = K.PoissonLikelihood(a.nc, a.nk, tot[:a.nc], tot[a.nc:], 10000)
acc = K.SurvivorAccumulator(max(1 << 16, a.rows // 100))
obs = K.diff_observer(model, acc, 5e-7)
nr = C.c_uint64(0)
best, sig_matrix, rows_matrix = 1e9, None, None
if not a.fused_only:
    out = K.CountMatrix(a.rows, S, 4, K.LAYOUT_TILED, with_kmers=True)
    for _ in range(a.iters + 1):
        acc.counters.zero()
        lib.kmd_stream_sync(None)
        t0 = time.perf_counter()
        K._native.check(lib.kmd_merge_partition(S, dk.ptr, None, dc.ptr, offs.ctypes.data, 4, K.LAYOUT_TILED, out.ld, a.rows,
                                                out.counts.ptr, out.kmer_lo.ptr, None, C.byref(nr), None))
        out.n_rows = int(nr.value)
        obs.process(out)
        lib.kmd_stream_sync(None)
        best = min(best, time.perf_counter() - t0)
    sig_matrix, rows_matrix = int(acc.read_counters()[1]), out.n_rows
# the same partition without the matrix or any rows in HBM: merge + test in one kernel (kmd_merge_filter)
if a.limbs == 2:
    assert a.fused_only, "--limbs 2 goes with --fused-only"
    # a k = 63 k-mer: the high limb (31 bases) orders the rows, the low limb (32 bases) is 64 bits of its own
    def low_limb(k_):
        z = (k_ + np.uint64(0x9E3779B97F4A7C15)) * np.uint64(0xBF58476D1CE4E5B9)
        return z ^ (z >> np.uint64(31))
    ss = K.StreamSet([(low_limb(k_), c_, k_) for k_, c_ in zip(ks, cs)])
    acc = K.SurvivorAccumulator(1 << 20, kmer_limbs=2)
    obs = K.diff_observer(model, acc, 5e-7)
else:
    ss = K.StreamSet([(k_, c_) for k_, c_ in zip(ks, cs)])
best_f, ev0, ev1 = 1e9, K.Event(), K.Event()
for _ in range(a.iters + 1):
    acc.counters.zero()
    lib.kmd_stream_sync(None)
    t0 = time.perf_counter()
    rows_f = K.merge_filter(ss, obs)
    lib.kmd_stream_sync(None)
    best_f = min(best_f, time.perf_counter() - t0)
cf = acc.read_counters()
sig_fused = int(cf[1])
assert rows_f == int(cf[0]) and (a.fused_only or (rows_f == rows_matrix and sig_fused == sig_matrix)), (rows_f, rows_matrix, int(cf[0]), sig_fused, sig_matrix)
bpr = 12 if a.limbs == 1 else 20
print("pipeline keys=%s S=%d records=%d rows=%d  fused merge+test (kmd_merge_filter) %.3f ms  %.3e rows/s  %.3e records/s  %.0f GB/s of %d B/record  sig=%d  candidates=%d"
      % (a.keys, S, n, rows_f, best_f * 1e3, rows_f / best_f, n / best_f, bpr * 1e-9 * n / best_f, bpr, sig_fused, int(cf[4])))
if a.overlap > 1:
    # partitions in flight on several streams: the boundary searches and the candidate evaluation of one call
    # run beside the merge kernel of another
    import threading
    T, per = a.overlap, max(4, a.iters)
    workers = []
    for t in range(T):
        st = C.c_void_p()
        K._native.check(lib.kmd_stream_create(C.byref(st)), "stream")
        acc_t = K.SurvivorAccumulator(1 << 20)
        workers.append((st, acc_t, K.diff_observer(model, acc_t, 5e-7)))
    def work(w, k, all_done):
        for _ in range(k):
            K.merge_filter(ss, w[2], stream=w[0])
        all_done.wait()                                    # (the threads leave together: DESIGN 10)
    def run_all(k):
        all_done = threading.Barrier(len(workers))
        th = [threading.Thread(target=work, args=(w, k, all_done)) for w in workers]
        for x in th: x.start()
        for x in th: x.join()
        for w in workers: lib.kmd_stream_sync(w[0])
    run_all(2)                                                    # warm-up: lists, the scratch of T concurrent calls
    t0 = time.perf_counter()
    run_all(per)
    dt = (time.perf_counter() - t0) / (T * per)
    for w in workers:
        cw = w[1].read_counters()
        assert int(cw[1]) == sig_fused * (per + 2) and int(cw[0]) == rows_f * (per + 2), (int(cw[0]), int(cw[1]))
        lib.kmd_stream_destroy(w[0])
    print("pipeline keys=%s S=%d records=%d rows=%d  fused, %d partitions in flight (streams, host threads) %.3f ms per partition  %.3e rows/s  %.3e records/s  %.0f GB/s of 12 B/record"
          % (a.keys, S, n, rows_f, T, dt * 1e3, rows_f / dt, n / dt, 12e-9 * n / dt))
if a.triples:
    # rows as (k-mer, control sum, case sum), then the test on the sums
    sums = K.RowSums(a.rows)
    best_s = 1e9
    for _ in range(a.iters + 1):
        acc.counters.zero()
        lib.kmd_stream_sync(None)
        t0 = time.perf_counter()
        K._native.check(lib.kmd_merge_sums(S, a.nc, dk.ptr, None, dc.ptr, offs.ctypes.data, sums.capacity, sums.kmers.ptr, None, sums.sum_c.ptr, sums.sum_k.ptr,
                                           C.byref(nr), None))
        sums.n_rows = int(nr.value)
        obs.process_sums(sums)
        lib.kmd_stream_sync(None)
        best_s = min(best_s, time.perf_counter() - t0)
    sig_sums = int(acc.read_counters()[1])
    assert int(acc.read_counters()[0]) == rows_f and sig_sums == sig_fused, (int(acc.read_counters()[0]), rows_f, sig_sums, sig_fused)
    print("pipeline keys=%s S=%d records=%d rows=%d  rows as sums + test on the sums %.2f ms  %.3e rows/s  %.3e records/s  sig=%d"
          % (a.keys, S, n, rows_f, best_s * 1e3, rows_f / best_s, n / best_s, sig_sums))
if not a.fused_only:
  print("pipeline keys=%s S=%d records=%d rows=%d  merge+filter %.2f ms  %.3e rows/s  %.3e records/s  sig=%d"
        % (a.keys, S, n, out.n_rows, best * 1e3, out.n_rows / best, n / best, sig_matrix))
