#!/usr/bin/env python3
"""bench.py -- throughput of the `kmdiff diff` hot path on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): the
256-partition synthetic count matrix, 20 controls v 20 cases, k = 31, 4-byte counts,
39 062 500 rows per partition (10^10 rows / 256).  One STEP = one partition through stage 1
(merge observer + Poisson LRT + threshold + survivor compaction, kmd_poisson_filter) with the
partition already resident in HBM (tiled SoA: counts[row/4096][sample][row%4096] + kmer[row];
--layout soa|rows select the plain column-major and the reference's row-major layouts).  After the K timed
steps the job's single exchange (counter all-reduce, and for BH/Holm the survivor p-value
all-gather) and the significance correction (stage 3) run inside the timed region too.

Sharding: partition p belongs to rank p % N (weak scaling: every rank does K steps).

The JSON line also carries
  roofline     : algorithmic bytes (168 B/row) / average kernel duration, measured with one HIP
                 event pair around the K back-to-back launches (duration / K), against the
                 8 TB/s HBM3E peak (`frac` = `frac_events`: this run's events); `traffic` = HBM bytes per launch from
                 the PMC passes in profiles/ (static), `traffic_frac` = traffic / this run's kernel time / peak;
                 `profiled` = the kernel's average duration in the committed rocprofv3 CSV of this command and
                 the fraction that follows from it (static; what a reader of profiles/ recomputes);
  cpu_baseline : the reference's own arithmetic (oracle/_ref, kind "reference") or the C
                 restatement (kind "port") on the host cores, on a bounded sample of the same
                 workload, tail function evaluated for every row as the reference does;
  pipeline     : (N = 1) the kmtricks-side input of the same path: one WHOLE configs[2] partition's per-sample k-mer
                 streams (39 062 500 rows, ~10^9 records, 12 GB; built on the device) resident in HBM -> survivors
                 (kmd_merge_filter: k-way merge fused with the test), 12 algorithmic bytes per record, HIP events around
                 back-to-back calls; `overlapped` = the same with six partitions in flight on streams (and host threads) of
                 their own, per partition; `batched` = twelve partitions (four distinct ones in turn) through
                 kmd_merge_filter_batch (one host thread, two or three in flight inside the library); `small` = the 4 M-row
                 partition earlier rounds quoted (single calls only); `sparse` = a partition of the MIXED presence profile (every
                 second row in one or two samples, the others in 95 % of them), single calls and the batch, its own roofline
                 block; `feed_inclusive` = the first partition again WITH the link: packed streams in page-locked memory ->
                 async copies -> kmd_unpack_streams -> kmd_merge_filter, double-buffered, beside the link's ceiling;
  h2d_inclusive: (N = 1) the MATRIX-feed case only (`matrices/` input): the headline step with the host-to-device copy
                 of the partition's count matrix (from page-locked memory) inside the timed loop, serial -- never `value`.
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x6B6D64696666
NC, NK, K_SIZE, COUNT_BYTES = 20, 20, 31, 4
N_PARTITIONS = 256
ROWS_PER_PARTITION = 39_062_500          # 10^10 / 256
THRESHOLD, CUTOFF, LOG_FACTORIAL = 0.05, 100000, 10000      # -s, -u, --log-factorial defaults
BYTES_PER_ROW = 8 * ((K_SIZE + 31) // 32) + (NC + NK) * COUNT_BYTES       # 168
HBM_PEAK_GBS = 8000.0                    # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def usable_cpus():
    """CPUs this process may really run on: the affinity mask, capped by the cgroup's CPU quota (a lease
    can show 256 CPUs in /proc and grant a fraction of them: threads beyond the grant only time-share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    note = "affinity mask %d" % n
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                q = max(1, int(float(quota) / period + 0.999))
                note += ", cgroup quota %d" % q
                n = min(n, q)
            break
        except Exception:
            continue
    return max(1, n), note


def cpu_baseline(rows_per_part, tc, tk):
    """Host-core baseline on a bounded sample (one partition per thread): first ONE thread alone
    (`per_thread`), then one thread per usable CPU."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as OL
    o = OL.load()
    cores, cores_note = usable_cpus()
    thr = THRESHOLD / CUTOFF
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libkmdiff_ref.so")
    if os.path.exists(ref_path):
        # the reference's LogFactorialTable + alglib::chisquarecdistribution around the
        # 20-line process() glue (oracle/ref_shim.cpp); one partition per thread
        R = C.CDLL(ref_path)
        R.kmdref_model_new.restype = C.c_void_p
        R.kmdref_model_new.argtypes = [C.c_size_t] * 3 + [C.c_uint64] * 2
        R.kmdref_model_process.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t] + [C.c_void_p] * 4
        model = R.kmdref_model_new(LOG_FACTORIAL, NC, NK, tc, tk)
        n = 1_000_000
        secs = [0.0] * cores
        nsig = [0] * cores

        def work(t):
            host, _, _ = o.synth_rows(SEED, t % N_PARTITIONS, 0, n, NC, NK, COUNT_BYTES)
            p = np.zeros(n); s = np.zeros(n, dtype=np.int32); mc = np.zeros(n); mk = np.zeros(n)
            t0 = time.perf_counter()
            R.kmdref_model_process(C.c_void_p(model), host.ctypes.data, n, p.ctypes.data, s.ctypes.data,
                                   mc.ctypes.data, mk.ctypes.data)
            nsig[t] = int((p <= thr).sum())
            secs[t] = time.perf_counter() - t0
        work(0)                                            # one thread alone
        per_thread = n / secs[0]
        th = [threading.Thread(target=work, args=(t,)) for t in range(cores)]
        [t.start() for t in th]
        [t.join() for t in th]
        return {"value": cores * n / max(secs), "unit": "k-mers/s", "cores": cores, "kind": "reference",
                "per_thread": per_thread, "cores_from": cores_note,
                "sample": "%d partitions x %d rows (20v20, u32), one per thread, row-major, reference "
                          "LogFactorialTable + alglib chisquarecdistribution for every row (oracle/_ref: -O2, no "
                          "-march); per_thread = one thread alone; %d survivors"
                          % (cores, n, sum(nsig))}
    n = 2_000_000
    c = OL.Counters()
    s = o.L.kmdo_bench_partitions(SEED, cores, n, NC, NK, COUNT_BYTES, tc, tk, LOG_FACTORIAL, thr, cores,
                                  C.byref(c))
    return {"value": cores * n / s, "unit": "k-mers/s", "cores": cores, "kind": "port", "cores_from": cores_note,
            "sample": "%d partitions x %d rows (20v20, u32), one per thread, row-major, C restatement "
                      "with the tail function for every row; %d survivors" % (cores, n, c.n_sig)}


def pipeline_leg(K, lib, rows, iters=6, n_distinct=1, n_batch=12, with_extras=True, profile=0, keep=None):
    """Streams -> survivors on one partition of `rows` rows: the per-sample (k-mer, count) streams kmtricks writes
    (records of sample s = the rows with a non-zero count in column s), built on the device (kmd_synth_streams),
    resident in HBM, through kmd_merge_filter.  12 algorithmic bytes per record (8-byte k-mer + 4-byte count, each
    read once).  rows = 39 062 500 is a whole configs[2] partition -- what the reference merges per task
    (merge.hpp:265-289)."""
    sets, tot = [], None
    for p in range(n_distinct):
        ss_p, tot_p = K.synth_streams(SEED, p, rows, NC, NK, profile=profile)
        sets.append(ss_p)
        tot = tot_p if tot is None else tot + tot_p
    ss = sets[0]
    model = K.PoissonLikelihood(NC, NK, tot[:NC], tot[NC:], LOG_FACTORIAL)
    if keep is not None:                                   # (the feed-inclusive leg sends this partition across the link again)
        keep["ss"], keep["model"] = ss, model
    cap = max(1 << 16, rows // 100)
    acc = K.SurvivorAccumulator(cap)
    obs = K.diff_observer(model, acc, THRESHOLD / CUTOFF, NC, NK)
    # warm-up: first-use costs, and the clocks -- the first calls behind an idle moment run 5-15 % slower than the steady
    # state a job of hundreds of partitions sees (twelve back-to-back calls on a configs[2] partition, HIP events:
    # 2.69 2.62 2.47 2.41 2.39 2.40 2.37 2.37 2.38 2.38 2.36 2.35 ms; tools/r06_calls2.py).  Round 5 timed the six calls
    # right behind ONE warm-up call: the ramp was in the average.  Six untimed calls now; every timed call's own time is
    # in the line (`ms_calls`), `ms` is their mean.
    n_warm = int(os.environ.get("KMD_BENCH_PIPE_WARM", "6"))
    for _ in range(max(1, n_warm)):
        n_rows = K.merge_filter(ss, obs)
    acc.counters.zero()
    ev = [K.Event() for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        n_rows = K.merge_filter(ss, obs)
        ev[i + 1].record()
    K._native.check(lib.kmd_stream_sync(None))
    ms_calls = [ev[i].elapsed_ms(ev[i + 1]) for i in range(iters)]
    ms = ev[0].elapsed_ms(ev[iters]) / iters
    c = acc.read_counters()
    assert int(c[0]) == iters * n_rows == iters * rows, (int(c[0]), n_rows, rows)
    n_sig0 = int(c[1]) // iters
    gbs = 12.0 * ss.total / (ms * 1e-3) / 1e9
    roof = lambda g: {"bound": "hbm", "achieved": g, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": g / HBM_PEAK_GBS}
    out = {"what": "one partition's per-sample k-mer streams resident in HBM -> survivors (kmd_merge_filter: k-way merge "
                   "fused with the Poisson test; tile plan + boundary search + merge kernel + candidate evaluation, host "
                   "round trip included)",
           "config": {"workload": "%s: 20v20, k=31, %d rows, %d records of 12 bytes (8-byte k-mer + 4-byte count) in 40 per-sample "
                                  "streams, built on the device (kmd_synth_streams)%s"
                                  % ("one configs[2] partition (10^10 rows / 256)" if rows == ROWS_PER_PARTITION else "a reduced partition", rows, ss.total,
                                     ("; MIXED presence profile: every second row in one or two samples, the others in 95 %% of the samples "
                                      "(%.1f records per row on average)" % (ss.total / float(rows))) if profile else ""),
                      "rows": rows, "records": ss.total, "samples": NC + NK},
           "records": ss.total, "rows": int(n_rows), "samples": NC + NK, "ms": ms, "ms_calls": ms_calls, "warmup_calls": max(1, n_warm), "kmers_per_s": n_rows / (ms * 1e-3),
           "records_per_s": ss.total / (ms * 1e-3), "bytes_algorithmic": 12 * ss.total, "roofline": roof(gbs), "n_sig": n_sig0}
    if with_extras == "batched":
        # (the sparse leg: single calls above, and the batch entry point -- no host threads, no refine timing)
        accs = [K.SurvivorAccumulator(cap) for _ in range(n_batch)]
        obs_b = [K.diff_observer(model, a_, THRESHOLD / CUTOFF, NC, NK) for a_ in accs]
        b_sets = [sets[i % n_distinct] for i in range(n_batch)]
        K.merge_filter_batch(b_sets, obs_b)
        K._native.check(lib.kmd_stream_sync(None))
        t0 = time.perf_counter()
        for _ in range(3):
            rows_b = K.merge_filter_batch(b_sets, obs_b)
        ms_b = (time.perf_counter() - t0) / (3 * n_batch) * 1e3
        assert rows_b == [rows] * n_batch
        rec_avg = sum(x.total for x in sets) / float(n_distinct)
        out["batched"] = {"partitions": n_batch, "in_flight": "2 or 3 (the library's choice)", "ms_per_partition": ms_b, "kmers_per_s": rows / (ms_b * 1e-3),
                          "records_per_s": rec_avg / (ms_b * 1e-3), "what": "kmd_merge_filter_batch, one host thread",
                          "roofline": roof(12.0 * rec_avg / (ms_b * 1e-3) / 1e9)}
        return out
    if not with_extras:
        return out
    # beside it, not inside: the optional pass that gives the survivors' p-values the reference's last bit
    # (kmd_pvalues_refine), on the sink one partition leaves
    r0, r1 = K.Event(), K.Event()
    best = float("inf")
    for _ in range(3):
        r0.record()
        K._native.check(lib.kmd_pvalues_refine(model.handle, n_sig0, acc.bufs["mean_control"].ptr, acc.bufs["mean_case"].ptr,
                                               acc.bufs["pvalue"].ptr, None), "pvalues_refine")
        r1.record()
        K._native.check(lib.kmd_stream_sync(None), "sync")
        best = min(best, r0.elapsed_ms(r1))
    out["refine_pvalues_ms"] = best
    # partitions in flight: a job has hundreds of partitions; with six of them on streams (and host threads) of
    # their own, the boundary searches, the candidate evaluation and the read-back of one run beside the merge
    # kernel of another
    in_flight, per = int(os.environ.get("KMD_BENCH_OVERLAP_THREADS", "6")), max(4, iters)      # (dev: the host threads of the overlapped leg)
    workers = []
    for w_i in range(in_flight):
        st = C.c_void_p()
        K._native.check(lib.kmd_stream_create(C.byref(st)), "kmd_stream_create")
        acc_t = K.SurvivorAccumulator(cap)
        workers.append((st, acc_t, K.diff_observer(model, acc_t, THRESHOLD / CUTOFF, NC, NK), sets[w_i % n_distinct]))

    def work(w, k):
        for _ in range(k):
            K.merge_filter(w[3], w[2], stream=w[0])
    def run_all(k):
        threads = [threading.Thread(target=work, args=(w, k)) for w in workers]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for w in workers:
            K._native.check(lib.kmd_stream_sync(w[0]))
    run_all(2)                                             # untimed: the scratch of six concurrent calls gets allocated here
    n_timed, rounds_o = 3, []                              # the MEDIAN of three timed rounds (six host threads: the one measurement here the host's scheduler has a say in)
    for _ in range(n_timed):
        t0 = time.perf_counter()
        run_all(per)
        rounds_o.append((time.perf_counter() - t0) / (in_flight * per) * 1e3)
    ms_o = sorted(rounds_o)[n_timed // 2]
    for w in workers:
        cw = w[1].read_counters()
        assert int(cw[0]) == (n_timed * per + 2) * rows, int(cw[0])
        lib.kmd_stream_destroy(w[0])
    rec_avg = sum(x.total for x in sets) / float(n_distinct)
    gbs_o = 12.0 * rec_avg / (ms_o * 1e-3) / 1e9
    out["overlapped"] = {"partitions_in_flight": in_flight, "ms_per_partition": ms_o, "ms_rounds": rounds_o, "kmers_per_s": rows / (ms_o * 1e-3),
                         "records_per_s": rec_avg / (ms_o * 1e-3), "roofline": roof(gbs_o)}
    # the same through ONE host thread: kmd_merge_filter_batch keeps two or three partitions in flight on streams of the
    # library's own and waits once per partition (n_distinct different partitions in HBM, taken in turn)
    b_sets = [sets[i % n_distinct] for i in range(n_batch)]
    accs = [K.SurvivorAccumulator(cap) for _ in range(n_batch)]
    obs_b = [K.diff_observer(model, a_, THRESHOLD / CUTOFF, NC, NK) for a_ in accs]
    K.merge_filter_batch(b_sets, obs_b)                    # untimed: scratch of six concurrent partitions, streams
    for a_ in accs:
        a_.counters.zero()
    K._native.check(lib.kmd_stream_sync(None))
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        rows_b = K.merge_filter_batch(b_sets, obs_b)
    ms_b = (time.perf_counter() - t0) / (reps * n_batch) * 1e3
    assert rows_b == [rows] * n_batch
    for a_ in accs:
        cb = a_.read_counters()
        assert int(cb[0]) == reps * rows, int(cb[0])
    assert int(accs[0].read_counters()[1]) == reps * n_sig0
    gbs_b = 12.0 * rec_avg / (ms_b * 1e-3) / 1e9
    out["batched"] = {"partitions": n_batch, "distinct_partitions": n_distinct, "in_flight": "2 or 3 (the library's choice)", "ms_per_partition": ms_b,
                      "kmers_per_s": rows / (ms_b * 1e-3), "records_per_s": rec_avg / (ms_b * 1e-3),
                      "what": "kmd_merge_filter_batch, one host thread", "roofline": roof(gbs_b)}
    return out


def feed_leg(K, lib, ss, model, n_parts=6, threads=8):
    """The path `kmdiff-hip diff` takes, link included: one whole partition's streams as the host hands them over -- packed
    per 256 records (kmd_pack_stream: what the command's decoder threads write into page-locked memory), kmd_memcpy_h2d_async
    on a copy stream, kmd_unpack_streams on a second stream behind an event of its copy, kmd_merge_filter on the unpacked
    arrays: the copies of partitions i + 1 and i + 2 run beside the kernels of partition i.  Never `value`."""
    from concurrent.futures import ThreadPoolExecutor
    S, offs = ss.n_samples, ss.offs
    bound = int(lib.kmd_pack_block_bound())
    t_pack0 = time.perf_counter()

    def pack(s_):
        a_, b_ = int(offs[s_]), int(offs[s_ + 1])
        n_ = b_ - a_
        nb_ = (n_ + 255) // 256
        if n_ == 0:
            return np.zeros(0, np.uint8), np.zeros(0, np.uint32)
        km = ss.kmers.to_host(np.uint64, n_, offset_bytes=a_ * 8)
        ct = ss.counts.to_host(np.uint32, n_, offset_bytes=a_ * 4)
        out_ = np.empty(nb_ * bound, dtype=np.uint8)
        tab_ = np.empty(nb_, dtype=np.uint32)
        got = int(lib.kmd_pack_stream(km.ctypes.data, ct.ctypes.data, n_, out_.ctypes.data, out_.nbytes, tab_.ctypes.data))
        assert got > 0 and got % 8 == 0
        return out_[:got].copy(), tab_
    with ThreadPoolExecutor(max_workers=threads) as pool:
        parts = list(pool.map(pack, range(S)))
    t_pack = time.perf_counter() - t_pack0
    base = np.zeros(S + 1, dtype=np.uint64)
    for s_ in range(S):
        base[s_ + 1] = base[s_] + parts[s_][0].nbytes
    P = int(base[S])
    table = np.concatenate([t for _, t in parts]) if S else np.zeros(0, np.uint32)
    h = C.c_void_p()
    K._native.check(lib.kmd_malloc_host(C.byref(h), P + table.nbytes + 64), "kmd_malloc_host")
    st, st_u = C.c_void_p(), C.c_void_p()
    K._native.check(lib.kmd_stream_create(C.byref(st)), "kmd_stream_create")          # the copies
    K._native.check(lib.kmd_stream_create(C.byref(st_u)), "kmd_stream_create")        # kmd_unpack_streams, behind an event of the copy it reads
    try:
        hb = (C.c_uint8 * (P + table.nbytes)).from_address(h.value)
        hv = np.frombuffer(hb, dtype=np.uint8)
        for s_ in range(S):
            hv[int(base[s_]):int(base[s_ + 1])] = parts[s_][0]
        hv[P:P + table.nbytes] = table.view(np.uint8)
        del parts
        # three packed buffers (two copies queued behind each other: the link never waits for an unpack), two sets of
        # unpacked arrays (one being merged, one being written)
        d_packed = [K.DeviceBuffer(P + table.nbytes + 64) for _ in range(3)]
        copied = [K.Event() for _ in range(3)]
        outs = []
        for _ in range(2):
            o = K.StreamSet.__new__(K.StreamSet)
            o.n_samples, o.two, o.offs, o.total = S, False, offs, ss.total
            o.kmers, o.counts, o.kmers_hi = K.DeviceBuffer(ss.total * 8), K.DeviceBuffer(ss.total * 4), None
            outs.append(o)
        cap = max(1 << 16, int(offs[-1]) // 2000)
        acc = K.SurvivorAccumulator(cap)
        obs = K.diff_observer(model, acc, THRESHOLD / CUTOFF, NC, NK)
        # the link alone: the packed partition, page-locked source, one copy at a time
        K._native.check(lib.kmd_memcpy_h2d(d_packed[0].ptr, h, P + table.nbytes, st), "h2d")
        t0 = time.perf_counter()
        for _ in range(2):
            K._native.check(lib.kmd_memcpy_h2d(d_packed[0].ptr, h, P + table.nbytes, st), "h2d")
        link_gbs = 2 * (P + table.nbytes) / (time.perf_counter() - t0) / 1e9

        def enqueue_copy(k):
            K._native.check(lib.kmd_memcpy_h2d_async(d_packed[k % 3].ptr, h, P + table.nbytes, st), "h2d_async")
            copied[k % 3].record(st)

        def enqueue_unpack(k):
            K._native.check(lib.kmd_stream_wait_event(st_u, copied[k % 3].ptr), "kmd_stream_wait_event")
            K._native.check(lib.kmd_unpack_streams(S, d_packed[k % 3].ptr, base.ctypes.data, d_packed[k % 3].ptr + P, offs.ctypes.data,
                                                   outs[k % 2].kmers.ptr, outs[k % 2].counts.ptr, st_u), "kmd_unpack_streams")
        # untimed: one partition through (first-use costs, the survivors it must reproduce)
        enqueue_copy(0)
        enqueue_unpack(0)
        K._native.check(lib.kmd_stream_sync(st_u))
        rows0 = K.merge_filter(outs[0], obs)
        n_sig0 = int(acc.read_counters()[1])
        acc.counters.zero()
        enqueue_copy(0)
        enqueue_copy(1)
        enqueue_unpack(0)
        t0 = time.perf_counter()
        rows_seen = 0
        for i in range(n_parts):
            K._native.check(lib.kmd_stream_sync(st_u))      # partition i lies unpacked in HBM
            if i + 2 < n_parts:
                enqueue_copy(i + 2)                         # (its buffer held partition i - 1: unpacked long since)
            if i + 1 < n_parts:
                enqueue_unpack(i + 1)                       # behind its copy; into the arrays partition i - 1 was merged from
            rows_seen += K.merge_filter(outs[i % 2], obs)   # beside the copies of partitions i + 1, i + 2
        K._native.check(lib.kmd_stream_sync(None))
        dt = (time.perf_counter() - t0) / n_parts
        c = acc.read_counters()
        assert rows_seen == n_parts * rows0 and int(c[1]) == n_parts * n_sig0, (rows_seen, rows0, int(c[1]), n_sig0)
    finally:
        lib.kmd_stream_destroy(st)
        lib.kmd_stream_destroy(st_u)
        lib.kmd_free_host(h)
    bpr = (P + table.nbytes) / float(ss.total)
    ceiling = link_gbs * 1e9 / ((P + table.nbytes) / float(rows0))
    return {"what": "one whole configs[2] partition per step: packed streams (kmd_pack_stream) in page-locked memory -> kmd_memcpy_h2d_async "
                    "on a copy stream -> kmd_unpack_streams -> kmd_merge_filter, double-buffered (the copy of partition i + 1 beside the kernels "
                    "of partition i): what `kmdiff-hip diff` does per partition once its files are decoded; never `value`",
            "partitions": n_parts, "rows": int(rows0), "records": int(ss.total), "packed_bytes": P + table.nbytes, "bytes_per_record": bpr,
            "ms_per_partition": dt * 1e3, "kmers_per_s": rows0 / dt, "records_per_s": ss.total / dt, "link_GBs": link_gbs,
            "h2d_GBs_sustained": (P + table.nbytes) / dt / 1e9, "link_ceiling_kmers_per_s": ceiling, "frac_of_link_ceiling": (rows0 / dt) / ceiling,
            "n_sig": n_sig0, "host_pack_seconds": t_pack, "host_pack_threads": threads}


def h2d_leg(K, lib, obs, mat, steps=3):
    """The headline step with the partition's host-to-device copy (page-locked source) in the loop."""
    nbytes = mat.counts.nbytes
    h = C.c_void_p()
    K._native.check(lib.kmd_malloc_host(C.byref(h), nbytes), "kmd_malloc_host")
    try:
        K._native.check(lib.kmd_memcpy_d2h(h, mat.counts.ptr, nbytes, None), "d2h")
        K._native.check(lib.kmd_stream_sync(None))
        t0 = time.perf_counter()
        for _ in range(steps):
            K._native.check(lib.kmd_memcpy_h2d(mat.counts.ptr, h, nbytes, None), "h2d")
            obs.process(mat)
        K._native.check(lib.kmd_stream_sync(None))
        dt = (time.perf_counter() - t0) / steps
    finally:
        lib.kmd_free_host(h)
    return {"value": mat.n_rows / dt, "unit": "k-mers/s", "ms_per_step": dt * 1e3, "h2d_GBs": nbytes / dt / 1e9,
            "what": "kmd_memcpy_h2d of the partition's %d-byte count matrix from page-locked memory + kmd_poisson_filter, "
                    "serial, per step" % nbytes}


def e2e_leg(parts=int(os.environ.get("KMD_BENCH_E2E_PARTS", "8")), rows=int(os.environ.get("KMD_BENCH_E2E_ROWS", "2000000"))):
    """The command end to end on files (never `value`): tools/cli_throughput.py fabricates a kmtricks run directory
    (configs[2]'s sample split, `parts` partitions of `rows` rows), runs `kmdiff-hip diff -t <threads the quota gives>` on it
    and, on the SAME files, oracle/cpu_pipeline -- liblz4 decode + the oracle's S-way merge + the oracle's test, one task
    per partition on as many threads (global_merge::merge, merge.hpp:239-307): the like-for-like CPU baseline."""
    import subprocess
    import tempfile
    here = os.path.dirname(os.path.abspath(__file__))
    threads = usable_cpus()[0]
    with tempfile.TemporaryDirectory(prefix="kmd_e2e_") as tmp:
        js = os.path.join(tmp, "e2e.json")
        r = subprocess.run([sys.executable, os.path.join(here, "tools", "cli_throughput.py"), "--parts", str(parts), "--rows", str(rows), "--nc", str(NC),
                            "--nk", str(NK), "--cpu-baseline", "--only", "-t %d" % threads, "--json", js], capture_output=True, text=True)
        if r.returncode != 0:
            return {"error": (r.stderr or r.stdout)[-400:]}
        with open(js) as f:
            got = json.load(f)
    gpu, cpu = got["kmdiff_hip_diff"][0], got["cpu_baseline_e2e"]
    return {"workload": "kmdiff-hip diff on %d partitions x %d rows, %dv%d, k = 31, files in the page cache" % (parts, rows, NC, NK),
            "run_dir": got["run_dir"], "kmdiff_hip_diff": gpu,
            "cpu_baseline_e2e": {"value": cpu["rows_per_s"], "unit": "k-mers/s", "cores": cpu["threads"], "kind": "port", "seconds": cpu["seconds"],
                                 "sample": cpu["what"]},
            "gpu_over_cpu": gpu["rows_per_s"] / cpu["rows_per_s"]}


def launch_ranks(n):
    """Start `python -m torch.distributed.run --nproc-per-node n bench.py <same arguments>` as a child process and
    return its exit code (the caller has not touched the GPU runtime yet)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on these hosts (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=ROWS_PER_PARTITION, help="rows per partition")
    ap.add_argument("--resident", type=int, default=8, help="distinct partitions kept in HBM per rank")
    ap.add_argument("--correction", default="bonferroni")
    ap.add_argument("--layout", default="tiled", choices=["tiled", "soa", "rows"])
    ap.add_argument("--pipeline-rows", type=int, default=ROWS_PER_PARTITION, help="rows of the partition of the streams -> survivors leg")
    ap.add_argument("--pipeline-partitions", type=int, default=4, help="distinct partitions' streams kept in HBM for that leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the streams -> survivors and H2D-inclusive legs (N = 1 extras)")
    ap.add_argument("--transport", default="auto", choices=["auto", "torch", "rccl"],
                    help="N > 1: the wire of the job's one exchange (kmd_correct_sharded).  torch: torch.distributed's communicator on zero-copy views "
                         "of the library's buffers (kmdiff_amd/dist.py); rccl: libkmdiff_hip_rccl.so's own communicator (ncclCommInitRank; the id "
                         "travels through torch.distributed); auto: torch, and rccl if the warm-up exchange fails on any rank")
    ap.add_argument("--e2e", action="store_true", help="N = 1 extra (minutes; not in the default run): `kmdiff-hip diff` on a fabricated kmtricks run "
                    "directory -- files, LZ4 decode, PCIe and all -- with the CPU doing the same job on the same files beside it (oracle/cpu_pipeline)")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: start the N ranks ourselves -- as a CHILD process, before this
    # process has imported torch or touched HIP (an exec after a GPU call takes the box down; a child does not) --
    # forward its output and leave with its return code.  Under torchrun (WORLD_SIZE set) this is skipped.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (before the GPU runtime loads: RCCL's IPC on these hosts is dmabuf only)
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # dev switches (not used by the driver): KMD_BENCH_OVERSUBSCRIBE=1 folds the ranks onto the GPUs
    # there are, KMD_BENCH_BACKEND=gloo swaps RCCL out -- together they run the N>1 code on a 1-GPU box
    if os.environ.get("KMD_BENCH_OVERSUBSCRIBE") == "1":
        local_rank %= torch.cuda.device_count()
    backend = os.environ.get("KMD_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    if world > 1:
        # a collective that a peer never joins (it died, it hangs) fails after this long instead of after the backends'
        # 10 / 30 minutes: every rank then leaves with an error, and the launcher with a non-zero status
        import datetime
        patience = datetime.timedelta(seconds=float(os.environ.get("KMD_BENCH_COLLECTIVE_TIMEOUT_S", "300")))
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=patience)
        else:
            dist.init_process_group(backend=backend, timeout=patience)
    # how many ranks the collective library really connected: an all-reduce of 1 over the job's backend
    n_ranks_seen = world
    if world > 1:
        one = torch.ones(1, dtype=torch.int64, device=torch.device("cuda", local_rank) if backend == "nccl" else "cpu")
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        n_ranks_seen = int(one.item())

    # (the N > 1 line checks itself before it times anything: every rank the launcher started is in the communicator)
    if n_ranks_seen != world:
        raise SystemExit("bench.py: the collective library connected %d ranks, the job has %d" % (n_ranks_seen, world))

    import kmdiff_amd as K
    from kmdiff_amd import dist as D
    lib = K._native.lib()
    K._native.check(lib.kmd_set_device(local_rank))

    # ---- the wire of the exchange (N > 1) ----------------------------------------------------------------------
    transport_state = {"name": "torch" if args.transport == "auto" else args.transport, "obj": None, "fallback": None}

    def rccl_transport():
        """libkmdiff_hip_rccl.so's own communicator: rank 0 draws the id, torch.distributed carries it to the others"""
        R = K._native.rccl_lib()
        ident = (C.c_uint8 * 128)()
        if rank == 0:
            if R.kmd_rccl_unique_id(ident) != 0:
                raise RuntimeError("kmd_rccl_unique_id: %s" % R.kmd_rccl_last_error())
        box = [bytes(ident)]
        dist.broadcast_object_list(box, src=0)
        ident = (C.c_uint8 * 128)(*box[0])
        tr = K._native.Transport()
        if R.kmd_transport_rccl_init(C.byref(tr), world, rank, ident) != 0:
            raise RuntimeError("kmd_transport_rccl_init: %s" % R.kmd_rccl_last_error())
        return tr

    def all_ranks_ok(ok):
        """True on every rank iff `ok` on every rank (plain tensors of torch.distributed, not the library's buffers)"""
        if world == 1:
            return bool(ok)
        t = torch.tensor([0 if ok else 1], dtype=torch.int64, device=torch.device("cuda", local_rank) if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(t.item()) == 0

    def exchange(*a):
        return D.correct_sharded(K, *a, transport=transport_state["obj"])

    if world > 1 and transport_state["name"] == "rccl":
        if backend != "nccl":
            raise SystemExit("bench.py: --transport rccl needs one GPU per rank (backend nccl), not a folded run")
        transport_state["obj"] = rccl_transport()
    layout = {"tiled": K.LAYOUT_TILED, "soa": K.LAYOUT_SOA, "rows": K.LAYOUT_ROWS}[args.layout]
    thr = THRESHOLD / CUTOFF

    # ---- setup (untimed): resident partitions, totals, model, survivor sink ------------------
    n_res = max(1, min(args.resident, args.steps))
    parts = [(rank + i * world) % N_PARTITIONS for i in range(n_res)]
    mats = [K.synth_matrix(SEED, p, args.rows, NC, NK, COUNT_BYTES, layout) for p in parts]
    tot_buf = K.DeviceBuffer((NC + NK) * 8).zero()
    for m in mats:
        K.column_sums(m, tot_buf)
    K._native.check(lib.kmd_stream_sync(None))
    totals = D.allreduce_totals(tot_buf.to_host(np.uint64, NC + NK))
    model = K.PoissonLikelihood(NC, NK, totals[:NC], totals[NC:], LOG_FACTORIAL)
    cap = max(1 << 20, int(args.rows * (args.steps + args.warmup) * 4e-4))
    acc = K.SurvivorAccumulator(cap)
    obs = K.diff_observer(model, acc, thr, NC, NK)

    for i in range(args.warmup):
        obs.process(mats[i % n_res])
    if args.warmup:
        # warm the end-of-job path too (first-use cost of the sort / correction kernels)
        n_w = acc.finish(sort=True)
        # ... which is also the rehearsal of the wire: should the exchange fail on ANY rank (--transport auto, N > 1: the
        # zero-copy views of torch_transport have never carried two ranks over RCCL before the first multi-GPU run), every
        # rank switches to the library's own communicator and rehearses again -- a second failure ends the job
        try:
            exchange(args.correction, THRESHOLD, acc.read_counters(), acc.bufs["pvalue"], acc.bufs["sign"], n_w)
            ok, why = True, None
        except Exception as e:                                  # noqa: BLE001 (whatever it was, the ranks must agree on what happens next)
            ok, why = False, repr(e)
        if not all_ranks_ok(ok):
            if world == 1 or args.transport != "auto" or backend != "nccl":
                raise SystemExit("bench.py: the warm-up exchange failed on a rank (%s)" % why)
            transport_state.update(name="rccl", obj=rccl_transport(), fallback="the torch transport failed in the warm-up exchange: %s" % (why or "on another rank"))
            exchange(args.correction, THRESHOLD, acc.read_counters(), acc.bufs["pvalue"], acc.bufs["sign"], n_w)
    K._native.check(lib.kmd_stream_sync(None))
    acc.counters.zero()
    K._native.check(lib.kmd_stream_sync(None))

    ev0, ev1 = K.Event(), K.Event()       # one HIP-event pair around the K launches (same stream)

    # ---- timed region: exactly K steps + the job's exchange and correction -------------------
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        obs.process(mats[i % n_res])
    ev1.record()
    if os.environ.get("KMD_BENCH_TEST_DIE_RANK") == str(rank) and world > 1:     # tests: a rank dies before the exchange
        os._exit(3)
    n_surv = acc.finish(sort=True)
    keep, g_counters, (n_ctrl, n_case) = exchange(args.correction, THRESHOLD, acc.read_counters(), acc.bufs["pvalue"], acc.bufs["sign"], n_surv)
    torch.cuda.synchronize()
    D.barrier()
    elapsed = time.perf_counter() - t0
    my_elapsed = elapsed
    elapsed_min = -D.max_over_ranks(-elapsed)
    elapsed = D.max_over_ranks(elapsed)

    my_kernel_ms = ev0.elapsed_ms(ev1) / args.steps                       # back-to-back launches
    kernel_ms_min = -D.max_over_ranks(-my_kernel_ms)
    avg_kernel_ms = D.max_over_ranks(my_kernel_ms)
    # every rank's own numbers, in rank order (the N > 1 line shows which rank was the slow one)
    per_rank_ms = [[my_kernel_ms, my_elapsed * 1e3 / args.steps]]
    if world > 1:
        box = [None] * world
        dist.all_gather_object(box, per_rank_ms[0])
        per_rank_ms = box
    near = int(acc.read_counters()[K._native.CNT_NEAR_THRESHOLD])    # rows decided with correctly rounded log / exp
    kept = D.allreduce_counters([int(keep.sum()), n_ctrl, n_case, near])

    if rank == 0:
        total_rows = float(args.rows) * args.steps * world
        value = total_rows / elapsed
        achieved = args.rows * BYTES_PER_ROW / (avg_kernel_ms * 1e-3) / 1e9
        copy_gbs = read_gbs = None
        try:
            nb = 1 << 30
            a, b = K.DeviceBuffer(nb), K.DeviceBuffer(nb)
            e0, e1 = K.Event(), K.Event()
            for _ in range(2):
                K._native.check(lib.kmd_copy_probe(b.ptr, a.ptr, nb, None))
            e0.record()
            for _ in range(5):
                K._native.check(lib.kmd_copy_probe(b.ptr, a.ptr, nb, None))
            e1.record()
            copy_gbs = 5 * 2 * nb / (e0.elapsed_ms(e1) * 1e-3) / 1e9
            # bare streaming read, 16-byte non-temporal loads: the measured read ceiling of this GPU
            sink = K.DeviceBuffer(8).zero()
            for _ in range(2):
                K._native.check(lib.kmd_read_probe(a.ptr, nb, 64 + 16, sink.ptr, None))
            e0.record()
            for _ in range(8):
                K._native.check(lib.kmd_read_probe(a.ptr, nb, 64 + 16, sink.ptr, None))
            e1.record()
            read_gbs = 8 * nb / (e0.elapsed_ms(e1) * 1e-3) / 1e9
            a.free(); b.free()
        except Exception:
            pass
        # roofline.frac / frac_events: THIS run's HIP events.  What profiles/ holds of the same command (rocprofv3 cannot
        # run inside this process) is quoted beside it, with its source: the PMC traffic per launch, and the kernel's
        # average duration in the committed kernel-trace CSV (tools/refresh_profiles_r05.sh makes both in one lease and
        # checks the CSV against the HIP events of the profiled run itself: they agree to 3 %; a run under the
        # profiler is 2-3 % slower than one without, MI355X_MICROARCH.md "DVFS give-back")
        traffic, traffic_source, profiled = None, None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_source = "profiles/traffic.json (static: %s; not measured by this run)" % tj.get("source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes")
                if tj.get("profiled_kernel_us") and args.rows == tj.get("rows_per_launch") and args.layout == tj.get("layout", "tiled"):
                    p_ms = tj["profiled_kernel_us"] * 1e-3
                    profiled = {"avg_kernel_ms": p_ms, "frac": args.rows * BYTES_PER_ROW / (p_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "events_ms_same_run": tj.get("profiled_events_ms"), "source": tj.get("profiled_source"),
                                "what": "static: average duration of the kernel in the committed rocprofv3 --kernel-trace --stats CSV of this "
                                        "command, and the HIP-event average of that profiled run"}
            except Exception:
                traffic = None
        out = {
            "metric": "k-mers tested/sec + HBM GB/s (% roofline), k=31, 20v20 samples",
            "value": value, "unit": "k-mers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "backend": backend if world > 1 else None, "rccl_ranks": n_ranks_seen,
            "transport": (transport_state["name"] if world > 1 else None), "transport_fallback": transport_state["fallback"],
            "per_rank": {"ms_per_step_min": elapsed_min * 1e3 / args.steps, "ms_per_step_max": elapsed * 1e3 / args.steps,
                         "kernel_ms_min": kernel_ms_min, "kernel_ms_max": avg_kernel_ms,
                         "kernel_ms": [x[0] for x in per_rank_ms], "ms_per_step": [x[1] for x in per_rank_ms]},
            # value against what the ranks' own kernels sustain (their HIP events): what the barriers, the exchange and the
            # correction cost the job -- NOT the driver's scaling efficiency (that one compares runs at different N)
            "scaling_efficiency": value / sum(args.rows / (x[0] * 1e-3) for x in per_rank_ms),
            "config": {"workload": "configs[2]: 256-partition synthetic matrix, 20v20, k=31, u32 counts; "
                                   "step = one partition of %d rows resident in HBM (%s layout); partition p on "
                                   "rank p %% N" % (args.rows, args.layout),
                       "rows_per_step_per_gpu": args.rows, "bytes_per_row_algorithmic": BYTES_PER_ROW,
                       "threshold": thr, "correction": args.correction, "log_factorial": LOG_FACTORIAL,
                       "resident_partitions": n_res, "device": K.device_name(),
                       "counters": {"total": int(g_counters[0]), "n_sig": int(g_counters[1]),
                                    "n_sig_control": int(g_counters[2]), "n_sig_case": int(g_counters[3]),
                                    "kept_after_correction": int(kept[0]),
                                    "near_threshold": int(kept[3])},
                       "copy_probe_GBs": copy_gbs, "read_probe_GBs": read_gbs},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "frac_events": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_frac": (traffic / (avg_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "profiled": profiled,
                         "kernel": "k_filter_%s<u32>" % ("rows" if args.layout == "rows" else "soa"), "avg_kernel_ms": avg_kernel_ms},
        }
        if world == 1 and not args.no_pipeline:
            try:
                out["h2d_inclusive"] = h2d_leg(K, lib, obs, mats[0])
            except Exception as e:                          # (a box without room for the page-locked copy)
                out["h2d_inclusive"] = {"error": str(e)}
            # the resident matrices of the headline leg make room for the streams of whole partitions (4 x 12 GB)
            del mats[1:]
            K._native.check(lib.kmd_release_cache())
            kept = {}
            out["pipeline"] = pipeline_leg(K, lib, args.pipeline_rows, iters=6, n_distinct=args.pipeline_partitions, keep=kept)
            try:
                out["pipeline"]["feed_inclusive"] = feed_leg(K, lib, kept["ss"], kept["model"], threads=min(8, usable_cpus()[0]))
            except Exception as e:                          # (a host without the memory for the packed copy)
                out["pipeline"]["feed_inclusive"] = {"error": repr(e)}
            kept.clear()
            K._native.check(lib.kmd_release_cache())
            if args.pipeline_rows > 4_000_000:
                out["pipeline"]["small"] = pipeline_leg(K, lib, 4_000_000, iters=6, n_distinct=1, with_extras="batched")
            # rows of few records beside rows of many: the MIXED profile at the same size, with a roofline block of its own
            out["pipeline"]["sparse"] = pipeline_leg(K, lib, args.pipeline_rows, iters=6, n_distinct=1, with_extras="batched", profile=K.SYNTH_MIXED)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.rows, int(totals[:NC].sum()), int(totals[NC:].sum()))
        if args.e2e and world == 1:
            out["e2e"] = e2e_leg()
        print(json.dumps(out), flush=True)
    if transport_state["obj"] is not None:
        K._native.rccl_lib().kmd_transport_rccl_destroy(C.byref(transport_state["obj"]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
