"""The guard of `p <= threshold` (include/kmdiff/merge.hpp:78; SURVEY 7, hard part 2): a bit-identical survivor set
cannot rest on two libms agreeing in the last bit.  Rows whose p-value lies within 1e-8 (relative) of the
threshold are counted (KMD_CNT_NEAR_THRESHOLD) and decided with correctly rounded log / exp (kmd_ddmath.h),
i.e. with the value a glibc-built reference computes wherever glibc's own four calls were correctly rounded
(> 98 % of rows, tests/test_rounded_math.py).  Here the threshold is put ON a row's p-value, and one ulp to
either side, and the device's survivor set is held against the oracle's."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as OL
from test_gpu_parity import totals_of

pytestmark = pytest.mark.gpu
SEED = 0x6B6D64696666


@pytest.fixture(scope="module")
def K():
    import kmdiff_amd as K
    assert K.device_count() >= 1, "no GPU: the HIP path has no CPU fallback"
    return K


def survivors(K, model, mat, n, thr):
    acc = K.SurvivorAccumulator(n)
    K.diff_observer(model, acc, thr).process(mat)
    ns = acc.finish()
    got = acc.get()
    return got["row"].astype(np.int64), got["pvalue"], acc.read_counters()


@pytest.mark.parametrize("lf_n", [10000, 6])
def test_threshold_placed_on_a_p_value(K, oracle, lf_n):
    """lf_n = 6: most survivors have a sum beyond the log-factorial table, where the reference's table term is its running sum
    (log_factorial_table.cpp:13-22) and the filters' Stirling's series; the guard repeats the running sum for such a row."""
    n, nc, nk = 60_000, 6, 6
    host, _, _ = oracle.synth_rows(SEED, 3, 0, n, nc, nk, 4)
    tcs, tks = totals_of(host, nc)
    lf = oracle.lf_table(lf_n)
    ref = oracle.diff_partition(host, OL.LAYOUT_ROWS, nc, nk, int(tcs.sum()), int(tks.sum()), lf, 1e-3)
    rows, ps = ref["row"].astype(np.int64), ref["pvalue"]
    assert len(rows) > 200
    model = K.PoissonLikelihood(nc, nk, tcs, tks, lf_n)
    mat = K.CountMatrix.from_host(host, K.LAYOUT_ROWS)
    lib = K._native.lib()
    # every reference p-value as the device would compute it for a near-threshold row: correctly rounded libm
    sc = host[rows][:, :nc].sum(axis=1, dtype=np.uint64)
    sk = host[rows][:, nc:].sum(axis=1, dtype=np.uint64)
    p_rounded = np.array([lib.kmd_test_row_pvalue_rounded(C.c_void_p(model.handle), int(a), int(b)) for a, b in zip(sc, sk)])
    inside = p_rounded >= 0                                          # (sums inside the log-factorial table, or below 2^20)
    assert inside.all()
    beyond = (sc >= lf_n) | (sk >= lf_n)
    agree = inside & (p_rounded == ps)
    assert agree.sum() >= 0.95 * inside.sum()                        # glibc gave the rounded value itself
    order = np.argsort(ps)
    picks = [i for i in order[:: max(1, len(order) // 40)] if inside[i] and ps[i] > 1e-300][:40]
    assert lf_n >= 100 or sum(bool(beyond[i]) for i in picks) >= 20
    n_checked = 0
    for i in picks:
        for thr in (ps[i], np.nextafter(ps[i], 0.0), np.nextafter(ps[i], 1.0)):
            got_rows, got_p, c = survivors(K, model, mat, n, float(thr))
            assert int(c[K._native.CNT_NEAR_THRESHOLD]) >= 1         # the row on the threshold was seen
            # what the device must do: decide near rows by the rounded value, everything else is far away
            want_rounded = set(rows[(np.where(inside, p_rounded, ps) <= thr)].tolist())
            assert set(got_rows.tolist()) == want_rounded
            if agree[i]:
                # ... which is the reference's own decision wherever glibc returned the rounded value
                near_band = np.abs(ps - thr) <= 1e-8 * thr
                if agree[near_band].all():
                    assert set(got_rows.tolist()) == set(rows[ps <= thr].tolist())
                    n_checked += 1
            # and the p-value reported for the row on the threshold is the rounded one, bit for bit
            k = np.nonzero(got_rows == rows[i])[0]
            if len(k):
                assert got_p[k[0]] == p_rounded[i]
    assert n_checked >= 0.9 * 3 * len(picks)


def test_no_near_threshold_rows_in_an_ordinary_partition(K, oracle):
    """10^7 synthetic rows at the default threshold: the counter stays at zero (nothing to resolve)."""
    n, nc, nk = 10_000_000, 20, 20
    mat = K.synth_matrix(SEED, 1, n, nc, nk, 4, K.LAYOUT_TILED)
    tot = K.column_sums(mat)
    model = K.PoissonLikelihood(nc, nk, tot[:nc], tot[nc:], 10000)
    acc = K.SurvivorAccumulator(n // 100)
    K.diff_observer(model, acc, 0.05 / 100000).process(mat)
    c = acc.read_counters()
    assert int(c[K._native.CNT_SIG]) > 100 and int(c[K._native.CNT_NEAR_THRESHOLD]) == 0


def test_near_list_survives_streams_and_cache_release(K, oracle):
    """The list of near-threshold rows is kept per stream and handed back empty by the kernel that resolves it:
    launches on streams that are created and destroyed in between, and after kmd_release_cache, decide the row
    on the threshold the same way every time."""
    n, nc, nk = 20_000, 5, 5
    host, _, _ = oracle.synth_rows(SEED, 4, 0, n, nc, nk, 4)
    tcs, tks = totals_of(host, nc)
    ref = oracle.diff_partition(host, OL.LAYOUT_ROWS, nc, nk, int(tcs.sum()), int(tks.sum()), oracle.lf_table(10000), 1e-3)
    ps = ref["pvalue"]
    thr = float(np.sort(ps[ps > 1e-300])[len(ps) // 2])                  # ON a row's p-value
    model = K.PoissonLikelihood(nc, nk, tcs, tks, 10000)
    mat = K.CountMatrix.from_host(np.ascontiguousarray(host.T), K.LAYOUT_SOA)
    lib = K._native.lib()

    def run(stream):
        acc = K.SurvivorAccumulator(n)
        K.diff_observer(model, acc, thr).process(mat, stream=stream)
        if stream is not None:
            assert lib.kmd_stream_sync(stream) == 0
        ns = acc.finish()
        c = acc.read_counters()
        return ns, sorted(acc.get()["row"].tolist()), int(c[K._native.CNT_NEAR_THRESHOLD])

    first = run(None)
    assert first[2] >= 1
    for _ in range(3):
        st = C.c_void_p()
        assert lib.kmd_stream_create(C.byref(st)) == 0
        assert run(st) == first and run(st) == first              # twice on the same stream: the list came back empty
        assert lib.kmd_stream_destroy(st) == 0
    assert lib.kmd_release_cache() == 0
    assert run(None) == first


def test_threshold_on_a_p_value_through_the_fused_merge(K, oracle):
    """The same guard behind kmd_merge_filter (rows reach the exact evaluation as candidates of the merge; their list
    entries name the candidate, the survivor's `row` is its k-mer): thresholds ON a row's p-value and one ulp either
    side give the survivors the matrix path gives -- which test_threshold_placed_on_a_p_value holds to the oracle."""
    n, nc, nk = 30_000, 6, 6
    host, lo, _ = oracle.synth_rows(SEED, 5, 0, n, nc, nk, 4)
    tcs, tks = totals_of(host, nc)
    ref = oracle.diff_partition(host, OL.LAYOUT_ROWS, nc, nk, int(tcs.sum()), int(tks.sum()), oracle.lf_table(10000), 1e-3)
    ps = np.sort(ref["pvalue"][ref["pvalue"] > 1e-300])
    assert len(ps) > 100
    model = K.PoissonLikelihood(nc, nk, tcs, tks, 10000)
    mat = K.CountMatrix.from_host(host, K.LAYOUT_ROWS, kmer_lo=lo)
    ss = K.StreamSet([(lo[host[:, s] > 0], host[host[:, s] > 0, s]) for s in range(nc + nk)])
    seen_near = 0
    for p in ps[:: max(1, len(ps) // 12)][:12]:
        for thr in (float(p), float(np.nextafter(p, 0.0)), float(np.nextafter(p, 1.0))):
            a = K.SurvivorAccumulator(n)
            K.diff_observer(model, a, thr).process(mat)
            na = a.finish()
            ga, ca = a.get(), a.read_counters()
            b = K.SurvivorAccumulator(n)
            assert K.merge_filter(ss, K.diff_observer(model, b, thr)) == n
            nb = b.finish(by_kmer=True)
            gb, cb = b.get(), b.read_counters()
            assert nb == na and [int(x) for x in cb[:4]] == [int(x) for x in ca[:4]]
            assert int(cb[K._native.CNT_NEAR_THRESHOLD]) == int(ca[K._native.CNT_NEAR_THRESHOLD]) >= 1
            order = np.argsort(ga["kmer_lo"], kind="stable")
            assert gb["kmer_lo"].tolist() == ga["kmer_lo"][order].tolist()
            assert gb["pvalue"].tolist() == ga["pvalue"][order].tolist()        # incl. the rounded p of the row on the threshold
            assert gb["sign"].tolist() == ga["sign"][order].tolist()
            seen_near += int(cb[K._native.CNT_NEAR_THRESHOLD])
    assert seen_near >= 36


def test_near_threshold_flips_in_a_batch_that_shares_one_sink(K, oracle):
    """kmd_merge_filter_batch with every partition naming the SAME sink and counters: the pass over a
    partition's near-threshold rows strikes records out and compacts the whole sink, so the candidate steps of such
    partitions must not overlap.  Forced here (KMD_TEST_NEAR_FLIP: every listed row's decision is turned over, so each
    partition strikes some records and appends others) with the threshold ON a p-value many rows share: the shared sink
    of the batch holds exactly what the partitions' single calls put into sinks of their own."""
    import os
    n, nc, nk, parts = 30_000, 6, 6, 9
    lf = oracle.lf_table(10000)
    hosts = [oracle.synth_rows(SEED, 20 + p, 0, n, nc, nk, 4) for p in range(parts)]
    tcs, tks = totals_of(np.concatenate([h[0] for h in hosts]), nc)
    ref = oracle.diff_partition(hosts[0][0], OL.LAYOUT_ROWS, nc, nk, int(tcs.sum()), int(tks.sum()), lf, 1e-2)
    vals, cnt = np.unique(ref["pvalue"][ref["pvalue"] > 1e-300], return_counts=True)
    thr = float(vals[np.argmax(cnt)])                                   # the p-value most rows share: many near rows per partition
    assert cnt.max() >= 5
    model = K.PoissonLikelihood(nc, nk, tcs, tks, 10000)
    sets = [K.StreamSet([(lo[h[:, s] > 0], h[h[:, s] > 0, s]) for s in range(nc + nk)]) for h, lo, _ in hosts]
    os.environ["KMD_TEST_NEAR_FLIP"] = "1"
    try:
        singles, near_total, c_sum = [], 0, np.zeros(4, dtype=np.int64)
        for ss in sets:
            a = K.SurvivorAccumulator(n)
            assert K.merge_filter(ss, K.diff_observer(model, a, thr)) == n
            a.finish(by_kmer=True)
            g, c = a.get(), a.read_counters()
            assert int(c[K._native.CNT_NEAR_THRESHOLD]) >= 1 and int(c[K._native.CNT_NEAR_UNRESOLVED]) == 0
            near_total += int(c[K._native.CNT_NEAR_THRESHOLD])
            c_sum += np.array([int(x) for x in c[:4]])
            singles.append(g)
        assert near_total >= 3 * parts
        for _ in range(3):                                              # (a race shows up in some runs only)
            shared = K.SurvivorAccumulator(parts * n)
            obs = [K.diff_observer(model, shared, thr) for _ in range(parts)]
            assert K.merge_filter_batch(sets, obs) == [n] * parts
            ns = shared.finish(by_kmer=True)
            g, c = shared.get(), shared.read_counters()
            assert [int(x) for x in c[:4]] == c_sum.tolist() and ns == int(c_sum[1])
            want_k = np.concatenate([x["kmer_lo"] for x in singles])
            order = np.argsort(want_k, kind="stable")
            assert g["kmer_lo"].tolist() == want_k[order].tolist()
            for f in ("pvalue", "sign", "mean_control", "mean_case"):
                assert g[f].tolist() == np.concatenate([x[f] for x in singles])[order].tolist(), f
    finally:
        del os.environ["KMD_TEST_NEAR_FLIP"]


def test_threshold_on_a_p_value_of_a_row_beyond_the_table(K, oracle):
    """Count sums of thousands against --log-factorial 300: the filters take Stirling's series for ln k!, the reference its
    k-term running sum (log_factorial_table.cpp:13-22) -- 5e-10 apart on p.  A threshold ON such a row's p-value (and one ulp
    to either side) is decided by the guard, which repeats the running sum: the survivor set is the oracle's wherever glibc
    gave the oracle the rounded logarithms, and the row on the threshold reports the rounded p-value."""
    from test_gpu_tilemerge import make_streams
    rng = np.random.default_rng(77)
    S, nc, lf_n = 40, 20, 300
    universe = np.unique(rng.integers(0, 1 << 62, 4000, dtype=np.uint64))
    streams = make_streams(rng, universe, S, rng.uniform(0.3, 1.0, S), count_hi=300)
    want, wlo = oracle.merge_partition(streams)
    tcs, tks = totals_of(want, nc)
    ref = oracle.diff_partition(want, OL.LAYOUT_ROWS, nc, S - nc, int(tcs.sum()), int(tks.sum()), oracle.lf_table(lf_n), 1.0)
    ps = ref["pvalue"]
    sc, sk = want[:, :nc].sum(axis=1, dtype=np.uint64), want[:, nc:].sum(axis=1, dtype=np.uint64)
    assert ((sc >= lf_n) & (sk >= lf_n)).all() and int(sc.max()) > 2000
    model = K.PoissonLikelihood(nc, S - nc, tcs, tks, lf_n)
    lib = K._native.lib()
    p_rounded = np.array([lib.kmd_test_row_pvalue_rounded(C.c_void_p(model.handle), int(a), int(b)) for a, b in zip(sc, sk)])
    agree = p_rounded == ps
    assert agree.mean() > 0.97
    ss = K.StreamSet(streams)
    picks = [i for i in np.argsort(ps)[:: len(ps) // 14] if 1e-300 < ps[i] < 0.9][:12]
    n_oracle = 0
    for i in picks:
        for thr in (float(ps[i]), float(np.nextafter(ps[i], 0.0)), float(np.nextafter(ps[i], 1.0))):
            acc = K.SurvivorAccumulator(len(ps))
            K.merge_filter(ss, K.diff_observer(model, acc, thr))
            acc.finish(by_kmer=True)
            got, c = acc.get(), acc.read_counters()
            assert int(c[K._native.CNT_NEAR_THRESHOLD]) >= 1
            assert got["kmer_lo"].tolist() == wlo[p_rounded <= thr].tolist()
            near_band = np.abs(ps - thr) <= 1e-8 * thr
            if agree[near_band].all():
                assert got["kmer_lo"].tolist() == wlo[ps <= thr].tolist()
                n_oracle += 1
            k = np.nonzero(got["kmer_lo"] == wlo[i])[0]
            if len(k):
                assert got["pvalue"][k[0]] == p_rounded[i]
    assert n_oracle >= 30


@pytest.mark.parametrize("layout", ["tiled", "rows", "soa"])
def test_more_near_rows_than_one_launch_lists_in_pieces(K, oracle, layout):
    """KMD_CNT_NEAR_UNRESOLVED through the Python mirror: 6 000 equal rows with the threshold ON their p-value
    are more rows within 1e-8 of it than one launch lists (4096) -- finish() refuses the sink; diff_observer.
    process_in_pieces runs the matrix again in windows of 4096 rows, none of which can overflow the list, and the sink
    then holds exactly the oracle's set."""
    nc, nk, n_bg, n_same = 4, 4, 20_000, 6_000
    host, _, _ = oracle.synth_rows(SEED, 9, 0, n_bg, nc, nk, 4)
    lf = oracle.lf_table(10000)
    lib = K._native.lib()
    chosen = None
    for cand in [[c0, 0, 0, 0, a, b, 2, 1] for c0 in (0, 1) for a in (2, 3, 4, 5) for b in (1, 2, 3)]:
        row = np.array(cand, dtype=np.uint32)
        full = np.vstack([host[:n_bg // 2], np.tile(row, (n_same, 1)), host[n_bg // 2:]])
        tot = full.sum(axis=0, dtype=np.uint64)
        one = oracle.diff_partition(row[None, :], OL.LAYOUT_ROWS, nc, nk, int(tot[:nc].sum()), int(tot[nc:].sum()), lf, 1.0)
        p_star = float(one["pvalue"][0])
        model = K.PoissonLikelihood(nc, nk, tot[:nc], tot[nc:], 10000)
        if lib.kmd_test_row_pvalue_rounded(C.c_void_p(model.handle), int(row[:nc].sum()), int(row[nc:].sum())) == p_star and 1e-12 < p_star < 1e-2:
            chosen = (full, tot, p_star, model)
            break
    assert chosen is not None
    full, tot, p_star, model = chosen
    want = oracle.diff_partition(full, OL.LAYOUT_ROWS, nc, nk, int(tot[:nc].sum()), int(tot[nc:].sum()), lf, p_star)
    assert (want["pvalue"] == p_star).sum() >= n_same
    lay = {"tiled": K.LAYOUT_TILED, "rows": K.LAYOUT_ROWS, "soa": K.LAYOUT_SOA}[layout]
    mat = K.CountMatrix.from_host(full if layout == "rows" else np.ascontiguousarray(full.T), lay)      # ([sample][row] but for row-major)
    acc = K.SurvivorAccumulator(len(full))
    obs = K.diff_observer(model, acc, p_star)
    obs.process(mat)
    assert int(acc.read_counters()[K._native.CNT_NEAR_UNRESOLVED]) > 0
    with pytest.raises(K._native.KmdError):
        acc.finish()
    acc.counters.zero()
    obs.process_in_pieces(mat, 4096)
    c = acc.read_counters()
    assert int(c[K._native.CNT_NEAR_UNRESOLVED]) == 0 and int(c[K._native.CNT_NEAR_THRESHOLD]) >= n_same
    assert int(c[K._native.CNT_TOTAL]) == len(full)
    acc.finish()
    got = acc.get()
    assert got["row"].tolist() == want["row"].tolist() and got["sign"].tolist() == want["sign"].tolist()
