"""Parity of the HIP path (through the C-ABI of libkmdiff_hip.so) with the CPU oracle and the
committed golden vectors.  Run on the MI355X box:  python -m pytest tests -m gpu

Bars (BASELINE.json north_star):
  * k-mer identity, row index, sign, counters, mean_control, mean_case: bit-exact;
  * p-values: |p_hip - p_ref| <= 1e-10 absolute (P_ABS_TOL); the tests additionally hold the
    device to a relative 1e-9 (P_REL_TOL) because survivors have p << 1e-10/1e-3.
The device uses ROCm's ocml log/exp, the oracle glibc's: they differ by <= 1 ulp, which is the
only source of p differences (the operation order is the reference's on both sides).
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as OL

pytestmark = pytest.mark.gpu

P_ABS_TOL = 1e-10
P_REL_TOL = 1e-9
SEED = 0x6B6D64696666
THR = 0.05 / 100000          # -s 0.05 / -u 100000 (cmd/diff.hpp:147)
fh = float.fromhex


@pytest.fixture(scope="module")
def K():
    import kmdiff_amd as K
    assert K.device_count() >= 1, "no GPU: the HIP path has no CPU fallback"
    return K


def assert_p_close(got, want):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape
    assert np.abs(got - want).max(initial=0.0) <= P_ABS_TOL
    nz = want > 0
    assert (got[~nz] == 0).all() or np.abs(got[~nz]).max(initial=0.0) < 1e-300
    rel = np.abs(got[nz] - want[nz]) / want[nz]
    assert rel.max(initial=0.0) <= P_REL_TOL, rel.max()


def totals_of(counts_rows, nc):
    t = counts_rows.sum(axis=0, dtype=np.uint64)
    return t[:nc], t[nc:]


# ---------------------------------------------------------------------------------------------
def test_log_factorial_table_is_the_reference_table(K, oracle):
    m = K.PoissonLikelihood(2, 2, [1, 1], [1, 1], 2000)
    assert m.lf_table().tolist() == oracle.lf_table(2000).tolist()
    t = m.lf_table()
    assert t[10] == 15.104412573075514            # tests/factorial_test.cpp:12


def layout_of(K, name):
    return {"rows": K.LAYOUT_ROWS, "soa": K.LAYOUT_SOA, "tiled": K.LAYOUT_TILED}[name]


@pytest.mark.parametrize("layout_name", ["rows", "soa", "tiled"])
@pytest.mark.parametrize("dtype", [np.uint32, np.uint16, np.uint8])
def test_process_golden_rows(K, golden_dir, layout_name, dtype):
    """IModel::process over the golden rows made by the reference's own sources."""
    with open(os.path.join(golden_dir, "poisson_rows.json")) as f:
        g = json.load(f)
    layout = layout_of(K, layout_name)
    for case in g["cases"]:
        rows = np.array(case["rows"], dtype=np.uint32)
        if rows.max() > np.iinfo(dtype).max:
            continue
        rows = rows.astype(dtype)
        model = K.PoissonLikelihood(case["nc"], case["nk"], case["total_controls"], case["total_cases"],
                                    case["preload"])
        mat = K.CountMatrix.from_host(rows if layout == K.LAYOUT_ROWS else rows.T, layout)
        p, s, mc, mk = model.process(mat)
        assert s.tolist() == case["sign"]
        assert [float(x).hex() for x in mc] == case["mean_control"]
        assert [float(x).hex() for x in mk] == case["mean_case"]
        assert_p_close(p, [fh(x) for x in case["p"]])


def test_reference_model_test_signs(K):
    """tests/model_test.cpp:45-81 through the HIP model."""
    model = K.PoissonLikelihood(30, 30, [1] * 30, [1] * 30, 10)
    v = np.array([[200] * 30 + [100] * 30, [100] * 30 + [200] * 30, [100] * 60], dtype=np.uint32)
    p, s, mc, mk = model.process(K.CountMatrix.from_host(v, K.LAYOUT_ROWS))
    assert s.tolist() == [K.SIGN_CONTROL, K.SIGN_CASE, K.SIGN_NO]
    assert abs(p[0] - 1.0932047323640264e-223) <= 1e-9 * 1.0932047323640264e-223
    assert (mc[0], mk[0]) == (6000.0, 3000.0)


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("layout_name", ["soa", "rows", "tiled"])
@pytest.mark.parametrize("count_bytes", [4, 2, 1])
def test_synth_device_equals_oracle_replay(K, oracle, layout_name, count_bytes):
    layout = layout_of(K, layout_name)
    n = 20011
    m = K.synth_matrix(SEED, 5, n, 3, 4, count_bytes, layout, row0=1234)
    want, lo, _ = oracle.synth_rows(SEED, 5, 1234, n, 3, 4, count_bytes)
    assert (m.to_host() == want).all()
    assert (m.kmers_to_host()[0] == lo).all()
    m2 = K.synth_matrix(SEED, 5, 100, 3, 4, 4, layout, kmer_limbs=2)
    _, lo2, hi2 = oracle.synth_rows(SEED, 5, 0, 100, 3, 4, 4, kmer_limbs=2)
    glo, ghi = m2.kmers_to_host()
    assert (glo == lo2).all() and (ghi == hi2).all()
    assert (K.column_sums(m) == want.sum(axis=0, dtype=np.uint64)).all()


def run_filter(K, mat, nc, nk, tcs, tks, preload, thr, cap=None, limbs=1):
    model = K.PoissonLikelihood(nc, nk, tcs, tks, preload)
    acc = K.SurvivorAccumulator(mat.n_rows if cap is None else cap, kmer_limbs=limbs)
    obs = K.diff_observer(model, acc, thr, nc, nk)
    obs.process(mat)
    n = acc.finish()
    return obs, acc, n


def check_against_oracle(K, oracle, mat, host_rows, kmers, nc, nk, preload, thr):
    tcs, tks = totals_of(host_rows, nc)
    obs, acc, n = run_filter(K, mat, nc, nk, tcs, tks, preload, thr, limbs=2 if kmers[1] is not None else 1)
    want = oracle.diff_partition(host_rows, OL.LAYOUT_ROWS, nc, nk, int(tcs.sum()), int(tks.sum()),
                                 oracle.lf_table(preload), thr)
    got = acc.get()
    c = acc.read_counters()
    assert (int(c[0]), int(c[1]), int(c[2]), int(c[3])) == want["counters"]
    assert obs.total() == host_rows.shape[0] and obs.nb_sign() == n
    assert obs.nb_signs() == (want["counters"][2], want["counters"][3])
    assert (got["row"] - mat.row_base).tolist() == want["row"].tolist()      # identity + order
    assert got["sign"].tolist() == want["sign"].tolist()
    assert got["mean_control"].tolist() == want["mean_control"].tolist()
    assert got["mean_case"].tolist() == want["mean_case"].tolist()
    assert_p_close(got["pvalue"], want["pvalue"])
    idx = want["row"].astype(np.int64)
    if kmers[0] is not None:
        assert got["kmer_lo"].tolist() == kmers[0][idx].tolist()
    if kmers[1] is not None:
        assert got["kmer_hi"].tolist() == kmers[1][idx].tolist()
    return want["counters"]


@pytest.mark.parametrize("layout_name", ["soa", "rows", "tiled"])
@pytest.mark.parametrize("count_bytes,nc,nk", [(4, 4, 4), (4, 20, 20), (2, 5, 3), (1, 7, 9), (4, 50, 50)])
def test_filter_matches_oracle_on_synthetic_partition(K, oracle, layout_name, count_bytes, nc, nk):
    """diff_observer over one synthetic partition == the oracle's row loop."""
    layout = layout_of(K, layout_name)
    n = 150_003 if nc + nk <= 40 else 40_001
    limbs = 2 if nc == 50 else 1
    mat = K.synth_matrix(SEED, 2, n, nc, nk, count_bytes, layout, kmer_limbs=limbs)
    host, lo, hi = oracle.synth_rows(SEED, 2, 0, n, nc, nk, count_bytes, kmer_limbs=limbs)
    counters = check_against_oracle(K, oracle, mat, host, (lo, hi), nc, nk, 10000, THR)
    assert counters[1] > 0            # the planted signal produces survivors


@pytest.mark.parametrize("layout_name", ["soa", "rows", "tiled"])
def test_table_fallback_small_preload(K, oracle, layout_name):
    """Count sums >= --log-factorial take LogFactorialTable's O(k) fallback
    (log_factorial_table.hpp:14-18): wave-cooperative on the device."""
    layout = layout_of(K, layout_name)
    n = 30_000
    mat = K.synth_matrix(SEED, 9, n, 6, 6, 4, layout)
    host, lo, _ = oracle.synth_rows(SEED, 9, 0, n, 6, 6, 4)
    for preload in (64, 1, 0):
        check_against_oracle(K, oracle, mat, host, (lo, None), 6, 6, preload, THR)
    tcs, tks = totals_of(host, 6)
    obs, acc, _ = run_filter(K, mat, 6, 6, tcs, tks, 64, THR)
    assert int(acc.read_counters()[5]) > 0          # rows that went through the fallback


@pytest.mark.parametrize("thr", [1.0, 0.5, 1e-3, 1e-30, 0.0])
def test_thresholds_from_everything_to_nothing(K, oracle, thr):
    n = 8_191
    mat = K.synth_matrix(SEED, 1, n, 4, 4, 4, K.LAYOUT_SOA)
    host, lo, _ = oracle.synth_rows(SEED, 1, 0, n, 4, 4, 4)
    c = check_against_oracle(K, oracle, mat, host, (lo, None), 4, 4, 10000, thr)
    if thr >= 1.0:
        assert c[1] == n


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 2047, 2048, 2049, 4099])
def test_empty_and_ragged_tiles(K, oracle, n):
    for layout in (K.LAYOUT_SOA, K.LAYOUT_ROWS, K.LAYOUT_TILED):
        mat = K.synth_matrix(SEED, 0, n, 4, 4, 4, layout)
        host, lo, _ = oracle.synth_rows(SEED, 0, 0, n, 4, 4, 4)
        if n == 0:
            model = K.PoissonLikelihood(4, 4, [1] * 4, [1] * 4, 100)
            acc = K.SurvivorAccumulator(4)
            K.diff_observer(model, acc, 1.0).process(mat)
            assert acc.finish() == 0 and int(acc.read_counters()[0]) == 0
            continue
        check_against_oracle(K, oracle, mat, host, (lo, None), 4, 4, 10000, 0.01)


def test_unaligned_soa_pitch_and_padded_rows(K, oracle):
    """SoA with a column pitch that is not a multiple of 16 bytes (scalar-load kernel) and
    row-major with ld > nc+nk give the same survivors."""
    n, nc, nk = 10_007, 4, 4
    host, lo, _ = oracle.synth_rows(SEED, 6, 0, n, nc, nk, 4)
    tcs, tks = totals_of(host, nc)
    want = oracle.diff_partition(host, OL.LAYOUT_ROWS, nc, nk, int(tcs.sum()), int(tks.sum()),
                                 oracle.lf_table(10000), 0.01)
    # SoA, ld = n (odd): columns are only 4-byte aligned
    m = K.CountMatrix(n, nc + nk, 4, K.LAYOUT_SOA, ld=n, with_kmers=False)
    host_t = np.ascontiguousarray(host.T)          # keep alive across the ctypes call
    K._native.check(K._native.lib().kmd_memcpy_h2d(m.counts.ptr, host_t.ctypes.data, host_t.nbytes, None))
    _, acc, _ = run_filter(K, m, nc, nk, tcs, tks, 10000, 0.01)
    assert (acc.get()["row"]).tolist() == want["row"].tolist()
    # row-major, ld = 11 (padding column garbage must be ignored)
    padded = np.full((n, 11), 77, dtype=np.uint32)
    padded[:, :8] = host
    m = K.CountMatrix(n, nc + nk, 4, K.LAYOUT_ROWS, ld=11, with_kmers=False)
    K._native.check(K._native.lib().kmd_memcpy_h2d(m.counts.ptr, padded.ctypes.data, padded.nbytes, None))
    _, acc, _ = run_filter(K, m, nc, nk, tcs, tks, 10000, 0.01)
    assert (acc.get()["row"]).tolist() == want["row"].tolist()
    # u8 rows with an odd number of samples: rows are not dword aligned (direct kernel)
    host8, _, _ = oracle.synth_rows(SEED, 6, 0, n, 4, 3, 1)
    t8c, t8k = totals_of(host8, 4)
    want8 = oracle.diff_partition(host8, OL.LAYOUT_ROWS, 4, 3, int(t8c.sum()), int(t8k.sum()),
                                  oracle.lf_table(10000), 0.01)
    m = K.CountMatrix.from_host(host8, K.LAYOUT_ROWS)
    _, acc, _ = run_filter(K, m, 4, 3, t8c, t8k, 10000, 0.01)
    assert (acc.get()["row"]).tolist() == want8["row"].tolist()


def test_row_major_16_byte_pitch_shapes(K, oracle):
    """Row-major with a 16-byte aligned pitch takes the wave-private kernel: one pass per row
    (<= 4 / 8 / 10 / 16 vectors), several passes (wider rows), a last vector that reaches into
    the padding (garbage there must be ignored), control/case boundary inside a vector, fewer
    rows than one wave, rows that are not a multiple of 64."""
    rng = np.random.default_rng(77)
    lf = oracle.lf_table(10000)
    shapes = [(1, 1, 4), (3, 4, 4), (4, 4, 4), (7, 9, 4), (9, 7, 2), (16, 16, 4), (20, 20, 4), (20, 20, 2), (20, 20, 1),
              (17, 30, 4), (33, 35, 4), (100, 100, 4), (100, 100, 1), (5, 6, 1), (31, 33, 2), (40, 41, 4),
              (61, 70, 4), (130, 131, 4), (300, 260, 2), (700, 500, 1), (129, 1, 4)]       # >= 32 vectors: 16 lanes per row
    for it, (nc, nk, cb) in enumerate(shapes):
        S = nc + nk
        n = int(rng.choice([1, 63, 64, 65, 1000, 4099]))
        host, lo, _ = oracle.synth_rows(SEED + 100 + it, it, 0, n, nc, nk, cb)
        per16 = 16 // cb
        ld = (S + per16 - 1) // per16 * per16 + (per16 if it % 3 == 0 else 0)      # sometimes a whole spare vector
        padded = rng.integers(1, 200, (n, ld)).astype(host.dtype)                   # garbage in the padding
        padded[:, :S] = host
        m = K.CountMatrix(n, S, cb, K.LAYOUT_ROWS, ld=ld, with_kmers=False)
        K._native.check(K._native.lib().kmd_memcpy_h2d(m.counts.ptr, padded.ctypes.data, padded.nbytes, None))
        tcs, tks = totals_of(host, nc)
        want = oracle.diff_partition(host, OL.LAYOUT_ROWS, nc, nk, int(tcs.sum()), int(tks.sum()), lf, 0.05)
        obs, acc, ns = run_filter(K, m, nc, nk, tcs, tks, 10000, 0.05)
        got = acc.get()
        assert obs.total() == n, (nc, nk, cb, n)
        assert got["row"].tolist() == want["row"].tolist(), (nc, nk, cb, n, ld)
        assert got["sign"].tolist() == want["sign"].tolist()
        assert np.allclose(got["pvalue"], want["pvalue"], rtol=0, atol=1e-10)
        assert got["mean_control"].tolist() == want["mean_control"].tolist() and got["mean_case"].tolist() == want["mean_case"].tolist()


def test_row_major_any_pitch_shapes(K, oracle):
    """Row-major with a pitch that is NOT a multiple of 16 bytes (21v21 four-byte counts: 168 B) takes
    the flat wave-private kernel: 64 rows per span, or 32 ... 4 with several lanes per row when the rows
    are wide or only a smaller group of rows keeps spans 16-byte aligned (odd byte pitches); the
    control/case boundary at any position of the lane interleave; padding columns full of garbage;
    matrices whose last 16-byte vector is cut by the end of the buffer; fewer rows than one span."""
    rng = np.random.default_rng(78)
    lf = oracle.lf_table(10000)
    # (nc, nk, count bytes, spare columns)
    shapes = [(21, 21, 4, 0), (20, 21, 4, 0), (10, 11, 4, 0), (3, 3, 4, 0), (1, 1, 4, 0), (2, 1, 4, 0), (21, 21, 2, 0),
              (20, 21, 2, 0), (1, 2, 2, 0), (20, 20, 1, 0), (20, 21, 1, 0), (1, 2, 1, 0), (4, 3, 1, 0), (101, 101, 4, 0),
              (64, 65, 4, 0), (129, 130, 4, 0), (250, 251, 4, 0), (300, 261, 2, 0), (500, 501, 1, 0), (33, 32, 4, 0),
              (20, 20, 4, 1), (20, 20, 4, 3), (20, 20, 2, 3), (20, 20, 1, 7), (7, 6, 4, 2), (60, 61, 1, 0), (5, 5, 2, 1)]
    for it, (nc, nk, cb, spare) in enumerate(shapes):
        S = nc + nk
        ld = S + spare
        assert (ld * cb) % 16 != 0, (nc, nk, cb, spare)
        n = int(rng.choice([1, 3, 5, 31, 63, 64, 65, 1000, 4099, 8193]))
        host, lo, _ = oracle.synth_rows(SEED + 300 + it, it, 0, n, nc, nk, cb)
        padded = rng.integers(1, 200, (n, ld)).astype(host.dtype)
        padded[:, :S] = host
        m = K.CountMatrix(n, S, cb, K.LAYOUT_ROWS, ld=ld, with_kmers=False)
        K._native.check(K._native.lib().kmd_memcpy_h2d(m.counts.ptr, padded.ctypes.data, padded.nbytes, None))
        tcs, tks = totals_of(host, nc)
        want = oracle.diff_partition(host, OL.LAYOUT_ROWS, nc, nk, int(tcs.sum()), int(tks.sum()), lf, 0.05)
        obs, acc, ns = run_filter(K, m, nc, nk, tcs, tks, 10000, 0.05)
        got = acc.get()
        assert obs.total() == n, (nc, nk, cb, n)
        assert got["row"].tolist() == want["row"].tolist(), (nc, nk, cb, n, ld)
        assert got["sign"].tolist() == want["sign"].tolist()
        assert np.allclose(got["pvalue"], want["pvalue"], rtol=0, atol=1e-10)
        assert got["mean_control"].tolist() == want["mean_control"].tolist() and got["mean_case"].tolist() == want["mean_case"].tolist()


def test_survivor_capacity_overflow_is_reported(K, oracle):
    n = 5000
    mat = K.synth_matrix(SEED, 1, n, 4, 4, 4, K.LAYOUT_SOA)
    host, _, _ = oracle.synth_rows(SEED, 1, 0, n, 4, 4, 4)
    tcs, tks = totals_of(host, 4)
    model = K.PoissonLikelihood(4, 4, tcs, tks, 10000)
    acc = K.SurvivorAccumulator(10)
    K.diff_observer(model, acc, 1.0).process(mat)
    with pytest.raises(K.KmdError):
        acc.finish()
    c = acc.read_counters()
    assert int(c[1]) == n and int(c[2]) + int(c[3]) == n      # counters stay exact


def test_bad_arguments_fail_loudly(K):
    model = K.PoissonLikelihood(2, 2, [1, 1], [1, 1], 10)
    acc = K.SurvivorAccumulator(4)
    mat = K.CountMatrix(10, 5, 4, K.LAYOUT_SOA)             # 5 samples, model has 4
    with pytest.raises(ValueError):
        K.diff_observer(model, acc, 0.5).process(mat)
    with pytest.raises(K.KmdError):
        K.PoissonLikelihood(0, 2, [], [1, 1], 10)
    t = mat.tile()
    t.count_bytes = 3
    import ctypes as C
    s = acc.struct()
    rc = K._native.lib().kmd_poisson_filter(model.handle, C.byref(t), 0.5, C.byref(s), acc.counters.ptr, None)
    assert rc == -1 and b"count_bytes" in K._native.lib().kmd_last_error()


def test_random_shapes_fuzz(K, oracle):
    """60 random (samples, rows, count width, layout, table size, threshold) combinations,
    each checked row for row against the oracle."""
    rng = np.random.default_rng(20261001)
    layouts = [K.LAYOUT_SOA, K.LAYOUT_ROWS, K.LAYOUT_TILED]
    for it in range(60):
        nc, nk = int(rng.integers(1, 24)), int(rng.integers(1, 24))
        n = int(rng.integers(1, 9000))
        cb = int(rng.choice([1, 2, 4]))
        layout = layouts[int(rng.integers(0, 3))]
        preload = int(rng.choice([0, 3, 40, 700, 10000]))
        thr = float(rng.choice([1.0, 0.3, 1e-2, 1e-4, 5e-7, 1e-12]))
        part = int(rng.integers(0, 256))
        row0 = int(rng.integers(0, 10 ** 9))
        mat = K.synth_matrix(SEED + it, part, n, nc, nk, cb, layout, row0=row0)
        host, lo, _ = oracle.synth_rows(SEED + it, part, row0, n, nc, nk, cb)
        assert (mat.to_host() == host).all()
        check_against_oracle(K, oracle, mat, host, (lo, None), nc, nk, preload, thr)


def test_two_tiles_equal_one_partition(K, oracle):
    """A partition streamed as two tiles (row_base) accumulates to the same survivors."""
    n, nc, nk = 100_000, 4, 4
    host, lo, _ = oracle.synth_rows(SEED, 3, 0, n, nc, nk, 4)
    tcs, tks = totals_of(host, nc)
    model = K.PoissonLikelihood(nc, nk, tcs, tks, 10000)
    acc = K.SurvivorAccumulator(n)
    obs = K.diff_observer(model, acc, THR)
    a = K.synth_matrix(SEED, 3, 60_000, nc, nk, 4, K.LAYOUT_SOA, row0=0)
    b = K.synth_matrix(SEED, 3, 40_000, nc, nk, 4, K.LAYOUT_ROWS, row0=60_000)
    obs.process(a)
    obs.process(b)
    acc.finish()
    want = oracle.diff_partition(host, OL.LAYOUT_ROWS, nc, nk, int(tcs.sum()), int(tks.sum()),
                                 oracle.lf_table(10000), THR)
    got = acc.get()
    assert got["row"].tolist() == want["row"].tolist()
    assert got["kmer_lo"].tolist() == lo[want["row"].astype(np.int64)].tolist()
    assert obs.total() == n


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["nothing", "bonferroni", "sidak", "benjamini", "holm"])
def test_correction_matches_oracle(K, oracle, name):
    n, nc, nk = 200_000, 4, 4
    mat = K.synth_matrix(SEED, 4, n, nc, nk, 4, K.LAYOUT_SOA)
    tot = K.column_sums(mat)
    model = K.PoissonLikelihood(nc, nk, tot[:nc], tot[nc:], 10000)
    acc = K.SurvivorAccumulator(n)
    K.diff_observer(model, acc, 1e-3).process(mat)
    ns = acc.finish()
    assert ns > 50
    got = acc.get()
    ctype = K.CORRECTION_BY_NAME[name]
    for total in (n, 50 * n):
        keep, n_ctrl, n_case = K.aggregate(name, 0.05, total, acc.bufs["pvalue"], acc.bufs["sign"], ns)
        want = oracle.aggregate(ctype, 0.05, total, got["pvalue"])
        assert keep.tolist() == want.tolist()
        assert n_ctrl == int(((got["sign"] == 0) & (want == 1)).sum())
        assert n_case == int(((got["sign"] != 0) & (want == 1)).sum())
    keep, _, _ = K.aggregate(name, 0.05, n, acc.bufs["pvalue"], acc.bufs["sign"], 0)
    assert len(keep) == 0


def sharded_decisions(K, transports, name, total_per_rank, parts, thr=0.05):
    """kmd_correct_sharded on every virtual rank at once (one host thread each, as the collectives are matched
    calls): returns per rank (keep, global counters, n_kept, n_control, n_case)."""
    import ctypes as C
    import threading
    N = K._native
    lib = N.lib()
    ctype = K.CORRECTION_BY_NAME[name]
    out = [None] * len(parts)

    def work(r):
        p, s = parts[r]
        n = len(p)
        dp, ds = K.DeviceBuffer.from_host(p), K.DeviceBuffer.from_host(s)
        keep = K.DeviceBuffer(max(n, 1))
        local = np.zeros(N.NCOUNTERS, dtype=np.uint64)
        local[0], local[1] = total_per_rank[r], n
        g = np.zeros(N.NCOUNTERS, dtype=np.uint64)
        nk, nc, nca = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        rc = lib.kmd_correct_sharded(C.byref(transports[r]), ctype, thr, local.ctypes.data, g.ctypes.data, dp.ptr if n else None,
                                     ds.ptr if n else None, n, keep.ptr, C.byref(nk), C.byref(nc), C.byref(nca), None)
        out[r] = (rc, keep.to_host(np.uint8, n), g, int(nk.value), int(nc.value), int(nca.value))
    th = [threading.Thread(target=work, args=(r,)) for r in range(len(parts))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert all(o is not None and o[0] == 0 for o in out), [o and o[0] for o in out]
    return out


@pytest.mark.parametrize("name", ["benjamini", "holm", "bonferroni", "sidak", "nothing"])
@pytest.mark.parametrize("n_ranks", [2, 3, 8])
def test_sharded_correction_equals_single_list(K, name, n_ranks):
    """kmd_correct_sharded (kmd_shard.hip) over virtual ranks of one GPU -- one host thread per rank, the in-process
    transport (kmd_transport_local_create) as the wire: the decisions of kmd_correct over the whole list, for every
    corrector; ties across ranks, a rank without survivors, an early and a late stopping point."""
    import ctypes as C
    N = K._native
    n, nc, nk = 400_000, 4, 4
    mat = K.synth_matrix(SEED, 14, n, nc, nk, 4, K.LAYOUT_TILED)
    tot = K.column_sums(mat)
    model = K.PoissonLikelihood(nc, nk, tot[:nc], tot[nc:], 10000)
    acc = K.SurvivorAccumulator(n)
    K.diff_observer(model, acc, 2e-3).process(mat)
    ns = acc.finish()
    assert ns > 500
    got = acc.get()
    p_all, s_all = got["pvalue"].copy(), got["sign"].copy()
    p_all[5] = p_all[900]                                   # ties across ranks
    cuts = [0] + sorted(np.random.default_rng(n_ranks).choice(np.arange(1, ns), n_ranks - 1, replace=False).tolist()) + [ns]
    if n_ranks == 3:
        cuts[2] = cuts[1]                                   # rank 1 has no survivors
    parts = [(p_all[a:b], s_all[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    T = (N.Transport * n_ranks)()
    N.check(N.lib().kmd_transport_local_create(n_ranks, T), "kmd_transport_local_create")
    try:
        for total in (n, 40 * n):                            # one late, one early stopping point
            want, w_ctrl, w_case = K.aggregate(name, 0.05, total, K.DeviceBuffer.from_host(p_all), K.DeviceBuffer.from_host(s_all), ns)
            per = [total // n_ranks + (1 if r < total % n_ranks else 0) for r in range(n_ranks)]
            res = sharded_decisions(K, T, name, per, parts)
            keep = np.concatenate([o[1] for o in res])
            assert keep.tolist() == want.tolist()
            for r, o in enumerate(res):
                assert int(o[2][0]) == total and int(o[2][1]) == ns                    # the counters' sums, on every rank
                assert o[3] == int(o[1].sum()) and o[4] == int(((parts[r][1] == 0) & (o[1] == 1)).sum()) and o[5] == o[3] - o[4]
            assert sum(o[4] for o in res) == w_ctrl and sum(o[5] for o in res) == w_case
    finally:
        N.lib().kmd_transport_local_destroy(n_ranks, T)


@pytest.mark.parametrize("how", ["abort_before", "bad_argument", "leaves_mid_way"])
def test_sharded_correction_a_rank_that_fails_does_not_hang_the_others(K, how):
    """The in-process transport's barrier is abortable (a rank that failed outside a collective left the
    others in local_hub::barrier for ever, `kmdiff-hip diff --devices N` hung in join).  One of four virtual ranks
    gives up -- before its first collective (kmd_transport_abort: what the CLI's rank thread does when it throws), on
    an argument kmd_correct_sharded refuses (the entry point tells the others itself), or after the counters'
    all-reduce -- and every other rank comes back with an error within seconds."""
    import ctypes as C
    import threading
    import time
    N = K._native
    lib = N.lib()
    n_ranks, bad = 4, 2
    rng = np.random.default_rng(77)
    T = (N.Transport * n_ranks)()
    N.check(lib.kmd_transport_local_create(n_ranks, T), "create")
    rcs, msgs = [None] * n_ranks, [None] * n_ranks

    def work(r):
        p = np.sort(rng.uniform(0, 1e-6, 2000)) ** 2
        s = np.zeros(len(p), dtype=np.int32)
        dp, ds, keep = K.DeviceBuffer.from_host(p), K.DeviceBuffer.from_host(s), K.DeviceBuffer(len(p))
        local = np.zeros(N.NCOUNTERS, dtype=np.uint64)
        local[0], local[1] = 10**8, len(p)
        g = np.zeros(N.NCOUNTERS, dtype=np.uint64)
        nk, nc, nca = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        if r == bad and how == "abort_before":
            time.sleep(0.3)                                  # (the others are inside the all-reduce by now)
            rcs[r] = lib.kmd_transport_abort(C.byref(T[r]))
            return
        corr = K.CORRECTION_BY_NAME["benjamini"]
        if r == bad and how == "bad_argument":
            time.sleep(0.3)
            corr = 99                                        # refused before any collective
        if r == bad and how == "leaves_mid_way":
            # the counters' all-reduce by hand, then nothing more
            buf = K.DeviceBuffer.from_host(local)
            assert T[r].allreduce_u64(T[r].ctx, buf.ptr, N.NCOUNTERS, None) == 0
            time.sleep(0.3)
            rcs[r] = lib.kmd_transport_abort(C.byref(T[r]))
            return
        rcs[r] = lib.kmd_correct_sharded(C.byref(T[r]), corr, 0.05, local.ctypes.data, g.ctypes.data, dp.ptr, ds.ptr, len(p), keep.ptr,
                                         C.byref(nk), C.byref(nc), C.byref(nca), None)
        msgs[r] = lib.kmd_last_error().decode()
    th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(n_ranks)]
    t0 = time.time()
    [t.start() for t in th]
    [t.join(timeout=60) for t in th]
    assert not any(t.is_alive() for t in th), "a rank is still waiting for the one that gave up"
    assert time.time() - t0 < 30
    for r in range(n_ranks):
        if r == bad:
            assert rcs[r] == (N.KMD_E_INVALID if how == "bad_argument" else 0)
        else:
            assert rcs[r] not in (None, 0) and "another rank gave up" in msgs[r], (r, rcs[r], msgs[r])
    lib.kmd_transport_local_destroy(n_ranks, T)


def test_sharded_correction_one_rank_and_no_transport(K):
    """world = 1 (a NULL transport, or a one-rank one): kmd_correct itself."""
    import ctypes as C
    N = K._native
    rng = np.random.default_rng(5)
    p = np.sort(rng.uniform(0, 1e-6, 3000)) ** 2
    s = rng.integers(0, 3, 3000).astype(np.int32)
    want, _, _ = K.aggregate("benjamini", 0.05, 10**9, K.DeviceBuffer.from_host(p), K.DeviceBuffer.from_host(s), len(p))
    T = (N.Transport * 1)()
    N.check(N.lib().kmd_transport_local_create(1, T), "create")
    res = sharded_decisions(K, T, "benjamini", [10**9], [(p, s)])
    assert res[0][1].tolist() == want.tolist()
    N.lib().kmd_transport_local_destroy(1, T)
    dp, ds, keep = K.DeviceBuffer.from_host(p), K.DeviceBuffer.from_host(s), K.DeviceBuffer(len(p))
    local = np.zeros(N.NCOUNTERS, dtype=np.uint64)
    local[0] = 10**9
    nk = C.c_uint64(0)
    N.check(N.lib().kmd_correct_sharded(None, N.CORR_BENJAMINI, 0.05, local.ctypes.data, None, dp.ptr, ds.ptr, len(p), keep.ptr, C.byref(nk), None, None, None))
    assert keep.to_host(np.uint8, len(p)).tolist() == want.tolist() and nk.value == int(want.sum())


def test_sharded_correction_over_rccl_one_rank(K):
    """libkmdiff_hip_rccl.so: a communicator of ONE rank on this GPU (ncclCommInitRank) carries the collectives of
    kmd_correct_sharded -- what a one-GPU box can exercise of the RCCL transport (the two-rank form: tests/
    test_gpu_bench.py, skipped without a second GPU)."""
    import ctypes as C
    N = K._native
    R = N.rccl_lib()
    ident = C.create_string_buffer(128)
    assert R.kmd_rccl_unique_id(ident) == 0, R.kmd_rccl_last_error()
    t = N.Transport()
    assert R.kmd_transport_rccl_init(C.byref(t), 1, 0, ident) == 0, R.kmd_rccl_last_error()
    try:
        assert (t.rank, t.world) == (0, 1)
        # the two collectives themselves (world 1: the buffers come back as they went)
        a = K.DeviceBuffer.from_host(np.arange(16, dtype=np.uint64))
        assert t.allreduce_u64(t.ctx, a.ptr, 16, None) == 0
        assert a.to_host(np.uint64, 16).tolist() == list(range(16))
        b = K.DeviceBuffer(128)
        assert t.allgather(t.ctx, a.ptr, b.ptr, 128, None) == 0
        assert b.to_host(np.uint64, 16).tolist() == list(range(16))
        rng = np.random.default_rng(6)
        p = np.sort(rng.uniform(0, 1e-6, 2000)) ** 2
        s = rng.integers(0, 3, 2000).astype(np.int32)
        want, _, _ = K.aggregate("holm", 0.05, 10**8, K.DeviceBuffer.from_host(p), K.DeviceBuffer.from_host(s), len(p))
        res = sharded_decisions(K, [t], "holm", [10**8], [(p, s)])
        assert res[0][1].tolist() == want.tolist()
    finally:
        R.kmd_transport_rccl_destroy(C.byref(t))


def test_correction_golden_streams(K, golden_dir):
    """The reference's own corrector decisions (golden) on ascending streams: for BH/Holm the
    kept set is the prefix before the first rejection; for the stateless ones apply()."""
    with open(os.path.join(golden_dir, "correctors.json")) as f:
        g = json.load(f)
    for case in g["cases"]:
        p = np.array([fh(x) for x in case["p"]])
        buf = K.DeviceBuffer.from_host(p)
        keep, _, _ = K.aggregate(case["type"], fh(case["threshold"]), case["total"], buf, None, len(p))
        dec = np.array(case["apply"], dtype=np.uint8)
        if case["name"] in ("benjamini", "holm"):
            first = int(np.argmin(dec)) if (dec == 0).any() else len(dec)
            want = np.zeros(len(dec), dtype=np.uint8)
            want[:first] = 1
        else:
            want = dec
        assert keep.tolist() == want.tolist(), case["name"]


def test_gather_counts_of_survivors(K, oracle):
    import ctypes as C
    n, nc, nk = 50_000, 5, 5
    for layout in (K.LAYOUT_SOA, K.LAYOUT_ROWS, K.LAYOUT_TILED):
        mat = K.synth_matrix(SEED, 8, n, nc, nk, 2, layout, row0=777)
        host, _, _ = oracle.synth_rows(SEED, 8, 777, n, nc, nk, 2)
        tcs, tks = totals_of(host, nc)
        _, acc, ns = run_filter(K, mat, nc, nk, tcs, tks, 10000, 1e-4)
        assert ns > 0
        out = K.DeviceBuffer(ns * 10 * 8)
        t = mat.tile()
        K._native.check(K._native.lib().kmd_survivors_gather_counts(C.byref(t), 10, acc.bufs["row"].ptr, ns,
                                                                     out.ptr, None))
        K._native.check(K._native.lib().kmd_stream_sync(None))
        rows = acc.get()["row"].astype(np.int64) - 777
        assert (out.to_host(np.float64, ns * 10).reshape(ns, 10) == host[rows].astype(np.float64)).all()


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", ["fast", "sort"])
@pytest.mark.parametrize("layout_name", ["tiled", "soa", "rows"])
@pytest.mark.parametrize("count_bytes", [4, 2, 1])
def test_merge_partition_matches_oracle(K, oracle, layout_name, count_bytes, path, monkeypatch):
    """km::KmerMerger as driven at merge.hpp:265-289: device merge == oracle merge, through
    both device implementations (the tile merge + matrix fill, sort-based)."""
    monkeypatch.setenv("KMD_MERGE_PATH", path)
    rng = np.random.default_rng(17)
    universe = np.unique(rng.integers(0, 1 << 62, 60000, dtype=np.uint64))
    S = 9
    streams = []
    for s in range(S):
        pick = rng.random(len(universe)) < (0.05 + 0.1 * s)
        cnt = rng.integers(1, 70000 if s % 2 else 200, pick.sum()).astype(np.uint32)
        streams.append((universe[pick], cnt))
    streams[4] = (np.zeros(0, np.uint64), np.zeros(0, np.uint32))            # a sample with no k-mer here
    want, kmers = oracle.merge_partition(streams)
    m = K.merge_partition(streams, count_bytes=count_bytes, layout=layout_of(K, layout_name))
    assert m.n_rows == want.shape[0]
    cmax = {1: 255, 2: 65535, 4: 2 ** 32 - 1}[count_bytes]
    assert (m.to_host() == np.minimum(want, cmax)).all()
    assert (m.kmers_to_host()[0] == kmers).all()
    # empty partition
    e = K.merge_partition([(np.zeros(0, np.uint64), np.zeros(0, np.uint32))] * 3)
    assert e.n_rows == 0


@pytest.mark.parametrize("S", [32, 33, 64, 65, 104, 105, 128, 129, 256])
@pytest.mark.parametrize("presence", [0.6, 0.05])
def test_merge_bucket_capacity_boundaries(K, oracle, S, presence, monkeypatch):
    """The bucketed merge picks its kernel by sample count (256 / 512 / 1024 records per bucket from
    33 / 105 samples on): both sides of every switch, with rows of many records (a k-mer in 60 % of the
    samples) and of few (5 %: a bucket then holds dozens to hundreds of rows -- ranking over several
    key registers per lane, a row block larger than the LDS tile), forced onto the bucketed path."""
    monkeypatch.setenv("KMD_MERGE_PATH", "fast-only")
    rng = np.random.default_rng(1000 + S)
    universe = np.unique(rng.integers(0, 1 << 62, int(140_000 / (S * presence)) + 2000, dtype=np.uint64))
    streams = []
    for s in range(S):
        pick = rng.random(len(universe)) < presence
        streams.append((universe[pick], rng.integers(1, 5000, int(pick.sum())).astype(np.uint32)))
    want, kmers = oracle.merge_partition(streams)
    m = K.merge_partition(streams, count_bytes=4, layout=K.LAYOUT_TILED)
    assert m.n_rows == want.shape[0]
    assert (m.kmers_to_host()[0] == kmers).all()
    assert (m.to_host() == want).all()


@pytest.mark.parametrize("S,nc,presence", [(9, 4, 0.5), (40, 20, 0.65), (40, 20, 0.05), (33, 1, 0.3), (64, 63, 0.2), (105, 50, 0.4),
                                           (200, 100, 0.1), (256, 128, 0.02)])
def test_merge_sums_equals_merge_then_sum(K, oracle, S, nc, presence):
    """kmd_merge_sums + kmd_poisson_filter_sums (no matrix: every distinct k-mer leaves the merge as
    its control and case count sums) against the oracle merge: same k-mers, same two sums for every
    row -- with counts that take the sums past 2^32 --, and the same survivors (k-mer, sign, means
    bit-exact, p within 1e-10) as the oracle's diff_partition on the merged matrix; the all-ones
    k-mer and a sample without k-mers included."""
    rng = np.random.default_rng(4000 + S)
    universe = np.unique(np.concatenate([rng.integers(0, 1 << 62, int(120_000 / (S * presence)) + 3000, dtype=np.uint64),
                                         np.array([0, 2 ** 64 - 1], dtype=np.uint64)]))
    picks = []
    for s in range(S):
        pick = rng.random(len(universe)) < presence
        pick[-1] = s % 3 == 0                                            # the all-ones k-mer in a third of the samples
        if s == 2:
            pick[:] = False                                              # a sample with no k-mer here
        picks.append(pick)
    for big in (True, False):       # huge counts: the sums only (the oracle's table fallback loops over the count sum)
        streams = [(universe[p], rng.integers(1, 2 ** 32 - 1 if (big and s % 5 == 0) else 300, int(p.sum())).astype(np.uint32))
                   for s, p in enumerate(picks)]
        want, kmers = oracle.merge_partition(streams)
        sums = K.merge_sums(streams, nc)
        km, sc, sk, entry = sums.to_host()
        assert len(km) == want.shape[0]
        order = np.argsort(km, kind="stable")
        assert (km[order] == kmers).all()
        assert (sc[order] == want[:, :nc].sum(axis=1, dtype=np.uint64)).all()
        assert (sk[order] == want[:, nc:].sum(axis=1, dtype=np.uint64)).all()
    tcs, tks = totals_of(want, nc)
    ref = oracle.diff_partition(want, OL.LAYOUT_ROWS, nc, S - nc, int(tcs.sum()), int(tks.sum()), oracle.lf_table(10000), 0.01)
    model = K.PoissonLikelihood(nc, S - nc, tcs, tks, 10000)
    acc = K.SurvivorAccumulator(max(sums.n_rows, 1))
    obs = K.diff_observer(model, acc, 0.01)
    obs.process_sums(sums)
    acc.finish(sort=False)
    got = acc.get()
    assert obs.total() == want.shape[0] and len(got["row"]) == len(ref["row"])
    km_all = sums.kmers.to_host(np.uint64, sums.n_rows)
    got_km = km_all[got["row"].astype(np.int64)]
    by_kmer = np.argsort(got_km, kind="stable")
    assert (got_km[by_kmer] == kmers[ref["row"].astype(np.int64)]).all()
    assert got["sign"][by_kmer].tolist() == ref["sign"].tolist()
    assert got["mean_control"][by_kmer].tolist() == ref["mean_control"].tolist() and got["mean_case"][by_kmer].tolist() == ref["mean_case"].tolist()
    assert np.allclose(got["pvalue"][by_kmer], ref["pvalue"], rtol=0, atol=1e-10)


def test_sums_path_survivor_counts_from_the_streams(K, oracle):
    """The count rows of the sums path's survivors (what pop-strat and --keep-tmp need of a KmerSign,
    merge.hpp:91-92), looked up in the per-sample streams, against the rows of the oracle's merged matrix."""
    rng = np.random.default_rng(77)
    S, nc = 12, 5
    universe = np.unique(rng.integers(0, 1 << 62, 40_000, dtype=np.uint64))
    streams = []
    for s in range(S):
        pick = rng.random(len(universe)) < 0.5
        streams.append((universe[pick], rng.integers(1, 400 if s < nc else 40, int(pick.sum())).astype(np.uint32)))
    want, kmers = oracle.merge_partition(streams)
    sums = K.merge_sums(streams, nc)
    tcs, tks = totals_of(want, nc)
    model = K.PoissonLikelihood(nc, S - nc, tcs, tks, 10000)
    acc = K.SurvivorAccumulator(max(sums.n_rows, 1))
    K.diff_observer(model, acc, 1e-3).process_sums(sums)
    n = acc.finish(sort=False)
    assert n > 50
    got = K.gather_counts_streams(sums, acc.bufs["row"], n)
    km_all = sums.kmers.to_host(np.uint64, sums.n_rows)
    surv_km = km_all[acc.get()["row"].astype(np.int64)]
    idx = np.searchsorted(kmers, surv_km)
    assert (kmers[idx] == surv_km).all()
    assert (got == want[idx].astype(np.float64)).all()


def test_merge_sums_tiny_and_empty_inputs(K, oracle):
    """A handful of records, one stream only, nothing at all."""
    e = K.merge_sums([(np.zeros(0, np.uint64), np.zeros(0, np.uint32))] * 3, 1)
    assert e.n_rows == 0
    streams = [(np.array([5, 9], np.uint64), np.array([2, 3], np.uint32)), (np.array([9], np.uint64), np.array([7], np.uint32)),
               (np.zeros(0, np.uint64), np.zeros(0, np.uint32))]
    km, sc, sk, _ = K.merge_sums(streams, 1).to_host()
    order = np.argsort(km)
    assert km[order].tolist() == [5, 9] and sc[order].tolist() == [2, 3] and sk[order].tolist() == [0, 7]
    one = [(np.arange(1, 2001, dtype=np.uint64) * np.uint64(977), np.full(2000, 4, np.uint32))]
    km, sc, sk, _ = K.merge_sums(one, 0).to_host()
    assert len(km) == 2000 and (np.sort(km) == one[0][0]).all() and (sc == 0).all() and (sk == 4).all()


def test_merge_partition_clustered_keys_and_extremes(K, oracle, monkeypatch):
    """Heavily clustered keys overflow a bucket of the LDS merge: it must hand over to the sort
    path and still be exact; the all-ones key (k = 32, GGG...G) is a legal k-mer."""
    rng = np.random.default_rng(31)
    cluster = np.unique(rng.integers(1 << 40, (1 << 40) + 300000, 200000, dtype=np.uint64))
    spread = np.unique(rng.integers(0, 1 << 63, 50000, dtype=np.uint64))
    universe = np.unique(np.concatenate([cluster, spread, np.array([0, 2 ** 64 - 1], dtype=np.uint64)]))
    streams = []
    for s in range(5):
        pick = rng.random(len(universe)) < 0.5
        pick[-1] = s in (1, 3)                                    # the all-ones key in two samples
        streams.append((universe[pick], rng.integers(1, 9, pick.sum()).astype(np.uint32)))
    want, kmers = oracle.merge_partition(streams)
    for path in ("fast", "sort"):
        monkeypatch.setenv("KMD_MERGE_PATH", path)
        m = K.merge_partition(streams)
        assert m.n_rows == want.shape[0] and (m.to_host() == want).all() and (m.kmers_to_host()[0] == kmers).all()
    # spread keys only, including the extremes: served by the LDS merge itself
    uni2 = np.unique(np.concatenate([spread, np.array([0, 2 ** 64 - 1], dtype=np.uint64)]))
    st2 = [(uni2[rng.random(len(uni2)) < 0.6], None) for _ in range(4)]
    st2 = [(k, rng.integers(1, 9, len(k)).astype(np.uint32)) for k, _ in st2]
    st2[0] = (np.concatenate([st2[0][0][st2[0][0] != 2 ** 64 - 1], [np.uint64(2 ** 64 - 1)]]).astype(np.uint64),
              np.concatenate([st2[0][1][st2[0][0] != 2 ** 64 - 1], [7]]).astype(np.uint32))
    want2, k2 = oracle.merge_partition(st2)
    monkeypatch.setenv("KMD_MERGE_PATH", "fast")
    m2 = K.merge_partition(st2)
    assert m2.n_rows == want2.shape[0] and (m2.to_host() == want2).all() and (m2.kmers_to_host()[0] == k2).all()


@pytest.mark.parametrize("shape", ["random", "mixture"])
def test_merge_partition_bucket_refinement(K, oracle, shape, monkeypatch):
    """Random keys give Poisson-sized buckets and real partitions cluster: buckets over the wave
    capacity are cut again on the start table; "fast-only" makes the library report (instead of
    quietly sorting) if the bucketed merge did not serve the input."""
    monkeypatch.setenv("KMD_MERGE_PATH", "fast-only")
    rng = np.random.default_rng(41)
    if shape == "random":
        universe = np.unique(rng.integers(0, 1 << 62, 300_000, dtype=np.uint64))
    else:
        dense = [np.unique(rng.integers(c, c + (1 << 22), 30_000, dtype=np.uint64))
                 for c in (1 << 50, 3 << 59, (1 << 61) + 12345)]
        universe = np.unique(np.concatenate(dense + [rng.integers(0, 1 << 62, 250_000, dtype=np.uint64)]))
    S = 40
    streams = []
    for s in range(S):
        pick = rng.random(len(universe)) < 0.65
        streams.append((universe[pick], rng.integers(1, 300, pick.sum()).astype(np.uint32)))
    want, kmers = oracle.merge_partition(streams)
    m = K.merge_partition(streams, count_bytes=2, layout=K.LAYOUT_TILED)
    assert m.n_rows == want.shape[0]
    assert (m.kmers_to_host()[0] == kmers).all()
    assert (m.to_host() == want).all()


def test_merge_partition_two_limb_kmers(K, oracle):
    """32 < k <= 64: k-mers are (hi, lo) pairs compared as 128-bit numbers."""
    rng = np.random.default_rng(23)
    n = 30000
    hi = rng.integers(0, 50, n, dtype=np.uint64)                 # many equal high limbs
    lo = rng.integers(0, 1 << 63, n, dtype=np.uint64)
    order = np.lexsort((lo, hi))
    hi, lo = hi[order], lo[order]
    keep = np.ones(n, dtype=bool)
    keep[1:] = (hi[1:] != hi[:-1]) | (lo[1:] != lo[:-1])
    hi, lo = hi[keep], lo[keep]
    streams = []
    for s in range(6):
        pick = rng.random(len(lo)) < 0.4
        streams.append((lo[pick], rng.integers(1, 500, pick.sum()).astype(np.uint32), hi[pick]))
    want, wlo, whi = oracle.merge_partition2(streams)
    m = K.merge_partition(streams)
    assert m.n_rows == want.shape[0] and (m.to_host() == want).all()
    glo, ghi = m.kmers_to_host()
    assert (glo == wlo).all() and (ghi == whi).all()
    assert ((np.diff(whi.astype(np.int64)) > 0) | ((np.diff(whi.astype(np.int64)) == 0) & (wlo[1:] > wlo[:-1]))).all()


@pytest.mark.parametrize("S,hi_bits", [(6, 6), (40, 62), (70, 30)])
def test_merge_two_limb_bucketed_path(K, oracle, S, hi_bits, monkeypatch):
    """32 < k <= 64 through the bucketed LDS merge ("fast-only": no quiet sorting): buckets are cut
    on the top 64 bits of the (hi, lo) keys, the hash set compares full 128-bit keys -- including
    keys that share their top 64 bits and differ only below."""
    monkeypatch.setenv("KMD_MERGE_PATH", "fast-only")
    rng = np.random.default_rng(S)
    n = 120_000
    hi = rng.integers(0, 1 << hi_bits, n, dtype=np.uint64)
    lo = rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
    # families of keys equal in their top 64 bits: same hi, lo differing in the lowest bits only
    hi[:3000] = hi[3000:6000]
    lo[:3000] = lo[3000:6000] ^ rng.integers(1, 1 << min(hi_bits, 20), 3000, dtype=np.uint64)
    order = np.lexsort((lo, hi))
    hi, lo = hi[order], lo[order]
    keep = np.ones(n, dtype=bool)
    keep[1:] = (hi[1:] != hi[:-1]) | (lo[1:] != lo[:-1])
    hi, lo = hi[keep], lo[keep]
    streams = []
    for s in range(S):
        pick = rng.random(len(lo)) < 0.5
        streams.append((lo[pick], rng.integers(1, 70000, pick.sum()).astype(np.uint32), hi[pick]))
    streams[1] = (np.zeros(0, np.uint64), np.zeros(0, np.uint32), np.zeros(0, np.uint64))
    want, wlo, whi = oracle.merge_partition2(streams)
    m = K.merge_partition(streams, count_bytes=2, layout=K.LAYOUT_TILED)
    assert m.n_rows == want.shape[0]
    glo, ghi = m.kmers_to_host()
    assert (glo == wlo).all() and (ghi == whi).all()
    assert (m.to_host() == np.minimum(want, 65535)).all()


def test_merge_more_samples_than_the_bucketed_path_serves(K, oracle):
    """> 256 samples: the sort-based merge takes over (kMaxFastSamples); 300 streams, some empty."""
    rng = np.random.default_rng(300)
    universe = np.unique(rng.integers(0, 1 << 62, 20_000, dtype=np.uint64))
    streams = []
    for s in range(300):
        pick = rng.random(len(universe)) < (0.0 if s % 50 == 7 else 0.3)
        streams.append((universe[pick], rng.integers(1, 1000, pick.sum()).astype(np.uint32)))
    want, kmers = oracle.merge_partition(streams)
    m = K.merge_partition(streams, count_bytes=4, layout=K.LAYOUT_TILED)
    assert m.n_rows == want.shape[0] and (m.kmers_to_host()[0] == kmers).all() and (m.to_host() == want).all()


def test_reference_fixture_end_to_end(K, oracle):
    """tests/merge_test.cpp:12-46 on the device: the reference's 4-partition fixture through
    merge + Poisson filter: totals 160/160, 320 rows, 0 significant at 0.05/10000."""
    import test_kmtricks_fixture as F
    ids, parts, totals = F.load_fixture()
    model = K.PoissonLikelihood(1, 1, totals[:1], totals[1:], 10000)
    rows = sig = 0
    for streams in parts:
        m = K.merge_partition(streams)
        want, kmers = oracle.merge_partition(streams)
        assert (m.to_host() == want).all() and (m.kmers_to_host()[0] == kmers).all()
        acc = K.SurvivorAccumulator(max(m.n_rows, 1))
        obs = K.diff_observer(model, acc, 0.05 / 10000)
        obs.process(m)
        sig += acc.finish()
        rows += obs.total()
    assert (rows, sig) == (320, 0)


def test_merge_then_filter_equals_matrix_path(K, oracle):
    """Streams cut out of a synthetic matrix, merged on the device, give the survivors of the
    matrix itself (rows that are all-zero never exist in the generator)."""
    n, nc, nk = 60_000, 5, 4
    host, lo, _ = oracle.synth_rows(SEED, 21, 0, n, nc, nk, 4)
    streams = [(lo[host[:, s] > 0], host[host[:, s] > 0, s]) for s in range(nc + nk)]
    m = K.merge_partition(streams)
    assert m.n_rows == n and (m.to_host() == host).all()
    check_against_oracle(K, oracle, m, host, (lo, None), nc, nk, 10000, 1e-4)


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nc,nk,npc,stand", [(20, 20, 2, True), (12, 9, 2, False), (30, 34, 4, True), (100, 100, 2, True)])
def test_popstrat_retest_matches_oracle(K, oracle, nc, nk, npc, stand):
    """pop_strat_corrector::apply (popstrat.hpp:249-333) over the survivors of one partition:
    features bit-exact (host arithmetic), p-values to the same bar as stage 1."""
    S = nc + nk
    n = 120_000 if S <= 64 else 30_000
    rng = np.random.default_rng(11)
    Z = rng.normal(0, 0.1, size=(S, 10))
    Z[:nc, 0] += 0.05                          # a little structure correlated with phenotype
    mat = K.synth_matrix(SEED, 12, n, nc, nk, 4, K.LAYOUT_TILED)
    tot = K.column_sums(mat)
    model = K.PoissonLikelihood(nc, nk, tot[:nc], tot[nc:], 10000)
    acc = K.SurvivorAccumulator(n)
    K.diff_observer(model, acc, 1e-4).process(mat)
    ns = acc.finish()
    assert ns > 20
    pop = K.pop_strat_corrector(nc, nk, tot[:nc], tot[nc:], npc, Z, stand=stand)
    alt_d, null_d, null_like = pop.info()
    alt_o, null_o, totals_o, y = oracle.popstrat_setup(nc, nk, tot[:nc], tot[nc:], Z, npc, stand)
    assert alt_d.tolist() == alt_o.tolist()
    assert np.abs(null_d - null_o).max() <= 1e-9 * max(1.0, np.abs(null_o).max())
    counts = K.gather_counts(mat, acc.bufs["row"], ns)
    p_dev = pop.apply(counts, ns)
    host, _, _ = oracle.synth_rows(SEED, 12, 0, n, nc, nk, 4)
    rows = acc.get()["row"].astype(np.int64)
    p_ref = oracle.popstrat_pvalues(alt_o, null_o, totals_o, y, host[rows])
    assert np.abs(p_dev - p_ref).max() <= P_ABS_TOL
    nz = p_ref > 1e-300
    rel = np.abs(p_dev[nz] - p_ref[nz]) / p_ref[nz]
    assert rel.max(initial=0.0) <= 1e-7, rel.max()
    # sample-major input gives the same numbers
    cm = counts.to_host(np.float64, ns * S).reshape(ns, S).T.copy()
    p2 = pop.apply(K.DeviceBuffer.from_host(cm), ns, sample_major=True, ld=ns)
    assert (p2 == p_dev).all()


# ---------------------------------------------------------------------------------------------
def test_full_size_config2_properties(K, oracle):
    """BASELINE.json configs[1]: 10^8 rows, 4v4, k=31, one partition resident in HBM.
    Size-independent properties + exact oracle replay of sampled windows."""
    n, nc, nk = 100_000_000, 4, 4
    mat = K.synth_matrix(SEED, 0, n, nc, nk, 4, K.LAYOUT_SOA)
    tot = K.column_sums(mat)
    model = K.PoissonLikelihood(nc, nk, tot[:nc], tot[nc:], 10000)
    cap = n // 500
    acc = K.SurvivorAccumulator(cap)
    obs = K.diff_observer(model, acc, THR)
    obs.process(mat)
    ns = acc.finish()
    c = acc.read_counters()
    assert int(c[0]) == n and int(c[1]) == ns and int(c[2]) + int(c[3]) == ns
    assert int(c[6]) == 0                                                 # no p-value within 1e-8 of the threshold
    got = acc.get()
    assert (np.diff(got["row"].astype(np.int64)) > 0).all()               # ascending, no duplicates
    assert (np.diff(got["kmer_lo"].astype(np.int64)) > 0).all()           # ascending k-mers
    assert (got["pvalue"] <= THR).all()
    assert 1e-5 < ns / n < 1e-3
    # idempotence: a second pass reproduces the same survivors bit for bit
    acc2 = K.SurvivorAccumulator(cap)
    K.diff_observer(model, acc2, THR).process(mat)
    acc2.finish()
    g2 = acc2.get()
    for k in ("row", "kmer_lo", "pvalue", "sign", "mean_control", "mean_case"):
        assert (got[k] == g2[k]).all(), k
    # exact replay of sampled windows with the oracle
    rng = np.random.default_rng(7)
    lf = oracle.lf_table(10000)
    tc, tk = int(tot[:nc].sum()), int(tot[nc:].sum())
    starts = np.concatenate([[0, n - 4096], rng.integers(0, n - 4096, 30)])
    srows = got["row"].astype(np.int64)
    for s in starts:
        host, lo, _ = oracle.synth_rows(SEED, 0, int(s), 4096, nc, nk, 4)
        want = oracle.diff_partition(host, OL.LAYOUT_ROWS, nc, nk, tc, tk, lf, THR)
        sel = (srows >= s) & (srows < s + 4096)
        assert (srows[sel] - s).tolist() == want["row"].tolist()
        assert got["sign"][sel].tolist() == want["sign"].tolist()
        assert got["kmer_lo"][sel].tolist() == lo[want["row"].astype(np.int64)].tolist()
        assert_p_close(got["pvalue"][sel], want["pvalue"])


def replay_windows(K, oracle, part, n, nc, nk, limbs, tot, got, starts, thr=THR, width=2048):
    """Exact oracle replay of row windows of a synthetic partition against the device survivors."""
    lf = oracle.lf_table(10000)
    tc, tk = int(tot[:nc].sum()), int(tot[nc:].sum())
    srows = got["row"].astype(np.int64)
    for s in starts:
        host, lo, hi = oracle.synth_rows(SEED, part, int(s), width, nc, nk, 4, kmer_limbs=limbs)
        want = oracle.diff_partition(host, OL.LAYOUT_ROWS, nc, nk, tc, tk, lf, thr)
        sel = (srows >= s) & (srows < s + width)
        assert (srows[sel] - s).tolist() == want["row"].tolist()
        assert got["sign"][sel].tolist() == want["sign"].tolist()
        idx = want["row"].astype(np.int64)
        assert got["kmer_lo"][sel].tolist() == lo[idx].tolist()
        if limbs == 2:
            assert got["kmer_hi"][sel].tolist() == hi[idx].tolist()
        assert got["mean_control"][sel].tolist() == want["mean_control"].tolist()
        assert_p_close(got["pvalue"][sel], want["pvalue"])


def test_full_size_config3_256_partitions(K, oracle):
    """BASELINE.json configs[2]: 256 partitions x 39 062 500 rows (10^10 rows), 20v20, k=31,
    streamed one partition at a time through one GPU.  Whole-job invariants (a checksum of the
    per-partition checksums) + exact oracle replay of windows in sampled partitions."""
    nc, nk, rows, parts = 20, 20, 39_062_500, 256
    lib = K._native.lib()
    mat = K.CountMatrix(rows, nc + nk, 4, K.LAYOUT_TILED, with_kmers=True)
    # totals from a sample of partitions (the model only needs the same constants on both sides)
    totb = K.DeviceBuffer((nc + nk) * 8).zero()
    for p in (0, 100, 255):
        K._native.check(lib.kmd_synth_fill(SEED, p, 0, rows, nc, nk, 4, K.LAYOUT_TILED, mat.ld, mat.counts.ptr,
                                           mat.kmer_lo.ptr, None, None))
        K.column_sums(mat, totb)
    K._native.check(lib.kmd_stream_sync(None))
    tot = totb.to_host(np.uint64, nc + nk)
    model = K.PoissonLikelihood(nc, nk, tot[:nc], tot[nc:], 10000)
    rng = np.random.default_rng(99)
    check_parts = {0, 255} | set(int(x) for x in rng.integers(1, 255, 4))
    total = n_sig = n_ctrl = n_case = n_near = 0
    per_part = []
    for p in range(parts):
        K._native.check(lib.kmd_synth_fill(SEED, p, 0, rows, nc, nk, 4, K.LAYOUT_TILED, mat.ld, mat.counts.ptr,
                                           mat.kmer_lo.ptr, None, None))
        acc = K.SurvivorAccumulator(rows // 500)
        K.diff_observer(model, acc, THR).process(mat)
        c = acc.read_counters()
        assert int(c[0]) == rows and int(c[1]) == int(c[2]) + int(c[3]) and int(c[1]) < rows // 500
        n_near += int(c[6])
        total += int(c[0]); n_sig += int(c[1]); n_ctrl += int(c[2]); n_case += int(c[3])
        per_part.append(int(c[1]))
        if p in check_parts:
            acc.finish()
            got = acc.get()
            assert (np.diff(got["kmer_lo"].astype(np.int64)) > 0).all()
            assert int(got["kmer_lo"].min()) >> 54 == p and int(got["kmer_lo"].max()) >> 54 == p   # partition id bits
            starts = np.concatenate([[0, rows - 2048], rng.integers(0, rows - 2048, 6)])
            replay_windows(K, oracle, p, rows, nc, nk, 1, tot, got, starts)
    assert total == 10_000_000_000 and n_sig == n_ctrl + n_case == sum(per_part)
    assert n_near == 0            # KMD_CNT_NEAR_THRESHOLD: in 10^10 rows no p-value within 1e-8 of the threshold
    assert 0.5e-4 < n_sig / total < 5e-4
    assert max(per_part) < 1.5 * (n_sig / parts) and min(per_part) > 0.6 * (n_sig / parts)     # partitions look alike


def test_full_size_config4_k63_50v50(K, oracle):
    """BASELINE.json configs[3]: k = 63 (two 64-bit limbs), 50 + 50 samples, 4-byte counts."""
    nc, nk, rows = 50, 50, 16_000_000
    mat = K.synth_matrix(SEED, 3, rows, nc, nk, 4, K.LAYOUT_TILED, kmer_limbs=2)
    tot = K.column_sums(mat)
    model = K.PoissonLikelihood(nc, nk, tot[:nc], tot[nc:], 10000)
    acc = K.SurvivorAccumulator(rows // 200, kmer_limbs=2)
    obs = K.diff_observer(model, acc, THR)
    obs.process(mat)
    ns = acc.finish()
    c = acc.read_counters()
    assert int(c[0]) == rows and ns == int(c[2]) + int(c[3]) and ns > 100 and int(c[6]) == 0
    got = acc.get()
    assert (np.diff(got["kmer_hi"].astype(np.int64)) > 0).all()      # 128-bit k-mers ascending (hi limb strictly)
    assert int(c[5]) > 0                                             # 50-sample sums do leave the 10000-entry table
    rng = np.random.default_rng(4)
    replay_windows(K, oracle, 3, rows, nc, nk, 2, tot, got, np.concatenate([[0, rows - 2048], rng.integers(0, rows - 2048, 10)]))


def test_full_size_config5_popstrat_100v100(K, oracle):
    """BASELINE.json configs[4]: 100 + 100 samples, k = 31, population-stratification re-test on."""
    nc, nk, rows, npc = 100, 100, 4_000_000, 2
    mat = K.synth_matrix(SEED, 1, rows, nc, nk, 4, K.LAYOUT_TILED)
    tot = K.column_sums(mat)
    model = K.PoissonLikelihood(nc, nk, tot[:nc], tot[nc:], 10000)
    acc = K.SurvivorAccumulator(rows // 100)
    K.diff_observer(model, acc, THR).process(mat)
    ns = acc.finish()
    got = acc.get()
    assert ns > 100 and int(acc.read_counters()[6]) == 0
    rng = np.random.default_rng(8)
    Z = rng.normal(0, 0.1, size=(nc + nk, 10))
    pop = K.pop_strat_corrector(nc, nk, tot[:nc], tot[nc:], npc, Z)
    counts = K.gather_counts(mat, acc.bufs["row"], ns)
    p_dev = pop.apply(counts, ns)
    assert ((p_dev >= 0) & (p_dev <= 1)).all()
    # stage 1 windows + stage 2 on a sample of the survivors
    replay_windows(K, oracle, 1, rows, nc, nk, 1, tot, got, np.concatenate([[0], rng.integers(0, rows - 2048, 6)]))
    alt, null_model, totals_o, y = oracle.popstrat_setup(nc, nk, tot[:nc], tot[nc:], Z, npc, True)
    pick = rng.choice(ns, size=min(ns, 150), replace=False)
    rows_pick = got["row"].astype(np.int64)[pick]
    host = np.stack([oracle.synth_rows(SEED, 1, int(r), 1, nc, nk, 4)[0][0] for r in rows_pick])
    p_ref = oracle.popstrat_pvalues(alt, null_model, totals_o, y, host)
    assert np.abs(p_dev[pick] - p_ref).max() <= P_ABS_TOL
    nz = p_ref > 1e-300
    assert (np.abs(p_dev[pick][nz] - p_ref[nz]) / p_ref[nz]).max(initial=0.0) <= 1e-7
    # the re-tested p-values feed the corrector like the Poisson ones do
    keep, n_ctrl, n_case = K.aggregate("benjamini", 0.05, rows, pop.last_pvalues, acc.bufs["sign"], ns)
    assert keep.tolist() == oracle.aggregate(2, 0.05, rows, p_dev).tolist()
