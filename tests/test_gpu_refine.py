"""kmd_pvalues_refine: the survivors' p-values recomputed with correctly rounded log / exp, so that they carry the bits
a glibc-built reference gives them (PoissonLikelihood::process, include/kmdiff/model.hpp:142-176, with
LogFactorialTable::operator[], log_factorial_table.hpp:14-18 / log_factorial_table.cpp:13-22).
The bar of the other parity tests (1e-10 absolute / 1e-9 relative on p) is what the filters' own p-values meet in the
regimes those tests cover; tests/soak.py found its edge (1.1e-10 at p = 0.92 with count sums of ~7000; 4.7e-10 relative
with sums beyond the table).  After this pass the bar is: bit-equal to the oracle in >= 99 % of the records, and the rest
(where glibc's own log is one ulp off the rounded value) within 1e-9 relative / 1e-10 absolute (measured over 1.2 x 10^7 soak records: 2.5e-10 / 2.6e-11)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as OL
from test_gpu_parity import totals_of
from test_gpu_tilemerge import make_streams

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    import kmdiff_amd as K
    assert K.device_count() >= 1, "no GPU: the HIP path has no CPU fallback"
    return K


def refined_case(K, oracle, S, nc, n_kmers, count_hi, thr, lf_n, seed):
    rng = np.random.default_rng(seed)
    universe = np.unique(rng.integers(0, 1 << 62, n_kmers, dtype=np.uint64))
    streams = make_streams(rng, universe, S, np.clip(rng.uniform(0.3, 1.0, S), 0, 1), count_hi=count_hi)
    want, wlo = oracle.merge_partition(streams)
    tcs, tks = totals_of(want, nc)
    ref = oracle.diff_partition(want, OL.LAYOUT_ROWS, nc, S - nc, int(tcs.sum()), int(tks.sum()), oracle.lf_table(lf_n), thr)
    model = K.PoissonLikelihood(nc, S - nc, tcs, tks, lf_n)
    acc = K.SurvivorAccumulator(max(want.shape[0], 1))
    K.merge_filter(K.StreamSet(streams), K.diff_observer(model, acc, thr))
    n = acc.finish(by_kmer=True)
    raw = acc.get()["pvalue"].copy()
    assert acc.finish(by_kmer=True, refine=model) == n == len(ref["row"])
    got = acc.get()
    rr = ref["row"].astype(np.int64)
    assert got["kmer_lo"].tolist() == wlo[rr].tolist()                                   # nothing but p moved
    assert got["mean_control"].tolist() == ref["mean_control"].tolist() and got["sign"].tolist() == ref["sign"].tolist()
    sc, sk = want[rr, :nc].astype(np.uint64).sum(axis=1), want[rr, nc:].astype(np.uint64).sum(axis=1)
    return raw, got["pvalue"], ref["pvalue"], (sc < lf_n) & (sk < lf_n)


def bar(p, w):
    assert (p == w).mean() >= 0.99, (p == w).mean()
    d = np.abs(p - w)
    assert d.max() <= 1e-10
    nz = w > 0
    assert (d[nz] / w[nz]).max(initial=0.0) <= 1e-9
    assert (p[~nz] == 0).all()


@pytest.mark.parametrize("S,nc,count_hi,thr,lf_n", [(100, 43, 517, 1.0, 10000), (40, 20, 300, 1.0, 10000), (40, 20, 300, 1e-3, 10000),
                                                    (16, 8, 60, 0.3, 10000)])
def test_refined_pvalues_inside_the_table_are_the_oracles_bits(K, oracle, S, nc, count_hi, thr, lf_n):
    """The first case is the soak's: 100 samples of counts up to 516, threshold 1 -- every row survives, p up to 1, where the
    filter's own value is 1.1e-10 off."""
    raw, p, w, inside = refined_case(K, oracle, S, nc, 7000, count_hi, thr, lf_n, seed=100 + S)
    assert inside.sum() > 500
    bar(p[inside], w[inside])
    if (~inside).any():
        assert np.abs(p[~inside] - w[~inside]).max() <= 1e-10
    assert np.abs(raw - w).max() <= 5e-10 and (raw != w).any()                           # (what the pass is for)


@pytest.mark.parametrize("S,nc,count_hi,thr,lf_n", [(100, 43, 517, 1.0, 300), (40, 20, 300, 1.0, 50), (9, 4, 3000, 1e-2, 2000),
                                                    (40, 20, 3000, 1.0, 10000)])
def test_refined_pvalues_beyond_the_table_repeat_the_running_sum(K, oracle, S, nc, count_hi, thr, lf_n):
    """Sums beyond --log-factorial: LogFactorialTable::operator[] falls back to its k-term running sum
    (log_factorial_table.cpp:13-22); the pass repeats it term for term, 64 logarithms at a time."""
    raw, p, w, inside = refined_case(K, oracle, S, nc, 3000, count_hi, thr, lf_n, seed=200 + S)
    assert (~inside).sum() > 300
    bar(p[~inside], w[~inside])
    if inside.sum() >= 200:
        bar(p[inside], w[inside])


def test_refine_leaves_what_it_cannot_invert(K, oracle):
    """Records whose means are not those of a row (NaN, a fractional case sum, a control mean no integer sum gives) keep their
    p-value; sums of 2^20 and more get rounded logarithms but not the running sum; n = 0 and a zero sum are fine."""
    nc, nk = 3, 2
    tcs, tks = np.array([10 ** 6, 2 * 10 ** 6, 3 * 10 ** 6], dtype=np.uint64), np.array([4 * 10 ** 6, 10 ** 6], dtype=np.uint64)
    model = K.PoissonLikelihood(nc, nk, tcs, tks, 10000)
    dTc, dTk = float(tcs.sum()), float(tks.sum())
    sums = [(0, 7), (7, 0), (120, 31), (9999, 9999), (10000, 3), ((1 << 20) - 1, 5), (1 << 20, 5), (5, 1 << 21), (3_000_000_000, 12)]
    mc = np.array([float(c) * dTk / dTc for c, _ in sums] + [np.nan, 3.0 * dTk / dTc, 2.6, -1.0])
    mk = np.array([float(k) for _, k in sums] + [4.0, 4.5, 4.0, 4.0])
    p0 = np.full(len(mc), 0.125)
    dmc, dmk, dp = K.DeviceBuffer.from_host(mc), K.DeviceBuffer.from_host(mk), K.DeviceBuffer.from_host(p0)
    lib, check = K._native.lib(), K._native.check
    check(lib.kmd_pvalues_refine(model.handle, 0, None, None, None, None))
    check(lib.kmd_pvalues_refine(model.handle, len(mc), dmc.ptr, dmk.ptr, dp.ptr, None))
    check(lib.kmd_stream_sync(None))
    p = dp.to_host(np.float64, len(mc))
    rows = np.array([[c * (i == 0) for i in range(nc)] + [k * (i == 0) for i in range(nk)] for c, k in sums], dtype=np.uint32)
    w, _, _, _ = oracle.poisson_rows(rows, OL.LAYOUT_ROWS, nc, nk, int(tcs.sum()), int(tks.sum()), oracle.lf_table(10000))
    assert p[:6].tolist() == w[:6].tolist()
    # sums of 2^20 and more: Stirling's series for the table term, the logarithms rounded -- close, not the bits
    assert (p[6:9] != 0.125).all() and (np.abs(p[6:9] - w[6:9]) <= 2e-9 * w[6:9] + 1e-300).all()
    assert p[9:].tolist() == [0.125] * (len(mc) - 9)
    assert lib.kmd_pvalues_refine(model.handle, 3, None, dmk.ptr, dp.ptr, None) == -1            # KMD_E_INVALID


def test_refine_on_the_outputs_of_process(K, oracle):
    """The same pass over kmd_poisson_process's arrays (IModel::process, every row): all rows then carry the oracle's bits."""
    rng = np.random.default_rng(5)
    nc, nk, n = 6, 5, 20000
    rows = rng.integers(0, 400, (n, nc + nk)).astype(np.uint32)
    rows[rng.random(n) < 0.05] = 0
    tcs, tks = totals_of(rows, nc)
    model = K.PoissonLikelihood(nc, nk, tcs, tks, 10000)
    pv, sg, mc, mk = model.process(K.CountMatrix.from_host(rows, layout=K.LAYOUT_ROWS))
    out = {"pvalue": pv, "mean_control": mc, "mean_case": mk}
    w, ws, wmc, wmk = oracle.poisson_rows(rows, OL.LAYOUT_ROWS, nc, nk, int(tcs.sum()), int(tks.sum()), oracle.lf_table(10000))
    dmc, dmk, dp = K.DeviceBuffer.from_host(out["mean_control"]), K.DeviceBuffer.from_host(out["mean_case"]), K.DeviceBuffer.from_host(out["pvalue"])
    K._native.check(K._native.lib().kmd_pvalues_refine(model.handle, n, dmc.ptr, dmk.ptr, dp.ptr, None))
    K._native.check(K._native.lib().kmd_stream_sync(None))
    bar(dp.to_host(np.float64, n), w)


def test_running_sum_through_the_table_is_the_term_by_term_sum(K, oracle):
    """LogFactorialTable::log_factorial (log_factorial_table.cpp:13-22) on the device, two ways: 64 correctly rounded
    logarithms and 64 ordered additions per step (what k_resolve_near runs for the rare row near the threshold), and the
    fast one of kmd_pvalues_refine -- logarithms from a table, and, while the partial sums stay inside one binade, the 64
    roundings taken independently and summed as integers.  They must agree to the bit for every k; the oracle's own sum
    (glibc logarithms) agrees wherever glibc returned the rounded logarithm of every term that mattered."""
    rng = np.random.default_rng(11)
    ks = np.concatenate([np.arange(0, 700), rng.integers(2, 1 << 20, 1200), np.array([(1 << 20) - 1, 1 << 19, (1 << 19) + 1, 65535, 65536, 65537])]).astype(np.uint64)
    dk = K.DeviceBuffer.from_host(ks)
    da, db = K.DeviceBuffer(len(ks) * 8), K.DeviceBuffer(len(ks) * 8)
    lib, check = K._native.lib(), K._native.check
    check(lib.kmd_test_running_sums(dk.ptr, len(ks), da.ptr, db.ptr, None))
    check(lib.kmd_stream_sync(None))
    plain, fast = da.to_host(np.float64, len(ks)), db.to_host(np.float64, len(ks))
    assert plain.tolist() == fast.tolist()
    empty = np.zeros(0)
    small = ks < 60000                                                        # (the oracle's loop is O(k) on one core)
    want = np.array([oracle.lf_at(empty, int(k)) for k in ks[small]])
    assert (fast[small] == want).mean() >= 0.97
    assert np.abs(fast[small] - want).max() <= 2e-11 * max(1.0, want.max())
