"""Pop-strat re-test (K3, kmd_popstrat_apply) against the oracle's restatement of pop_strat_corrector::apply
(popstrat.hpp:249-333) and glm_irls (linear_model.cpp:297-410) at the edges the reference code has:
the iteration limit (break before the weights are copied), a singular and a NaN Hessian, rows that all
fail g_i > 1e-305, both likelihoods underflowing to zero, no standardisation, 2 .. 10 principal components,
--epsilon.  (The oracle's IRLS itself is held against an independent Newton-Raphson in test_irls_cases.py.)
Bars: |p_dev - p_ref| <= 1e-10 absolute and 1e-7 relative (FP64 pow / log / exp of ocml vs glibc)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    import kmdiff_amd as K
    assert K.device_count() >= 1, "no GPU: the HIP path has no CPU fallback"
    return K


def oracle_setup(oracle, nc, nk, tc, tk, Z, npc, stand, max_iter, Y):
    n, fn = nc + nk, 2 + npc
    nul, alt, tot = np.zeros((n, fn)), np.zeros((n, fn + 1)), np.zeros(n)
    oracle.L.kmdo_popstrat_features(nc, nk, tc.ctypes.data, tk.ctypes.data, Z.ctypes.data, Z.shape[1], npc, int(stand),
                                    nul.ctypes.data, alt.ctypes.data, tot.ctypes.data)
    w = np.zeros(fn)
    oracle.L.kmdo_glm_irls(nul.ctypes.data, Y.ctypes.data, n, fn, max_iter, w.ctypes.data, None, None)
    return alt, w, tot


def check(K, oracle, nc, nk, npc, stand, max_iter, rows, Z=None, Y=None, totals_scale=1.0, seed=3, want_spread=True, libm_rows=None, outside=None):
    """libm_rows: a list -- rows outside the bars are then allowed IF the oracle's own p-value for them moves by more than
    the bars when the pow() inside its sigmoid is jittered by one ulp (kmdo_sigmoid_jitter): their p-value is a property of
    libm's last bit, not of the data (an unstandardised design -- stand=False, which the reference cannot reach -- has a
    column of ~1e7 beside one of ~1e-6, and IRLS without pivoting on a Hessian of condition number > 1e16 amplifies one
    ulp to anything; the reference's own standardize() leaves the k-mer column unscaled too, so separation cases exist in
    the reachable mode as well).  Their indices are appended to the list.
    outside: a list (tests/soak.py's tally) -- a row outside the bars that the jitter does NOT explain is then appended as
    (index, p_dev, p_ref, jitter lo, jitter hi) instead of failing the check."""
    S = nc + nk
    rng = np.random.default_rng(seed)
    if Z is None:
        Z = rng.normal(0, 0.1, size=(S, 10))
        Z[:nc, 0] += 0.05
    Z = np.ascontiguousarray(Z, dtype=np.float64)
    tc = (rng.integers(8_000_000, 12_000_000, nc) * totals_scale).astype(np.uint64)
    tk = (rng.integers(8_000_000, 12_000_000, nk) * totals_scale).astype(np.uint64)
    Yv = np.concatenate([np.ones(nc), np.zeros(nk)]) if Y is None else np.ascontiguousarray(Y, dtype=np.float64)
    it = max_iter if max_iter > 0 else 100
    pop = K.pop_strat_corrector(nc, nk, tc, tk, npc, Z, Y=Yv, stand=stand, max_iter=max_iter)
    alt_o, null_o, tot_o = oracle_setup(oracle, nc, nk, tc, tk, Z, npc, stand, it, Yv)
    alt_d, null_d, _ = pop.info()
    assert alt_d.tolist() == alt_o.tolist()
    both_nan = np.isnan(null_d) & np.isnan(null_o)
    null_scale = max(1.0, np.abs(null_o[~both_nan]).max(initial=0.0))
    null_dev = np.abs(null_d - null_o)[~both_nan].max(initial=0.0)
    if null_dev > 1e-9 * null_scale:
        # the null model itself: allowed (libm_rows given) if the oracle's own fit moves as much under one ulp of jitter;
        # the design is then marked (-1) and its rows are not compared -- every one of them inherits the null fit's doubt
        assert libm_rows is not None, null_dev
        spread = 0.0
        for sd in range(1, 9):
            oracle.L.kmdo_sigmoid_jitter(sd * 7919, 1)
            try:
                _, n2, _ = oracle_setup(oracle, nc, nk, tc, tk, Z, npc, stand, it, Yv)
            finally:
                oracle.L.kmdo_sigmoid_jitter(0, 0)
            with np.errstate(invalid="ignore"):
                spread = max(spread, float(np.nanmax(np.abs(n2 - null_o))))
        if not (null_dev <= 8 * spread + 1e-9 * null_scale) and outside is not None:
            outside.append((-1, float(null_dev), float(null_scale), float(spread), 0.0))      # (tally: the null fit itself, unexplained)
            libm_rows.append(-1)
            return None
        assert null_dev <= 8 * spread + 1e-9 * null_scale, (null_dev, spread)
        libm_rows.append(-1)
        return None
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    n = len(rows)
    p_dev = pop.apply(K.DeviceBuffer.from_host(rows), n)

    def ref_of(r):
        return oracle.L.kmdo_popstrat_pvalue(alt_o.ctypes.data, S, alt_o.shape[1], Yv.ctypes.data, tot_o.ctypes.data,
                                             r.ctypes.data, null_o.ctypes.data, it)
    p_ref = np.array([ref_of(r) for r in rows])
    fin_d, fin_r = np.isfinite(p_dev), np.isfinite(p_ref)
    with np.errstate(invalid="ignore"):
        d = np.abs(p_dev - p_ref)
        bad = (fin_d != fin_r) | (fin_r & fin_d & ((d > 1e-10) | (d > 1e-7 * np.abs(p_ref))))
    if libm_rows is None:
        assert np.isfinite(p_dev).all() == np.isfinite(p_ref).all()
        assert not bad.any(), (int(bad.sum()), np.nanmax(d))
    else:
        for i in np.nonzero(bad)[0]:
            seen = [p_ref[i]]
            for lg in (0, 1, 2):
                for sd in range(1, 17):
                    oracle.L.kmdo_sigmoid_jitter(sd * 104729 + lg, lg)
                    try:
                        seen.append(ref_of(rows[i]))
                    finally:
                        oracle.L.kmdo_sigmoid_jitter(0, 0)
            seen = np.array(seen)
            if np.isfinite(seen).all() and np.isfinite(p_dev[i]):
                lo, hi = seen.min(), seen.max()
                # the oracle itself is unsure: one ulp moves it by at least a quarter of what the device differs by,
                # and the device is no further from the oracle's range than a few times its width
                explained = hi - lo >= 0.25 * abs(p_dev[i] - p_ref[i]) and lo - 4 * (hi - lo) <= p_dev[i] <= hi + 4 * (hi - lo)
                if not explained and outside is not None:
                    outside.append((int(i), float(p_dev[i]), float(p_ref[i]), float(lo), float(hi)))
                    continue
                assert explained, (i, p_dev[i], p_ref[i], lo, hi)
            else:
                assert not np.isfinite(seen).all() or len(np.unique(seen)) > 1, (i, p_dev[i], seen[:4])
            libm_rows.append(int(i))
    ok = fin_r
    if want_spread:
        assert len(np.unique(p_ref[ok])) > min(10, n // 4)
    return p_ref


def count_rows(rng, n, nc, nk, effect=3.0):
    """count vectors of survivors: a case / control effect of random size and direction, plus noise"""
    S = nc + nk
    base = rng.gamma(2.0, 30.0, (n, 1))
    eff = np.where(rng.random((n, 1)) < 0.5, effect, 1.0 / effect)
    lam = np.repeat(base, S, axis=1)
    lam[:, nc:] *= eff
    return rng.poisson(lam).astype(np.float64)


@pytest.mark.parametrize("max_iter", [1, 2, 3, 4, 7])
def test_iteration_limit_drops_the_last_update(K, oracle, max_iter):
    """linear_model.cpp:386-395: the iteration that reaches max_iters leaves before `weight` is copied."""
    rng = np.random.default_rng(max_iter)
    # (limit 1: the weights stay at 1, every sigmoid saturates, both likelihoods are 0 -> one p-value for all)
    check(K, oracle, 12, 12, 2, True, max_iter, count_rows(rng, 300, 12, 12), want_spread=max_iter > 1)


@pytest.mark.parametrize("npc", [2, 3, 4, 5, 6, 7, 8, 9, 10])
@pytest.mark.parametrize("stand", [True, False])
def test_principal_components_and_standardisation(K, oracle, npc, stand):
    rng = np.random.default_rng(100 + npc)
    nc, nk = (20, 20) if npc < 8 else (35, 30)
    # unstandardised totals of 1e7 make the Hessian wildly scaled: the no-pivot LU still has to agree
    check(K, oracle, nc, nk, npc, stand, 0, count_rows(rng, 400, nc, nk), want_spread=stand)


def test_more_features_than_samples(K, oracle):
    """Fewer samples than columns of the design (6 samples, 8 PCs: 10 null features): standardize() fills stddev -- sized
    by rows -- by column (popstrat.cpp:330,349), past its end in the reference; here and in the oracle the entries
    exist (tests/soak.py met glibc's heap check on such a design).  The Hessian is singular: every fit stops at once."""
    rng = np.random.default_rng(15)
    check(K, oracle, 3, 3, 8, True, 0, count_rows(rng, 100, 3, 3), want_spread=False)
    check(K, oracle, 2, 2, 10, True, 0, count_rows(rng, 100, 2, 2), want_spread=False)


def test_singular_hessian_leaves_the_weights_at_one(K, oracle):
    """A survivor whose count vector is all zero has a zero k-mer column: last pivot 0, det == 0
    (linear_model.cpp:182-186, 366-372) -> the fit stops before its first update."""
    rng = np.random.default_rng(5)
    rows = count_rows(rng, 64, 10, 10)
    rows[::3] = 0.0
    check(K, oracle, 10, 10, 2, True, 0, rows)


def test_nan_hessian(K, oracle):
    """A principal component that is zero for every sample, unstandardised: a zero pivot in the middle of the
    no-pivot LU, 0/0, det is NaN -> both the null fit and every survivor's fit stop at once."""
    rng = np.random.default_rng(6)
    Z = rng.normal(0, 0.1, size=(20, 10))
    Z[:, 0] = 0.0
    check(K, oracle, 10, 10, 2, False, 0, count_rows(rng, 100, 10, 10), Z=Z, want_spread=False)


def test_rows_that_all_fail_the_variance_filter(K, oracle):
    """g_i = mu_i (1 - mu_i) > 1e-305 fails for every sample (linear_model.cpp:336, 343) when the phenotypes
    put mu outside (0, 1) on entry: the fit returns its initial weights."""
    rng = np.random.default_rng(7)
    check(K, oracle, 8, 8, 2, True, 0, count_rows(rng, 50, 8, 8), Y=np.full(16, 2.0), want_spread=False)


def test_both_likelihoods_zero_fixup(K, oracle):
    """popstrat.hpp:312-316: null and alternative likelihood both underflow to 0 -> (0.001, 1).  Unstandardised
    totals of 1e9 under weights left at 1 (iteration limit 1) saturate every sigmoid."""
    rng = np.random.default_rng(8)
    p = check(K, oracle, 10, 10, 2, False, 1, count_rows(rng, 80, 10, 10), totals_scale=100.0, want_spread=False)
    # LLR = -2 ln(0.001 / 1) = 13.8155...: one and the same p-value for every survivor
    assert len(np.unique(p)) == 1 and p[0] == oracle.chisqc(1, -2.0 * np.log(0.001 / 1.0))


def test_perfect_separation(K, oracle):
    """Counts that separate cases from controls perfectly: the weights grow until the MSE stops changing."""
    rng = np.random.default_rng(9)
    rows = np.zeros((40, 20))
    rows[:, 10:] = rng.integers(200, 400, (40, 10))
    rows[:, :10] = rng.integers(0, 3, (40, 10))
    check(K, oracle, 10, 10, 2, True, 0, rows, want_spread=False)


def test_large_batch_matches_small_batches(K, oracle):
    """The answer of a survivor does not depend on who shares its wave or launch."""
    rng = np.random.default_rng(10)
    nc = nk = 20
    rows = count_rows(rng, 5000, nc, nk)
    Z = rng.normal(0, 0.1, size=(40, 10))
    tc = rng.integers(8_000_000, 12_000_000, nc).astype(np.uint64)
    tk = rng.integers(8_000_000, 12_000_000, nk).astype(np.uint64)
    pop = K.pop_strat_corrector(nc, nk, tc, tk, 2, Z)
    p_all = pop.apply(K.DeviceBuffer.from_host(rows), len(rows))
    for a, b in [(0, 1), (1, 64), (64, 129), (129, 1000), (1000, 5000)]:
        p = pop.apply(K.DeviceBuffer.from_host(rows[a:b]), b - a)
        assert (p == p_all[a:b]).all()


def test_epsilon_clamps_small_likelihood_ratios(K, oracle):
    """--epsilon (cli.cpp:336-339 -> set_params -> s_epsilon, popstrat.hpp:162-175): |LLR| below it counts as 0,
    i.e. p = 1 (popstrat.hpp:321-326); 0 keeps the default 1e-30."""
    rng = np.random.default_rng(12)
    nc = nk = 10
    rows = count_rows(rng, 300, nc, nk, effect=1.05)                 # hardly any effect: small likelihood ratios
    Z = rng.normal(0, 0.1, size=(20, 10))
    tc = rng.integers(8_000_000, 12_000_000, nc).astype(np.uint64)
    tk = rng.integers(8_000_000, 12_000_000, nk).astype(np.uint64)
    buf = K.DeviceBuffer.from_host(rows)
    p0 = K.pop_strat_corrector(nc, nk, tc, tk, 2, Z).apply(buf, len(rows))
    p_same = K.pop_strat_corrector(nc, nk, tc, tk, 2, Z, epsilon=0.0).apply(buf, len(rows))
    p_big = K.pop_strat_corrector(nc, nk, tc, tk, 2, Z, epsilon=0.5).apply(buf, len(rows))
    assert (p0 == p_same).all()
    llr_small = p0 > oracle.chisqc(1, 0.5)                            # LLR < 0.5  <=>  p above the tail at 0.5
    assert llr_small.any() and (~llr_small).any()
    assert (p_big[llr_small] == 1.0).all() and (p_big[~llr_small] == p0[~llr_small]).all()


@pytest.mark.parametrize("nc,nk,npc,stand,max_iter", [(10, 10, 2, True, 0), (20, 20, 2, False, 0), (33, 31, 4, True, 3), (50, 50, 5, True, 0),
                                                      (100, 100, 2, True, 0), (40, 40, 10, True, 0), (6, 5, 2, True, 1)])
def test_group_kernel_equals_lane_kernel(K, oracle, monkeypatch, nc, nk, npc, stand, max_iter):
    """The two K3 kernels -- one lane per survivor, a group of lanes per survivor (small batches, many features) --
    perform the same operations in the same order: bit-identical p-values, both layouts of the counts; and both
    meet the oracle (every other test of this file runs under whichever kernel the batch size selects)."""
    rng = np.random.default_rng(nc * 100 + npc)
    S = nc + nk
    rows = count_rows(rng, 700, nc, nk)
    rows[5] = 0.0                                                      # a singular fit among them
    Z = rng.normal(0, 0.1, size=(S, 10))
    tc = rng.integers(8_000_000, 12_000_000, nc).astype(np.uint64)
    tk = rng.integers(8_000_000, 12_000_000, nk).astype(np.uint64)
    pop = K.pop_strat_corrector(nc, nk, tc, tk, npc, Z, stand=stand, max_iter=max_iter)
    buf = K.DeviceBuffer.from_host(rows)
    buf_t = K.DeviceBuffer.from_host(np.ascontiguousarray(rows.T))
    out = {}
    for kind in ("lane", "group"):
        monkeypatch.setenv("KMD_POPSTRAT_KERNEL", kind)
        out[kind] = pop.apply(buf, len(rows))
        assert (pop.apply(buf_t, len(rows), sample_major=True, ld=len(rows)) == out[kind]).all()
    assert (out["lane"] == out["group"]).all()
    assert len(np.unique(out["lane"])) > 50 or max_iter == 1


@pytest.mark.parametrize("nc,nk,npc,seed", [(43, 3, 4, 1), (43, 3, 4, 2), (50, 2, 3, 3), (12, 40, 6, 4)])
def test_rows_whose_p_value_hangs_on_the_last_bit_of_pow(K, oracle, nc, nk, npc, seed):
    """A STRESS case, not a mode the reference can reach: the unstandardised design (stand=False) with a few samples on
    one side.  (In the reference s_stand starts true and set_params can only turn it on, popstrat.hpp:155,174-175: every
    reference run standardises -- with standardize()'s bugs, which kmd_popstrat_create repeats; the CLI ignores --stand
    accordingly.  Round 4 called this design "the default": wrong; PARITY.md 2 has the tally of the
    reachable mode alone.)  tests/soak.py found it: one row in ~3000 came out at p = 5e-5 on the device and p = 1 in the
    oracle -- and the oracle itself gives either, depending on one ulp of the pow() in its sigmoid (kmdo_sigmoid_jitter).
    Such rows are no parity failure and no parity success: they are counted.  Every other row keeps the bars (1e-10
    absolute, 1e-7 relative), and there are few of the former."""
    rng = np.random.default_rng(9000 + seed)
    rows = count_rows(rng, 3000, nc, nk, effect=float(rng.choice([1.2, 3.0])))
    Z = rng.normal(0, 0.1, size=(nc + nk, 10))
    libm_rows = []
    check(K, oracle, nc, nk, npc, False, 0, rows, Z=Z, seed=seed, want_spread=False, libm_rows=libm_rows)
    assert len(libm_rows) <= 30, len(libm_rows)


def test_reference_linear_vectors_through_the_device_routines(K):
    """The only numbers of stage 2 the reference's own tests hold (tests/linear_test.cpp:29-31 sigmoid(1) and predict,
    :80-151 the 4 x 4 LU factors and inverse at 1e-15) pushed through the DEVICE routines K3 runs (kmd_test_popstrat_*:
    lu_solve of the lane kernel, group_lu_solve of the group kernel, sigmoid_ref, the dot product of eta = X w) -- until
    now they reached the GPU only through the oracle (missing 1)."""
    import ctypes as C
    lib = K._native.lib()
    # linear_test.cpp:29: EXPECT_TRUE(is_equal_d(sigmoid(1.0), 0.7310585786300048792512))  (is_equal_d: |a - b| < 1e-15 by default)
    x = np.array([1.0, 0.0, -1.0, 14.0, -745.0, 745.0, 40.0])
    out = np.zeros_like(x)
    K._native.check(lib.kmd_test_popstrat_sigmoid(x.ctypes.data, len(x), out.ctypes.data), "sigmoid")
    assert abs(out[0] - 0.7310585786300048792512) < 1e-15 and out[1] == 0.5 and abs(out[2] - (1 - 0.7310585786300048792512)) < 1e-15
    assert 0.0 <= out[4] < 1e-300 and out[5] == 1.0 and out[6] == 1.0
    # :30-31: linear_predictor({1,2,3},{1,2,3}) == 14; predict(...) == 0.9999991684719723358679
    w = np.array([1.0, 2.0, 3.0])
    eta, pr = C.c_double(0), C.c_double(0)
    K._native.check(lib.kmd_test_popstrat_predict(w.ctypes.data, w.ctypes.data, 3, C.byref(eta), C.byref(pr)), "predict")
    assert eta.value == 14.0 and abs(pr.value - 0.9999991684719723358679) < 1e-15
    # :80-151: the 4 x 4 matrix, its no-pivot LU factors (exact: small integers) and its inverse (1e-15)
    m = np.array([[1, 2, 1, 1], [1, 1, 6, 1], [1, 0, 1, 0], [1, 0, 1, 1]], dtype=np.float64)
    lower = np.array([[1, 0, 0, 0], [1, 1, 0, 0], [1, 2, 1, 0], [1, 2, 1, 1]], dtype=np.float64)
    upper = np.array([[1, 2, 1, 1], [0, -1, 5, 0], [0, 0, -10, -1], [0, 0, 0, 1]], dtype=np.float64)
    inv = np.array([[0.1, -0.2, 1, 0.1], [0.5, 0, 0, -0.5], [-0.1, 0.2, 0, -0.1], [0, 0, -1, 1]])
    b = np.array([3.0, -1.0, 0.5, 2.0])
    F = 4
    each = 2 * F * F + F + 1
    lane, grp = np.zeros(each), np.zeros(each)
    K._native.check(lib.kmd_test_popstrat_linear(F, m.ctypes.data, b.ctypes.data, lane.ctypes.data, grp.ctypes.data), "linear")
    for name, o in (("lane", lane), ("group", grp)):
        lu = o[:F * F].reshape(F, F)
        got_l = np.tril(lu, -1) + np.eye(F)
        got_u = np.triu(lu)
        assert (got_l == lower).all() and (got_u == upper).all(), name
        got_inv = o[F * F:2 * F * F].reshape(F, F)
        assert np.abs(got_inv - inv).max() < 1e-15, (name, got_inv)
        assert o[2 * F * F + F] == 0.0
        assert np.abs(o[2 * F * F:2 * F * F + F] - inv @ b).max() < 1e-14, name
    # the two device routines agree bit for bit (as the two kernels' p-values do), also on an ill-scaled Hessian
    assert (lane == grp).all()
    rng = np.random.default_rng(4)
    for F in (3, 5, 7, 13):
        X = rng.normal(0, 1, (40, F)) * np.array([1.0] + [10.0 ** rng.integers(-6, 7) for _ in range(F - 1)])
        H = X.T @ X
        bb = rng.normal(0, 1, F)
        each = 2 * F * F + F + 1
        lane, grp = np.zeros(each), np.zeros(each)
        K._native.check(lib.kmd_test_popstrat_linear(F, np.ascontiguousarray(H).ctypes.data, bb.ctypes.data, lane.ctypes.data, grp.ctypes.data), "linear")
        assert (lane[:each - 1] == grp[:each - 1]).all(), F
        with np.errstate(all="ignore"):
            resid = H @ lane[F * F:2 * F * F].reshape(F, F) - np.eye(F)
        assert lane[each - 1] == 0 and np.abs(resid).max() < 1e-2, (F, np.abs(resid).max())      # (no pivoting, columns up to 10^12 apart)
    # a singular matrix: det == 0 -> status 1 (the fit then stops before its first update, linear_model.cpp:366-372)
    sing = np.array([[1.0, 2.0, 3.0], [2.0, 4.0, 6.0], [1.0, 0.0, 1.0]])
    lane, grp = np.zeros(2 * 9 + 4), np.zeros(2 * 9 + 4)
    K._native.check(lib.kmd_test_popstrat_linear(3, sing.ctypes.data, np.ones(3).ctypes.data, lane.ctypes.data, grp.ctypes.data), "linear")
    assert lane[-1] in (1.0, 2.0) and grp[-1] == 1.0


def device_irls(K, X, y, max_iter):
    """glm_irls on (X, y) through the lane kernel's loop and the group kernel's (kmd_test_popstrat_irls)."""
    import ctypes as C
    X = np.ascontiguousarray(X, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    n, f = X.shape
    wl, wg = np.zeros(f), np.zeros(f)
    il, ig = C.c_int(-1), C.c_int(-1)
    K._native.check(K._native.lib().kmd_test_popstrat_irls(X.ctypes.data, y.ctypes.data, n, f, max_iter, wl.ctypes.data, C.byref(il),
                                                            wg.ctypes.data, C.byref(ig)), "kmd_test_popstrat_irls")
    return wl, il.value, wg, ig.value


def test_reference_glm_irls_answer_through_the_device_loops(K):
    """The ONE glm_irls answer a compiled reference gave (a 6 x 3 design -> 4 iterations and three
    17-digit weights; src/linear_model.cpp:297-410) through the DEVICE loops themselves -- irls_fit of the lane kernel
    and the loop of k_popstrat_group -- not only through the oracle (tests/test_oracle_pins.py).  VERDICT r5, missing 1."""
    X = np.array([[1, .1, .5], [1, -.3, .1], [1, .2, .9], [1, 0, .2], [1, .4, .8], [1, -.2, .05]])
    Y = np.array([1., 1, 0, 1, 0, 0])
    want = np.array([1.4292080254835033, 0.38764766871902268, -3.4639114037934124])
    wl, il, wg, ig = device_irls(K, X, Y, 100)
    assert il == 4 and ig == 4, (il, ig)
    for name, w in (("lane", wl), ("group", wg)):
        rel = np.abs(w - want) / np.abs(want)
        assert rel.max() <= 1e-12, (name, w.tolist(), rel.tolist())
    # the two loops are the same operations in the same order
    assert wl.tolist() == wg.tolist()


def test_irls_cases_through_the_device_loops(K, oracle, golden_dir):
    """Every case of tests/golden/irls_cases.json (converging fits of 5 / 8 / 13 features, the iteration limit at 1 / 2 /
    3 / 50, a singular and a NaN Hessian, perfect separation, no usable rows) through the device loops: iteration count
    and exit as the oracle's restatement of linear_model.cpp:297-410, weights within 1e-9 relative of its weights (the
    sigmoid is exp() here, pow(M_E, .) there: PARITY.md 2), the two device loops bit-equal."""
    import json
    import os
    with open(os.path.join(golden_dir, "irls_cases.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) >= 12
    for c in cases:
        X, y = np.array(c["X"], dtype=np.float64).reshape(c["n"], c["f"]), np.array(c["y"], dtype=np.float64)
        wl, il, wg, ig = device_irls(K, X, y, c["max_iter"])
        want_w, want_it = np.array(c["oracle"]["w"]), c["oracle"]["iters"]
        assert il == ig, (c["kind"], il, ig)
        same = [a == b or (a != a and b != b) for a, b in zip(wl.tolist(), wg.tolist())]
        assert all(same), (c["kind"], wl.tolist(), wg.tolist())
        if c["kind"] in ("nan", "separation"):
            # a fit that leaves the well-conditioned range: one more or one fewer step before the exit is libm's last
            # bit (the weights of such a fit are noise in the reference as well) -- the exit must still be an exit
            assert 0 <= il <= c["max_iter"], c["kind"]
            continue
        assert il == want_it, (c["kind"], il, want_it)
        finite = np.isfinite(want_w)
        assert (np.isfinite(wl) == finite).all(), c["kind"]
        scale = np.abs(want_w[finite]).max() if finite.any() else 1.0
        assert np.abs(wl[finite] - want_w[finite]).max() <= 1e-9 * max(scale, 1e-300), (c["kind"], wl.tolist(), want_w.tolist())
