"""The restatement of glm_irls (src/linear_model.cpp:297-410; oracle/kmd_oracle.c kmdo_glm_irls) against
tests/golden/irls_cases.json: what an independent Newton-Raphson (numpy, pivoted LAPACK solve; generator
tools/make_irls_golden.py) computes for the same inputs, and first-principles expectations for the
edge exits.  linear_model.cpp itself cannot be compiled in this image (it includes spdlog): this file and
the LU / inverse / sigmoid vectors of tests/linear_test.cpp (test_oracle_pins.py) are the evidence there is.
"""
import json
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def cases(golden_dir):
    with open(os.path.join(golden_dir, "irls_cases.json")) as f:
        return json.load(f)["cases"]


def run(oracle, c):
    X, y = np.array(c["X"]), np.array(c["y"])
    w, err, fl = np.zeros(c["f"]), np.zeros(1), np.zeros(1, dtype=np.int32)
    it = oracle.L.kmdo_glm_irls(X.ctypes.data, y.ctypes.data, c["n"], c["f"], c["max_iter"], w.ctypes.data, err.ctypes.data, fl.ctypes.data)
    return w, int(it), int(fl[0]), float(err[0])


def test_oracle_reproduces_the_stored_runs(oracle, cases):
    for c in cases:
        w, it, fl, err = run(oracle, c)
        assert (it, fl) == (c["oracle"]["iters"], c["oracle"]["flags"]), c["kind"]
        assert [float(v).hex() for v in w] == [float(v).hex() for v in c["oracle"]["w"]], c["kind"]


def test_irls_is_newton_raphson(oracle, cases):
    """IRLS from the reference's start point = Newton's method: the same number of steps by an independent
    implementation gives the same weights to rounding; the reference's loose stopping rule (MSE change
    < 1e-6) still lands within 1e-6 of the maximum likelihood estimate on well-conditioned designs."""
    n_checked = 0
    for c in cases:
        if "newton_same_steps" not in c:
            continue
        w, _, _, _ = run(oracle, c)
        ref = np.array(c["newton_same_steps"])
        assert np.abs(w - ref).max() <= 1e-9 * np.abs(ref).max(), c["kind"]
        if "mle" in c:
            mle = np.array(c["mle"])
            assert np.abs(w - mle).max() <= 1e-6 * np.abs(mle).max()
        n_checked += 1
    assert n_checked >= 10


def test_irls_edge_exits(oracle, cases):
    by = {}
    for c in cases:
        by.setdefault(c["kind"], []).append(c)
    ones = lambda c: [1.0] * c["f"]
    # iteration limit (linear_model.cpp:386-395): the update of the iteration that reaches the limit is dropped
    lim = {c["max_iter"]: c for c in by["limit"]}
    w1, it1, _, _ = run(oracle, lim[1])
    assert it1 == 1 and w1.tolist() == ones(lim[1])
    for k in (2, 3, 4):
        wk, itk, _, _ = run(oracle, lim[k])
        assert itk == k
    # the weights after limit k + 1 are the update computed (and dropped) in iteration k of limit k
    assert run(oracle, lim[3])[0].tolist() != run(oracle, lim[2])[0].tolist()
    # singular Hessian: det == 0 -> leave before the first update: weights untouched (:182-186, :366-372)
    c = by["singular"][0]
    w, it, fl, _ = run(oracle, c)
    assert (it, fl) == (0, 1) and w.tolist() == ones(c)
    # a zero pivot in the middle of the no-pivot LU: 0/0 -> det is NaN
    c = by["nan"][0]
    w, it, fl, _ = run(oracle, c)
    assert (it, fl) == (0, 2) and w.tolist() == ones(c)
    # no row with g_i > 1e-305 (:336, :343): nothing to fit
    c = by["no_good_rows"][0]
    w, it, fl, _ = run(oracle, c)
    assert (it, fl) == (0, 0) and w.tolist() == ones(c)
    # perfect separation: the weights grow until the MSE stops changing (:349) -- finite, a few iterations
    c = by["separation"][0]
    w, it, fl, _ = run(oracle, c)
    assert fl == 0 and 3 <= it <= 50 and np.isfinite(w).all() and abs(w[1]) > 5
