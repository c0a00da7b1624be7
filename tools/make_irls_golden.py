#!/usr/bin/env python3
"""tests/golden/irls_cases.json -- evidence for the restatement of glm_irls (src/linear_model.cpp:297-410),
which cannot be compiled here (it includes spdlog, absent from the image): the oracle's kmdo_glm_irls
against an INDEPENDENT implementation that shares no code with it.

For every case the file stores the inputs, what the oracle returned, and what this script computed on
its own with numpy:
  * `newton_same_steps`: plain Newton-Raphson on the logistic log-likelihood (numpy.linalg.solve, pivoted
    LAPACK -- not the reference's no-pivot LU), started where the reference starts (mu = (y + 0.5) / 2)
    and run for exactly as many steps as the oracle reports updates that were COPIED into `weight`
    (linear_model.cpp:386-395: the update of the iteration that hits the limit is computed and dropped).
    IRLS is Newton's method, so the two must agree to rounding: the test holds them to 1e-9 relative.
  * `mle`: the same Newton iteration run to convergence (|step| < 1e-13): how far the reference's own
    stopping rule (change of the mean squared error < 1e-6, :349) leaves its answer from the maximum
    likelihood estimate -- reported, and bounded loosely by the test.
Edge exits are stored as what the oracle does and checked against first-principles expectations in
tests/test_irls_cases.py (flags, iteration counts, untouched weights), not against another number.

Run in the build container:  python3 tools/make_irls_golden.py   (needs oracle/liboracle.so)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as OL  # noqa: E402


def oracle_irls(o, X, y, max_iter):
    n, f = X.shape
    w = np.zeros(f)
    err = np.zeros(1)
    flags = np.zeros(1, dtype=np.int32)
    it = o.L.kmdo_glm_irls(np.ascontiguousarray(X).ctypes.data, np.ascontiguousarray(y).ctypes.data, n, f, max_iter,
                           w.ctypes.data, err.ctypes.data, flags.ctypes.data)
    return w, int(it), int(flags[0]), float(err[0])


def newton(X, y, steps=None, tol=1e-13, limit=200):
    """Newton-Raphson on the Bernoulli log-likelihood, first step from the reference's start point."""
    mu = (y + 0.5) / 2
    eta = np.log(mu / (1 - mu))
    w = None
    k = 0
    while True:
        g = mu * (1 - mu)
        z = eta + (y - mu) / g
        H = X.T @ (g[:, None] * X)
        w_new = np.linalg.solve(H, X.T @ (g * z))
        k += 1
        done = (steps is not None and k >= steps) or (steps is None and w is not None and np.abs(w_new - w).max() < tol) or k >= limit
        w = w_new
        if done:
            return w, k
        eta = X @ w
        mu = 1 / (1 + np.exp(-eta))


def main():
    o = OL.load()
    rng = np.random.default_rng(20261002)
    cases = []
    # ---- well-conditioned designs: intercept, npc principal components, a depth-like column, a k-mer column
    for (n, npc) in [(8, 2), (40, 2), (40, 5), (100, 10), (200, 2), (200, 10)]:
        f = 3 + npc
        X = np.ones((n, f))
        X[:, 1:1 + npc] = rng.normal(0, 1.0, (n, npc))
        X[:, 1 + npc] = rng.normal(0, 1.0, n)
        y = np.concatenate([np.ones(n // 2), np.zeros(n - n // 2)])
        X[:, f - 1] = rng.gamma(2.0, 1.0, n) * (1 + 0.6 * y) * 1e-3 * 300     # a k-mer column with a moderate effect
        w, it, fl, err = oracle_irls(o, X, y, 100)
        copied = it if it < 100 else it - 1
        # the oracle leaves the loop at the top of iteration it + 1 (MSE change < 1e-6): `it` updates copied
        ws, _ = newton(X, y, steps=copied)
        wm, km = newton(X, y)
        cases.append({"kind": "converging", "n": n, "f": f, "max_iter": 100, "X": X.tolist(), "y": y.tolist(),
                      "oracle": {"w": w.tolist(), "iters": it, "flags": fl, "ret_error": err},
                      "newton_same_steps": ws.tolist(), "mle": wm.tolist(), "mle_steps": km})
    # ---- iteration limit 1, 2, 3 (break before copy): weight = the update of iteration limit - 1 (limit 1: ones)
    n, f = 30, 5
    X = np.ones((n, f)); X[:, 1:] = rng.normal(0, 1, (n, f - 1))
    y = (rng.random(n) < 1 / (1 + np.exp(-(X @ np.array([0.2, 1.0, -0.5, 0.3, 0.8]))))).astype(float)
    for lim in (1, 2, 3, 4):
        w, it, fl, err = oracle_irls(o, X, y, lim)
        expect = np.ones(f) if lim == 1 else newton(X, y, steps=lim - 1)[0]
        cases.append({"kind": "limit", "n": n, "f": f, "max_iter": lim, "X": X.tolist(), "y": y.tolist(),
                      "oracle": {"w": w.tolist(), "iters": it, "flags": fl, "ret_error": err},
                      "newton_same_steps": expect.tolist()})
    # ---- singular Hessian: the last column is all zero -> last pivot exactly 0, det == 0 (:182-186, :366-372)
    Xs = X.copy(); Xs[:, f - 1] = 0.0
    w, it, fl, err = oracle_irls(o, Xs, y, 100)
    cases.append({"kind": "singular", "n": n, "f": f, "max_iter": 100, "X": Xs.tolist(), "y": y.tolist(),
                  "oracle": {"w": w.tolist(), "iters": it, "flags": fl, "ret_error": err}})
    # ---- NaN: a zero column in the MIDDLE -> 0/0 in the no-pivot LU, det is NaN
    Xn = X.copy(); Xn[:, 2] = 0.0
    w, it, fl, err = oracle_irls(o, Xn, y, 100)
    cases.append({"kind": "nan", "n": n, "f": f, "max_iter": 100, "X": Xn.tolist(), "y": y.tolist(),
                  "oracle": {"w": w.tolist(), "iters": it, "flags": fl, "ret_error": err}})
    # ---- perfect separation: the weights run away until every g_i = mu (1 - mu) underflows 1e-305 (:336, :343)
    Xp = np.ones((20, 2)); Xp[:, 1] = np.concatenate([np.linspace(1, 2, 10), np.linspace(-2, -1, 10)])
    yp = np.concatenate([np.ones(10), np.zeros(10)])
    w, it, fl, err = oracle_irls(o, Xp, yp, 5000)
    cases.append({"kind": "separation", "n": 20, "f": 2, "max_iter": 5000, "X": Xp.tolist(), "y": yp.tolist(),
                  "oracle": {"w": w.tolist(), "iters": it, "flags": fl, "ret_error": err}})
    # ---- no usable row at all (:336, :343): every g_i = mu_i (1 - mu_i) <= 1e-305 on entry.  Phenotypes are 0 / 1 in
    # kmdiff (popstrat.cpp:168), where this cannot happen at the first pass (mu = 0.25 / 0.75); y = 2 puts mu at 1.25
    yb = np.full(n, 2.0)
    w, it, fl, err = oracle_irls(o, X, yb, 100)
    cases.append({"kind": "no_good_rows", "n": n, "f": f, "max_iter": 100, "X": X.tolist(), "y": yb.tolist(),
                  "oracle": {"w": w.tolist(), "iters": it, "flags": fl, "ret_error": err}})
    out = os.path.join(ROOT, "tests", "golden", "irls_cases.json")
    with open(out, "w") as fh:
        json.dump({"generator": "tools/make_irls_golden.py", "numpy": np.__version__, "cases": cases}, fh)
    for c in cases:
        line = "%-10s n=%3d f=%2d limit=%4d  oracle: iters=%d flags=%d" % (c["kind"], c["n"], c["f"], c["max_iter"], c["oracle"]["iters"], c["oracle"]["flags"])
        if "newton_same_steps" in c:
            a, b = np.array(c["oracle"]["w"]), np.array(c["newton_same_steps"])
            line += "  |w - newton(same steps)| / |w| = %.2e" % (np.abs(a - b).max() / max(1e-300, np.abs(b).max()))
        if "mle" in c:
            line += "  to the MLE: %.2e" % (np.abs(np.array(c["oracle"]["w"]) - np.array(c["mle"])).max() / np.abs(np.array(c["mle"])).max())
        print(line)
    print("wrote", out)


if __name__ == "__main__":
    main()
