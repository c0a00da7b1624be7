"""Pin the CPU oracle (oracle/kmd_oracle.c) to the reference BEFORE anything trusts it:
  * the reference's own known answers (tests/factorial_test.cpp:7-16, corrector_test.cpp:9-45,
    model_test.cpp:45-81, linear_test.cpp:29-31,80-151) and SURVEY.md 8c's compiled-reference
    values, and
  * tests/golden/*.json, produced by the reference's own sources (tools/make_golden.py via
    oracle/_ref).  Bit-exact (==) on every double.
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as OL

fh = float.fromhex


def load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


# ---- reference tests/factorial_test.cpp:7-16 -------------------------------------------------
def test_log_factorial_reference_known_answers(oracle):
    t = oracle.lf_table(50)
    assert oracle.lf_at(t, 0) == 0
    assert oracle.lf_at(t, 1) == 0
    assert oracle.lf_at(t, 10) == 15.104412573075514
    assert oracle.lf_at(t, 50) == 148.47776695177302
    assert oracle.lf_at(t, 51) == 152.40959258449737
    assert oracle.lf_at(t, 100) == 363.7393755555635


def test_log_factorial_golden(oracle, golden_dir):
    g = load(golden_dir, "log_factorial.json")
    for size, tab in g["tables"].items():
        t = oracle.lf_table(int(size))
        for i, v in zip(tab["i"], tab["lf"]):
            assert oracle.lf_at(t, i) == fh(v), (size, i)


# ---- alglib chi-square tail -------------------------------------------------------------------
def test_chisqc_golden_v1(oracle, golden_dir):
    g = load(golden_dir, "chisqc_v1.json")
    for x, p in zip(g["x"], g["p"]):
        assert oracle.chisqc(1.0, fh(x)) == fh(p), x


def test_chisqc_golden_other_dof(oracle, golden_dir):
    g = load(golden_dir, "chisqc_vN.json")
    for v, ps in zip(g["v"], g["p"]):
        for x, p in zip(g["x"], ps):
            assert oracle.chisqc(v, fh(x)) == fh(p), (v, x)


def test_chisqc_survey_known_answers(oracle):
    # SURVEY.md 8c (compiled reference, g++ -O2)
    assert oracle.chisqc(1, 0.0) == 1.0
    assert oracle.chisqc(1, 30.0) == 4.3204630578274962e-08
    assert oracle.chisqc(1, 2000.0) == 0.0
    assert oracle.chisqc(1, 2 * 12.6) == 5.1682197538113503e-07
    assert oracle.chisqc(1, 2 * 12.7) == 4.6591811765011513e-07
    assert np.isnan(oracle.chisqc(1, -1.0))        # alglib asserts x >= 0 (specialfunctions.cpp:9564)


# ---- reference tests/corrector_test.cpp:9-45 --------------------------------------------------
def test_correctors_reference_known_answers(oracle):
    c = oracle.corrector(OL_CORR["nothing"], 0.05, 0)
    assert oracle.corrector_apply(c, 0.04) and not oracle.corrector_apply(c, 0.06)
    c = oracle.corrector(OL_CORR["bonferroni"], 0.05, 100)
    assert oracle.corrector_apply(c, 0.0004) and not oracle.corrector_apply(c, 0.0006)
    c = oracle.corrector(OL_CORR["benjamini"], 0.25, 25)
    assert oracle.corrector_apply(c, 0.009) and not oracle.corrector_apply(c, 0.02)
    c = oracle.corrector(OL_CORR["sidak"], 0.05, 100)
    assert oracle.corrector_apply(c, 0.00050) and not oracle.corrector_apply(c, 0.00052)
    c = oracle.corrector(OL_CORR["holm"], 0.05, 100)
    for _ in range(90):
        oracle.corrector_apply(c, 0)
    assert oracle.corrector_apply(c, 0.004) and not oracle.corrector_apply(c, 0.006)


OL_CORR = {"nothing": 0, "bonferroni": 1, "benjamini": 2, "sidak": 3, "holm": 4}


def test_correctors_golden(oracle, golden_dir):
    g = load(golden_dir, "correctors.json")
    for case in g["cases"]:
        c = oracle.corrector(case["type"], fh(case["threshold"]), case["total"])
        got = [oracle.corrector_apply(c, fh(p)) for p in case["p"]]
        assert got == case["apply"], case["name"]


# ---- PoissonLikelihood::process ----------------------------------------------------------------
def test_poisson_reference_model_test_signs(oracle):
    # tests/model_test.cpp:45-81: 30v30, totals all 1, preload 10
    lf = oracle.lf_table(10)
    v = np.array([[200] * 30 + [100] * 30], dtype=np.uint32)
    p, s, mc, mk = oracle.poisson_rows(v, OL.LAYOUT_ROWS, 30, 30, 30, 30, lf)
    assert s[0] == 0                                  # CONTROL
    assert p[0] == 1.0932047323640264e-223
    assert (mc[0], mk[0]) == (6000.0, 3000.0)
    v = np.array([[100] * 30 + [200] * 30], dtype=np.uint32)
    assert oracle.poisson_rows(v, OL.LAYOUT_ROWS, 30, 30, 30, 30, lf)[1][0] == 1   # CASE
    v = np.array([[100] * 60], dtype=np.uint32)
    assert oracle.poisson_rows(v, OL.LAYOUT_ROWS, 30, 30, 30, 30, lf)[1][0] == 2   # NO


def test_poisson_survey_known_answers(oracle):
    lf = oracle.lf_table(10000)
    r = np.array([[10, 12, 9, 11, 30, 28, 35, 31]], dtype=np.uint32)
    p, s, mc, mk = oracle.poisson_rows(r, OL.LAYOUT_ROWS, 4, 4, 4 * 10 ** 9, 4 * 12 * 10 ** 8, lf)
    assert p[0] == 8.1662770190581812e-08 and s[0] == 1
    assert mc[0] == 50.399999999999999 and mk[0] == 124
    lf = oracle.lf_table(100)
    rows = np.array([[0, 0, 50, 70], [60, 70, 0, 0], [5, 5, 5, 5], [100, 150, 5, 5]], dtype=np.uint32)
    p, s, mc, mk = oracle.poisson_rows(rows, OL.LAYOUT_ROWS, 2, 2, 2000, 2000, lf)
    assert list(p) == [4.6264695019718071e-38, 4.3427411549067706e-41, 1.0, 6.6139036424147987e-62]
    assert list(s) == [1, 0, 2, 0]


@pytest.mark.parametrize("layout", [OL.LAYOUT_ROWS, OL.LAYOUT_SOA])
@pytest.mark.parametrize("dtype", [np.uint32, np.uint16, np.uint8])
def test_poisson_golden(oracle, golden_dir, layout, dtype):
    g = load(golden_dir, "poisson_rows.json")
    for case in g["cases"]:
        rows = np.array(case["rows"], dtype=np.uint32)
        if rows.max() > np.iinfo(dtype).max:
            continue
        rows = rows.astype(dtype)
        lf = oracle.lf_table(case["preload"])
        tc, tk = sum(case["total_controls"]), sum(case["total_cases"])
        mat = rows if layout == OL.LAYOUT_ROWS else np.ascontiguousarray(rows.T)
        p, s, mc, mk = oracle.poisson_rows(mat, layout, case["nc"], case["nk"], tc, tk, lf)
        assert [float(x).hex() for x in p] == case["p"]
        assert s.tolist() == case["sign"]
        assert [float(x).hex() for x in mc] == case["mean_control"]
        assert [float(x).hex() for x in mk] == case["mean_case"]


def test_diff_partition_matches_rows_and_threshold(oracle, golden_dir):
    g = load(golden_dir, "poisson_rows.json")
    case = g["cases"][4]           # 20v20, 120 rows
    rows = np.array(case["rows"], dtype=np.uint32)
    lf = oracle.lf_table(case["preload"])
    tc, tk = sum(case["total_controls"]), sum(case["total_cases"])
    thr = 0.05 / 100000
    out = oracle.diff_partition(rows, OL.LAYOUT_ROWS, case["nc"], case["nk"], tc, tk, lf, thr)
    p = np.array([fh(x) for x in case["p"]])
    sel = np.nonzero(p <= thr)[0]
    assert out["row"].tolist() == sel.tolist()
    assert out["pvalue"].tolist() == p[sel].tolist()
    sign = np.array(case["sign"])[sel]
    assert out["counters"] == (120, len(sel), int((sign == 0).sum()), int((sign != 0).sum()))


# ---- aggregator decisions ----------------------------------------------------------------------
def test_aggregate_sorted_stops_at_first_reject(oracle):
    p = np.array([1e-9, 3e-3, 1e-12, 0.5, 2e-9, 1e-3])
    # BH, fdr .05, N=6 : sorted 1e-12,1e-9,2e-9,1e-3,3e-3,.5 ; cuts .0083,.0167,.025,.033,.0417,.05
    keep = oracle.aggregate(2, 0.05, 6, p)
    assert keep.tolist() == [1, 1, 1, 0, 1, 1]
    # Holm, alpha .05, N=6 : cuts .05/6,.05/5,.05/4,.05/3 = .0167 ,.05/2=.025, .05
    keep = oracle.aggregate(4, 0.05, 6, p)
    assert keep.tolist() == [1, 1, 1, 0, 1, 1]
    keep = oracle.aggregate(1, 0.05, 6, p)      # Bonferroni: p < .00833
    assert keep.tolist() == [1, 1, 1, 0, 1, 1]
    p2 = np.array([1e-12, 0.02, 1e-9, 1e-10])   # BH N=4 fdr .05: sorted 1e-12,1e-10,1e-9,.02 -> .02<.05 ok
    assert oracle.aggregate(2, 0.05, 4, p2).tolist() == [1, 1, 1, 1]
    p3 = np.array([1e-12, 0.03, 0.04, 0.045])   # sorted cuts .0125,.025,.0375,.05: .03 fails at rank 2 -> stop
    assert oracle.aggregate(2, 0.05, 4, p3).tolist() == [1, 0, 0, 0]


# ---- reference tests/linear_test.cpp -----------------------------------------------------------
def test_linear_reference_known_answers(oracle):
    L = oracle.L
    assert abs(L.kmdo_sigmoid(1.0) - 0.7310585786300048792512) < 1e-15     # linear_test.cpp:29
    assert L.kmdo_sigmoid(1.0) == 0.7310585786300049
    m = np.array([[1, 2, 1, 1], [1, 1, 6, 1], [1, 0, 1, 0], [1, 0, 1, 1]], dtype=np.float64)
    lower = np.zeros((4, 4))
    upper = np.zeros((4, 4))
    L.kmdo_lu(m.ctypes.data, 4, lower.ctypes.data, upper.ctypes.data)
    assert lower.tolist() == [[1, 0, 0, 0], [1, 1, 0, 0], [1, 2, 1, 0], [1, 2, 1, 1]]   # :117-122
    assert upper.tolist() == [[1, 2, 1, 1], [0, -1, 5, 0], [0, 0, -10, -1], [0, 0, 0, 1]]  # :124-129
    inv = np.zeros((4, 4))
    flags = L.kmdo_inverse(m.ctypes.data, 4, inv.ctypes.data)
    want = np.array([[0.1, -0.2, 1, 0.1], [0.5, 0, 0, -0.5], [-0.1, 0.2, 0, -0.1], [0, 0, -1, 1]])
    assert flags == 0 and np.abs(inv - want).max() < 1e-15                   # :131-150


def test_glm_irls_survey_known_answer(oracle):
    # compiled reference glm_irls on a 6x3 design
    X = np.array([[1, .1, .5], [1, -.3, .1], [1, .2, .9], [1, 0, .2], [1, .4, .8], [1, -.2, .05]])
    Y = np.array([1., 1, 0, 1, 0, 0])
    w = np.zeros(3)
    it = oracle.L.kmdo_glm_irls(X.ctypes.data, Y.ctypes.data, 6, 3, 100, w.ctypes.data, None, None)
    assert it == 4
    assert w.tolist() == [1.4292080254835033, 0.38764766871902268, -3.4639114037934124]
