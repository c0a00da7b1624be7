"""kmd_ddmath.h (host build, through the library's test hooks): log and exp rounded correctly to double.
These decide the rows whose p-value lies within 1e-8 of the threshold (KMD_CNT_NEAR_THRESHOLD), so that
the last bit of one libm or another cannot move a row across `p <= threshold` (merge.hpp:78).
Checked against mpmath at 200 digits; the same arguments through this host's libm (glibc) show how often
a glibc-built reference returns the rounded value itself."""
import math

import mpmath
import numpy as np
import pytest


@pytest.fixture(scope="module")
def L():
    from kmdiff_amd import _native
    return _native.lib()


def rounded(f, x):
    with mpmath.workprec(700):
        return float(f(mpmath.mpf(x)))          # mpf -> float rounds to nearest


def test_log_is_correctly_rounded(L):
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(0.5, 2.0, 1500), np.exp(rng.uniform(-700, 700, 1500)), rng.uniform(1, 1e9, 1000),
                         1.0 + rng.uniform(-1e-3, 1e-3, 500), np.array([1.0, 2.0, 0.5, 1e-300, 1e300, 3.0, 10.0])])
    bad = libm_bad = 0
    for x in xs:
        want = rounded(mpmath.log, float(x))
        bad += L.kmd_test_log_rounded(float(x)) != want
        libm_bad += math.log(float(x)) != want
    assert bad == 0
    assert libm_bad <= 0.01 * len(xs)            # glibc 2.35: almost always the rounded value too


def test_exp_is_correctly_rounded(L):
    rng = np.random.default_rng(2)
    xs = np.concatenate([rng.uniform(-700, 700, 2000), rng.uniform(-30, 0, 2000), rng.uniform(-1, 1, 500),
                         np.array([0.0, 1.0, -1.0, -14.5, 1e-10, -1e-10, 707.0, -707.0])])
    bad = libm_bad = 0
    for x in xs:
        want = rounded(mpmath.exp, float(x))
        bad += L.kmd_test_exp_rounded(float(x)) != want
        libm_bad += math.exp(float(x)) != want
    assert bad == 0
    assert libm_bad <= 0.01 * len(xs)


def test_tail_function_over_rounded_libm_agrees_with_the_oracle_almost_always(L, oracle):
    """igamc(1/2, x) with correctly rounded log / exp is bit-for-bit what the oracle (glibc) returns wherever
    glibc's two calls were correctly rounded themselves -- the premise of the guard."""
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.uniform(0.01, 60, 3000), rng.uniform(10, 20, 1000)])
    same = sum(L.kmd_test_igamc_half_rounded(float(x)) == oracle.chisqc(1, 2.0 * float(x)) for x in xs)
    assert same >= 0.98 * len(xs)
    for x in xs[:200]:
        a, b = L.kmd_test_igamc_half_rounded(float(x)), oracle.chisqc(1, 2.0 * float(x))
        assert abs(a - b) <= 1e-12 * b
