"""Host-side restatements (no GPU) of the two pieces of arithmetic kmd_pvalues_refine rests on (kmdiff_amd/csrc/kmd_filter.hip):

1. lf_running_sum_table: LogFactorialTable::log_factorial's running sum `res += log(k), k--` (log_factorial_table.cpp:13-22)
   taken 64 terms at a time -- while the partial sums stay inside one binade [2^E, 2^(E+1)) every one of them is a
   multiple of u = 2^(E-52), fl(r + l) = r + u RN(l / u) unless l / u lies exactly half-way between two integers, so the
   roundings are independent and their integer sum exact; steps that may leave the binade, or hold a tie, take the 64
   additions in order.  Here in Python floats (IEEE doubles) and exact integers, against the plain loop.
2. the inversion of mean_control = fl(fl(sum_c Tk) / Tc) (model.hpp:165): the sum is the integer nearest
   mean_control Tc / Tk or one of its four neighbours, and exactly one of them reproduces the mean."""
import math
import random

import numpy as np


def plain_sum(k, log=math.log):
    res = 0.0
    while k > 1:
        res += log(k)
        k -= 1
    return res


def stepped_sum(k, log=math.log, group=4):
    """the kernel's control flow, lane for lane: g0 = first term not yet added; a step = terms g0, g0 - 1, .. g0 - 63"""
    terms = lambda g, step: [log(g - (64 * step + t)) if g - (64 * step + t) > 1 else 0.0 for t in range(64)]
    res, g0 = 0.0, k
    n_fast = n_ordered = 0
    while g0 > 1:
        if res >= 32.0:
            E = math.frexp(res)[1] - 1
            u = math.ldexp(1.0, E - 52)
            inv_u = math.ldexp(1.0, 52 - E)
            R = res * inv_u
            assert R == int(R) and 2 ** 52 <= R < 2 ** 53
            q_max = math.ceil(log(g0) * inv_u) + 1.0
            n_safe = int((9007199254740992.0 - R) / (64.0 * q_max))
            done, Q = 0, 0.0
            while n_safe - done >= group and g0 > 1:
                xs = [t * inv_u for a in range(group) for t in terms(g0, a)]
                qs = [float(round(x)) if abs(x - round(x)) != 0.5 else None for x in xs]      # (round() ties to even, as rint)
                if any(q is None for q in qs):
                    break
                for q in qs:
                    Q += q                                   # integers below 2^53: exact in doubles
                    assert Q == int(Q) and Q < 2 ** 53
                done += group
                g0 = g0 - 64 * group if g0 > 64 * group else 0
            if done:
                res = (R + Q) * u
                n_fast += done
                continue
        ls = terms(g0, 0)
        fast = False
        if res >= 32.0:
            xs = [t * inv_u for t in ls]
            if all(abs(x - round(x)) != 0.5 for x in xs):
                Q = float(sum(int(round(x)) for x in xs))
                if R + Q < 9007199254740992.0:
                    res = (R + Q) * u
                    fast = True
                    n_fast += 1
        if not fast:
            for t in ls:
                res += t
            n_ordered += 1
        g0 = g0 - 64 if g0 > 64 else 0
    return res, n_fast, n_ordered


def test_stepped_running_sum_is_the_plain_one():
    rng = random.Random(5)
    ks = list(range(0, 200)) + [255, 256, 257, 4095, 4096, 4097, 65535, 65536, 65537] + [rng.randrange(2, 1 << 17) for _ in range(60)]
    fast_steps = 0
    for k in ks:
        got, n_fast, n_ordered = stepped_sum(k)
        assert got == plain_sum(k), k
        fast_steps += n_fast
        if k > 20000:
            assert n_ordered < 80 and n_fast > 3 * n_ordered, (k, n_fast, n_ordered)     # ordered steps: the binade crossings and the odd tie
    assert fast_steps > 10000


def test_stepped_running_sum_with_terms_that_tie():
    """Terms chosen so that l / u is exactly half-way (multiples of 2^-40 plus half an ulp of the sum's binade): the tie
    rounds to the even partial SUM, which the independent rounding cannot know -- those steps must go the ordered way."""
    tie_log = lambda j: 10.0 + (j % 7) * 2.0 ** -30 + (2.0 ** -37 if j % 5 == 0 else 0.0)   # binade of ~1e5: u = 2^-36, half = 2^-37
    for k in (12000, 20011, 33333):
        got, n_fast, n_ordered = stepped_sum(k, log=tie_log)
        assert got == plain_sum(k, log=tie_log), k
        assert n_ordered > 20                                # the ties were met


def test_sum_recovered_from_the_control_mean():
    rng = np.random.default_rng(3)
    for _ in range(20000):
        tc, tk = float(rng.integers(1, 1 << int(rng.integers(1, 45)))), float(rng.integers(1, 1 << int(rng.integers(1, 45))))
        sc = int(rng.integers(0, 1 << int(rng.integers(1, 40))))
        mc = float(sc) * tk / tc                              # model.hpp:165
        guess = mc * tc / tk
        if not guess < 9.0e15:
            continue
        g = int(np.rint(guess))
        hits = [c for c in (g, g - 1, g + 1, g - 2, g + 2) if c >= 0 and float(c) * tk / tc == mc]
        assert sc in hits, (sc, tc, tk, hits)
        # ... and only the true sum does: two sums below 2^40 are at least 2^-40 apart relatively, their means likewise
        assert hits == [sc], (sc, tc, tk, hits)
