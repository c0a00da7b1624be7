"""lower_bound_near (kmd_tilemerge.hip: the tile boundary searches of k_tile_bounds) restated line by line in Python and
held against numpy's searchsorted: any guess, any bracket, repeated keys, brackets of 0 and 1 records, the answer at
either end.  (The kernel's own results are checked through the merge's parity tests on the GPU; this pins the
algorithm -- doubling steps away from the guess, then bisection -- where no GPU is needed.)"""
import numpy as np


def lower_bound_near(keys, lo, hi, g, b):
    """first index in [lo, hi) whose key is >= b, hi if none; the loads it makes are counted"""
    loads = 0

    def less(i):
        nonlocal loads
        assert lo <= i < hi, (lo, hi, i)                      # never reads outside the bracket
        loads += 1
        return keys[i] < b
    if lo >= hi:
        return lo, 0
    g = min(max(g, lo), hi - 1)
    if less(g):
        a, e, step = g + 1, hi, 1
        while a + step - 1 < hi:
            if less(a + step - 1):
                a += step
            else:
                e = a + step - 1
                break
            step <<= 1
        a = min(a, e)
    else:
        a, e, step = lo, g, 1
        while e >= lo + step:
            if not less(e - step):
                e -= step
            else:
                a = e - step + 1
                break
            step <<= 1
    while a < e:
        mid = a + ((e - a) >> 1)
        if less(mid):
            a = mid + 1
        else:
            e = mid
    return a, loads


def test_search_from_a_guess_equals_searchsorted():
    rng = np.random.default_rng(77)
    for trial in range(3000):
        n = int(rng.integers(0, 400))
        keys = np.sort(rng.integers(0, max(2, n // int(rng.integers(1, 4)) + 2), n).astype(np.uint64))     # (many repeats)
        lo = int(rng.integers(0, n + 1))
        hi = int(rng.integers(lo, n + 1))
        b = np.uint64(rng.integers(0, int(keys.max()) + 3 if n else 3))
        g = int(rng.integers(-5, n + 6))
        want = lo + int(np.searchsorted(keys[lo:hi], b, side="left"))
        got, loads = lower_bound_near(keys, lo, hi, g, b)
        assert got == want, (trial, n, lo, hi, g, int(b))
        assert loads <= 2 * max(1, int(np.ceil(np.log2(max(2, hi - lo))))) + 3


def test_a_good_guess_costs_a_handful_of_loads():
    rng = np.random.default_rng(78)
    keys = np.sort(rng.integers(0, 1 << 40, 100_000).astype(np.uint64))
    worst = 0
    for _ in range(2000):
        at = int(rng.integers(0, len(keys)))
        g = at + int(rng.integers(-12, 13))                   # (interpolation between two coarse boundaries lands about this close)
        got, loads = lower_bound_near(keys, 0, len(keys), g, keys[at])
        assert got == int(np.searchsorted(keys, keys[at], side="left"))
        worst = max(worst, loads)
    assert worst <= 12, worst
