"""The error bound the fused merge's second pre-filter stage rests on (kmd_tilemerge.hip, row_may_pass_kl), checked on
the CPU: the likelihood ratio in single precision -- with the hardware's reciprocal and base-2 logarithm each off by a
whole ulp in the worst direction -- stays within the slack the kernel allows of the ratio in double precision, for count
sums below 2^16 and totals up to 16 : 1 apart (the conditions the stage is enabled under).  A row is dropped only if
lr_f32 + slack < cut, so |lr_f32 - LR| <= slack means no candidate (LR >= cut) is ever dropped."""
import itertools

import numpy as np

F = np.float32


def lr_f32(sc, sk, qc, qk, d_rcp_c, d_rcp_k, d_log_c, d_log_k):
    """row_may_pass_kl's arithmetic, operation by operation in float32; d_* in {-1, 0, +1}: ulps the reciprocal /
    the logarithm of the control / case term are off by"""
    def off(x, d):
        x = x.astype(F)
        up, dn = np.nextafter(x, F(np.inf)), np.nextafter(x, F(-np.inf))
        return np.where(d > 0, up, np.where(d < 0, dn, x)).astype(F)
    scf, skf = sc.astype(F), sk.astype(F)
    n = (scf + skf).astype(F)
    with np.errstate(divide="ignore", invalid="ignore"):
        rc = off((F(1) / (n * qc).astype(F)).astype(F), d_rcp_c)
        rk = off((F(1) / (n * qk).astype(F)).astype(F), d_rcp_k)
        lc = off(np.log2((scf * rc).astype(F).astype(np.float64)).astype(F), d_log_c)
        lk = off(np.log2((skf * rk).astype(F).astype(np.float64)).astype(F), d_log_k)
        tc = np.where(sc > 0, (scf * lc).astype(F), F(0)).astype(F)
        tk = np.where(sk > 0, (skf * lk).astype(F), F(0)).astype(F)
    lr = (F(0.69314718) * (tc + tk).astype(F)).astype(F)
    slack = (F(5e-6) * n + F(2e-6) * (np.abs(tc) + np.abs(tk)).astype(F) + F(1e-3)).astype(F)
    return lr.astype(np.float64), slack.astype(np.float64)


def lr_f64(sc, sk, Tc, Tk):
    sc, sk = sc.astype(np.float64), sk.astype(np.float64)
    n, T = sc + sk, float(Tc + Tk)
    with np.errstate(divide="ignore", invalid="ignore"):
        a = np.where(sc > 0, sc * np.log(sc / (n * Tc / T)), 0.0)
        b = np.where(sk > 0, sk * np.log(sk / (n * Tk / T)), 0.0)
    return a + b


def test_single_precision_likelihood_ratio_stays_within_its_slack():
    rng = np.random.default_rng(20260301)
    worst = 0.0
    for Tc, Tk in [(10 ** 9, 10 ** 9), (3 * 10 ** 8, 4 * 10 ** 9), (16 * 10 ** 7, 10 ** 7), (12345, 67890), (7, 100)]:
        qc, qk = F(Tc / (Tc + Tk)), F(Tk / (Tc + Tk))
        # small sums (every pair), sums over the whole range, one-sided rows, rows split like the totals
        a, b = np.meshgrid(np.arange(0, 200), np.arange(0, 200), indexing="ij")
        sums = [(a.ravel()[1:], b.ravel()[1:]),
                (rng.integers(0, 1 << 16, 200_000), rng.integers(0, 1 << 16, 200_000)),
                (rng.integers(1, 1 << 16, 50_000), np.zeros(50_000, dtype=np.int64)),
                (np.zeros(50_000, dtype=np.int64), rng.integers(1, 1 << 16, 50_000))]
        t = rng.integers(2, 1 << 16, 100_000)
        split = np.clip(np.round(t * Tc / (Tc + Tk) + rng.integers(-3, 4, len(t))), 0, t).astype(np.int64)
        sums.append((split, t - split))
        for sc, sk in sums:
            keep = (sc + sk) > 0
            sc, sk = sc[keep], sk[keep]
            want = lr_f64(sc, sk, Tc, Tk)
            for d in itertools.product((-1, 1), repeat=4):
                got, slack = lr_f32(sc, sk, qc, qk, *d)
                err = np.abs(got - want)
                assert (err <= slack).all(), (Tc, Tk, d, float((err / slack).max()), sc[np.argmax(err / slack)], sk[np.argmax(err / slack)])
                worst = max(worst, float((err / slack).max()))
    assert worst < 0.5, worst                                  # (the kernel allows ten times the analysed error)
