#!/usr/bin/env python3
"""GPU box: a randomized soak of the product paths against the oracle, beyond the committed (seeded, small) tests:
  * kmd_merge_filter on random partitions (2..260 samples, 0..60 000 k-mers, spread / clustered / consecutive keys, empty
    samples, huge counts, thresholds 1 .. 1e-9) == oracle merge + diff_partition       (tests/test_gpu_tilemerge.run_fused)
  * kmd_poisson_filter on random count matrices (every layout and count width), kmd_merge_filter_batch against single calls
  * kmd_popstrat_apply on random designs (samples, principal components, iteration limits, effect sizes)
  * kmd_pack_block / kmd_unpack_streams round trips on random streams
  * kmd_correct_sharded over 2..9 virtual ranks == kmd_correct over the whole list, all correctors
usage: python3 tests/soak.py [--seconds 300] [--seed N]      (prints one line per 50 cases; any mismatch raises)"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kmdiff_amd as K                     # noqa: E402
import oracle_lib as OL                    # noqa: E402
from test_gpu_parity import totals_of, sharded_decisions       # noqa: E402
from test_gpu_tilemerge import make_streams, run_fused         # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=300)
ap.add_argument("--seed", type=int, default=int(time.time()))
ap.add_argument("--cases", type=int, default=0, help="stop after this many cases (0: by --seconds alone): the same cases on every box")
ap.add_argument("--only", default="", help="'popstrat': pop-strat designs only")
ap.add_argument("--tally", action="store_true", help="pop-strat rows outside the bars that jitter does not explain are counted (with the worst) instead of ending the run")
ap.add_argument("--popstrat-stand", default="both", choices=["both", "true", "false"],
                help="'true': standardised designs only -- the only mode the reference can reach (s_stand starts true and set_params can "
                     "only turn it on, popstrat.hpp:155,174-175); 'false' is a stress case")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
oracle = OL.load()
N = K._native
print("soak seed", a.seed, flush=True)
DEV = {"raw_abs": 0.0, "raw_rel": 0.0, "ref_rel_in": 0.0, "ref_abs_in": 0.0, "in_rows": 0, "in_equal": 0, "beyond_abs": 0.0, "beyond_rel": 0.0, "beyond_rows": 0}


def fused_case(streams, nc, thr, lf_n, two=False):
    """tests/test_gpu_tilemerge.run_fused with the p-values MEASURED: everything else exact; p as the filter wrote it, then
    after kmd_pvalues_refine -- rows with both sums inside the table apart from the others"""
    S = len(streams)
    if two:
        want, wlo, whi = oracle.merge_partition2(streams)
    else:
        want, wlo = oracle.merge_partition(streams)
    tcs, tks = totals_of(want, nc)
    ref = oracle.diff_partition(want, OL.LAYOUT_ROWS, nc, S - nc, int(tcs.sum()), int(tks.sum()), oracle.lf_table(lf_n), thr)
    model = K.PoissonLikelihood(nc, S - nc, tcs, tks, lf_n)
    acc = K.SurvivorAccumulator(max(want.shape[0], 1), kmer_limbs=2 if two else 1)
    K.merge_filter(K.StreamSet(streams), K.diff_observer(model, acc, thr))
    n = acc.finish(by_kmer=True)
    got = acc.get()
    c = acc.read_counters()
    assert (int(c[0]), int(c[1]), int(c[2]), int(c[3])) == ref["counters"] and int(c[0]) == want.shape[0]
    rr = ref["row"].astype(np.int64)
    assert n == len(rr) and got["kmer_lo"].tolist() == wlo[rr].tolist() and got["sign"].tolist() == ref["sign"].tolist()
    if two:
        assert got["kmer_hi"].tolist() == whi[rr].tolist()
    assert got["mean_control"].tolist() == ref["mean_control"].tolist() and got["mean_case"].tolist() == ref["mean_case"].tolist()
    if n == 0:
        return
    w = ref["pvalue"]
    d = np.abs(got["pvalue"] - w)
    DEV["raw_abs"] = max(DEV["raw_abs"], float(d.max()))
    nz = w > 0
    if nz.any():
        DEV["raw_rel"] = max(DEV["raw_rel"], float((d[nz] / w[nz]).max()))
    check_ = K._native.check
    check_(K._native.lib().kmd_pvalues_refine(model.handle, n, acc.bufs["mean_control"].ptr, acc.bufs["mean_case"].ptr, acc.bufs["pvalue"].ptr, None))
    check_(K._native.lib().kmd_stream_sync(None))
    p2 = acc.bufs["pvalue"].to_host(np.float64, n)
    sc, sk = want[rr, :nc].astype(np.uint64).sum(axis=1), want[rr, nc:].astype(np.uint64).sum(axis=1)
    inside = (sc < lf_n) & (sk < lf_n)
    d2 = np.abs(p2 - w)
    if inside.any():
        DEV["in_rows"] += int(inside.sum()); DEV["in_equal"] += int((p2[inside] == w[inside]).sum())
        DEV["ref_abs_in"] = max(DEV["ref_abs_in"], float(d2[inside].max()))
        m = inside & nz
        if m.any():
            DEV["ref_rel_in"] = max(DEV["ref_rel_in"], float((d2[m] / w[m]).max()))
    if (~inside).any():
        DEV["beyond_equal"] = DEV.get("beyond_equal", 0) + int((p2[~inside] == w[~inside]).sum())
        DEV["beyond_rows"] += int((~inside).sum())
        DEV["beyond_abs"] = max(DEV["beyond_abs"], float(d2[~inside].max()))
        m = ~inside & nz
        if m.any():
            DEV["beyond_rel"] = max(DEV["beyond_rel"], float((d2[m] / w[m]).max()))
t0, n_fused, n_pack, n_shard, n_k1, n_batch = time.time(), 0, 0, 0, 0, 0


def k1_case():
    """kmd_poisson_filter on a random count matrix: every layout and count width, 2..300 samples, rows of zeros, thresholds
    1 .. 1e-9 -- rows, sign, means, counters exact; p after kmd_pvalues_refine bit-equal in >= 99 %, the rest 1e-9 relative"""
    S = int(rng.choice([2, 3, 7, 16, 21, 40, 41, 64, 100, 129, 255, 300]))
    nc = int(rng.integers(1, S))
    n = int(rng.integers(1, 40000 if S <= 64 else 6000))
    dt = [np.uint8, np.uint16, np.uint32][int(rng.integers(0, 3))]
    hi = int(rng.choice([2, 4, 30, 255, 256, 3000, 70000]))
    hi = min(hi, int(np.iinfo(dt).max))
    rows = rng.integers(0, hi + 1, (n, S)).astype(dt)
    rows[rng.random((n, S)) < rng.uniform(0, 0.95)] = 0
    rows[rng.random(n) < 0.02] = 0
    if dt == np.uint32 and rng.random() < 0.2:
        rows[rng.integers(0, n, 3), rng.integers(0, S, 3)] = np.uint32(4_000_000_000)
    tcs, tks = totals_of(rows, nc)
    if int(tcs.sum()) == 0 or int(tks.sum()) == 0:
        return
    layout = [K.LAYOUT_ROWS, K.LAYOUT_SOA, K.LAYOUT_TILED][int(rng.integers(0, 3))]
    thr, lf_n = float(rng.choice([1.0, 0.05, 1e-3, 1e-6, 1e-9])), int(rng.choice([10000, 10000, 500, 20]))
    ref = oracle.diff_partition(rows, OL.LAYOUT_ROWS, nc, S - nc, int(tcs.sum()), int(tks.sum()), oracle.lf_table(lf_n), thr)
    model = K.PoissonLikelihood(nc, S - nc, tcs, tks, lf_n)
    lo = rng.integers(0, 1 << 63, n, dtype=np.uint64)
    mat = K.CountMatrix.from_host(rows if layout == K.LAYOUT_ROWS else np.ascontiguousarray(rows.T), layout, kmer_lo=lo, row_base=int(rng.integers(0, 1 << 40)))
    acc = K.SurvivorAccumulator(n)
    K.diff_observer(model, acc, thr).process(mat)
    ns = acc.finish(refine=model)
    got, c = acc.get(), acc.read_counters()
    tag = (S, nc, n, dt.__name__, layout, thr, lf_n)
    assert (int(c[0]), int(c[1]), int(c[2]), int(c[3])) == ref["counters"], tag
    rr = ref["row"].astype(np.int64)
    assert ns == len(rr) and (got["row"] - np.uint64(mat.row_base)).tolist() == rr.tolist() and got["kmer_lo"].tolist() == lo[rr].tolist(), tag
    assert got["sign"].tolist() == ref["sign"].tolist() and got["mean_control"].tolist() == ref["mean_control"].tolist(), tag
    assert got["mean_case"].tolist() == ref["mean_case"].tolist(), tag
    if ns:
        w, p = ref["pvalue"], got["pvalue"]
        small = (rows[rr, :nc].astype(np.uint64).sum(axis=1) < (1 << 20)) & (rows[rr, nc:].astype(np.uint64).sum(axis=1) < (1 << 20))
        d = np.abs(p - w)
        rel = np.where(w > 0, d / np.where(w > 0, w, 1.0), 0.0)
        # (the bar on the relative deviation: 1e-9, or -- count sums beyond 5e5 -- one ulp of each of the two logarithms of
        # model.hpp:155-156 times its count sum: kmd_pvalues_refine takes correctly rounded logarithms, glibc's own is one ulp
        # off that in ~1 call in 10^3, and `k log(lambda)` multiplies the ulp by the sum: 1.9e-9 at sums of 7e5, PARITY.md 1)
        sums = rows[rr, :].astype(np.uint64).sum(axis=1).astype(np.float64)
        rel_bar = np.maximum(1e-9, 2.0e-15 * sums)
        if not (d[small].max(initial=0) <= 1e-10 and (rel[small] <= rel_bar[small]).all()):
            k = int(np.argmax(np.where(small, rel, 0.0)))
            sc_k, sk_k = int(rows[rr[k], :nc].astype(np.uint64).sum()), int(rows[rr[k], nc:].astype(np.uint64).sum())
            print("FAILED k1 p-value: %r worst row %d: p_dev %.17g p_ref %.17g rel %.3g sums %d %d (Tc %d Tk %d); %d of %d rows beyond 1e-9 relative"
                  % (tag, k, p[k], w[k], rel[k], sc_k, sk_k, int(tcs.sum()), int(tks.sum()), int((rel[small] > 1e-9).sum()), int(small.sum())), flush=True)
            np.savez("gpurun_out/soak_k1_fail.npz", rows=rows, nc=nc, thr=thr, lf_n=lf_n, layout=layout, p_dev=p, p_ref=w, rr=rr)
        assert d[small].max(initial=0) <= 1e-10 and (rel[small] <= rel_bar[small]).all(), (tag, d[small].max(initial=0))
        DEV["k1_rel_beyond_1e-9"] = DEV.get("k1_rel_beyond_1e-9", 0) + int((rel[small] > 1e-9).sum())
        if (~small).any():                       # sums of 2^20 and more keep the filter's value: measured, not asserted beyond 1e-6
            DEV["k1_large_sum_rel"] = max(DEV.get("k1_large_sum_rel", 0.0), float(rel[~small].max()))
            DEV["k1_large_sum_abs"] = max(DEV.get("k1_large_sum_abs", 0.0), float(d[~small].max()))
            assert rel[~small].max() <= 1e-6, (tag, rel[~small].max())
        if small.sum() > 200:
            assert (p[small] == w[small]).mean() >= 0.97, (tag, (p[small] == w[small]).mean())
        DEV["k1_rows"] = DEV.get("k1_rows", 0) + int(ns)
        DEV["k1_equal"] = DEV.get("k1_equal", 0) + int((p == w).sum())


def batch_case():
    """kmd_merge_filter_batch over 2..10 random partitions (some empty, some with two-limb k-mers apart), sinks shared or
    not, against single kmd_merge_filter calls: bit for bit"""
    S = int(rng.choice([3, 12, 40, 80]))
    nc = int(rng.integers(1, S))
    sets, hosts, tot = [], [], np.zeros(S, dtype=np.uint64)
    for j in range(int(rng.integers(2, 11))):
        n = 0 if rng.random() < 0.1 else int(rng.integers(1, 30000 if S <= 12 else 6000))
        universe = np.unique(rng.integers(0, 1 << 62, n, dtype=np.uint64))
        streams = make_streams(rng, universe, S, rng.uniform(0.02, 1.0, S), count_hi=int(rng.integers(2, 400)))
        sets.append(K.StreamSet(streams))
        hosts.append(streams)
        tot += np.array([int(t[1].sum(dtype=np.uint64)) for t in streams], dtype=np.uint64)
    if int(tot[:nc].sum()) == 0 or int(tot[nc:].sum()) == 0:
        return
    model = K.PoissonLikelihood(nc, S - nc, tot[:nc], tot[nc:], 10000)
    thr = float(rng.choice([0.5, 1e-2, 1e-5]))
    single = []
    for ss in sets:
        acc = K.SurvivorAccumulator(1 << 16)
        rows = K.merge_filter(ss, K.diff_observer(model, acc, thr)) if ss.total else 0
        single.append((rows, acc.finish(by_kmer=True), acc.get(), [int(x) for x in acc.read_counters()[:4]]))
    shared = rng.random() < 0.5
    if shared:
        one = K.SurvivorAccumulator(1 << 19)
        accs = [one] * len(sets)
    else:
        accs = [K.SurvivorAccumulator(1 << 16) for _ in sets]
    rows_b = K.merge_filter_batch(sets, [K.diff_observer(model, x, thr) for x in accs])
    assert [int(r) for r in rows_b] == [s_[0] for s_ in single]
    if shared:
        n = one.finish(by_kmer=True)
        got = one.get()
        allk = np.concatenate([s_[2]["kmer_lo"] for s_ in single]) if n else np.zeros(0, np.uint64)
        allp = np.concatenate([s_[2]["pvalue"] for s_ in single]) if n else np.zeros(0)
        order = np.argsort(allk, kind="stable")
        assert n == len(allk) and got["kmer_lo"].tolist() == allk[order].tolist()
        # (the same k-mer may be in two partitions of this soak: compare as multisets of (k-mer, p))
        assert sorted(zip(got["kmer_lo"].tolist(), got["pvalue"].tolist())) == sorted(zip(allk.tolist(), allp.tolist()))
        assert [int(x) for x in one.read_counters()[:4]] == [sum(s_[3][i] for s_ in single) for i in range(4)]
    else:
        for j, x in enumerate(accs):
            n = x.finish(by_kmer=True)
            got = x.get()
            assert n == single[j][1] and [int(v) for v in x.read_counters()[:4]] == single[j][3], j
            for key in ("kmer_lo", "pvalue", "sign", "mean_control", "mean_case"):
                assert got[key].tolist() == single[j][2][key].tolist(), (j, key)


def popstrat_case():
    """kmd_popstrat_apply (pop_strat_corrector::apply + glm_irls) on random designs: 2..60 samples a side, 2..10 principal
    components, iteration limits, effect sizes from none to separation: tests/test_gpu_popstrat.check's bars"""
    from test_gpu_popstrat import check as ps_check, count_rows
    nc, nk = int(rng.integers(2, 61)), int(rng.integers(2, 61))
    npc = int(rng.integers(2, 11))
    n, effect = int(rng.integers(1, 3000)), float(rng.choice([1.0, 1.2, 3.0, 30.0]))
    rows = count_rows(rng, n, nc, nk, effect=effect)
    sparse = rng.random() < 0.3
    if sparse:
        rows[rng.random(rows.shape) < 0.7] = 0
    zs, zshift = float(rng.choice([0.01, 0.1, 1.0])), float(rng.choice([0.0, 0.05, 0.5]))
    Z = rng.normal(0, zs, size=(nc + nk, 10))
    Z[:nc, 0] += zshift
    stand, max_iter, scale, seed = bool(rng.integers(0, 2)), int(rng.choice([0, 0, 1, 2, 5, 25])), float(rng.choice([1.0, 0.01, 100.0])), int(rng.integers(0, 1 << 30))
    if a.popstrat_stand != "both":
        stand = a.popstrat_stand == "true"
    libm_rows, outside = [], []
    try:
        ps_check(K, oracle, nc, nk, npc, stand, max_iter, rows, Z=Z, totals_scale=scale, seed=seed, want_spread=False, libm_rows=libm_rows,
                 outside=outside if a.tally else None)
    except AssertionError:
        np.savez("gpurun_out/soak_ps_fail.npz", nc=nc, nk=nk, npc=npc, stand=stand, max_iter=max_iter, rows=rows, Z=Z, scale=scale, seed=seed)
        print("FAILED pop-strat: nc=%d nk=%d npc=%d stand=%s max_iter=%d n=%d effect=%g sparse=%s zs=%g zshift=%g scale=%g seed=%d"
              % (nc, nk, npc, stand, max_iter, n, effect, sparse, zs, zshift, scale, seed), flush=True)
        raise
    if outside and outside[0][0] == -1:       # --tally: a null fit that differs by more than the jitter explains
        tag = "stand" if stand else "nostand"
        DEV["popstrat_unexplained_null_fits_" + tag] = DEV.get("popstrat_unexplained_null_fits_" + tag, 0) + 1
        print("  unexplained null fit (%s): nc=%d nk=%d npc=%d max_iter=%d scale=%g: |dev - oracle| %.3g of weights up to %.3g, jitter spread %.3g"
              % (tag, nc, nk, npc, max_iter, scale, outside[0][1], outside[0][2], outside[0][3]), flush=True)
        return
    if -1 in libm_rows:                       # the null fit itself hangs on libm's last bit: the design is counted, its rows are not compared
        key = "popstrat_libm_designs_stand" if stand else "popstrat_libm_designs_nostand"
        DEV[key] = DEV.get(key, 0) + 1
        return
    if outside:
        # rows outside the bars that one ulp of the oracle's pow() does not explain (--tally: counted, with the worst of them)
        tag = "stand" if stand else "nostand"
        if nc + nk < npc + 3:
            # fewer samples than columns of the alternative design (F = 3 + npc): X^T G X is singular by construction, the
            # no-pivot LU divides by rounding noise, and the p-value is noise in the reference and here alike (with fewer
            # samples than NULL columns the reference has already written past its stddev vector: undefined behaviour there)
            tag += "_rank_deficient"
        DEV["popstrat_unexplained_rows_" + tag] = DEV.get("popstrat_unexplained_rows_" + tag, 0) + len(outside)
        DEV["popstrat_unexplained_designs_" + tag] = DEV.get("popstrat_unexplained_designs_" + tag, 0) + 1
        nd = DEV["popstrat_unexplained_designs_" + tag]
        worst = max(outside, key=lambda t: abs(t[1] - t[2]))
        print("  unexplained rows (%s): %d of %d in nc=%d nk=%d npc=%d max_iter=%d effect=%g sparse=%s zs=%g zshift=%g scale=%g seed=%d; worst p_dev %.17g p_ref %.17g jitter [%.17g, %.17g]"
              % (tag, len(outside), n, nc, nk, npc, max_iter, effect, sparse, zs, zshift, scale, seed, worst[1], worst[2], worst[3], worst[4]), flush=True)
        if nd <= 12:
            idx = np.array([t[0] for t in outside])
            np.savez("gpurun_out/soak_ps_unexplained_%s_%d.npz" % (tag, nd), nc=nc, nk=nk, npc=npc, stand=stand, max_iter=max_iter, rows=rows[idx], Z=Z, scale=scale, seed=seed,
                     p_dev=np.array([t[1] for t in outside]), p_ref=np.array([t[2] for t in outside]), idx=idx, n_rows=n)
        for (i, pd, pr, lo, hi) in outside:
            d_abs = abs(pd - pr)
            if d_abs > DEV.get("popstrat_unexplained_max_abs_" + tag, (0.0,))[0]:
                DEV["popstrat_unexplained_max_abs_" + tag] = (d_abs, pd, pr)
            d_rel = d_abs / abs(pr) if pr else float("inf")
            if pr <= 0.05 and d_rel > DEV.get("popstrat_unexplained_max_rel_at_p_le_0.05_" + tag, (0.0,))[0]:
                DEV["popstrat_unexplained_max_rel_at_p_le_0.05_" + tag] = (d_rel, pd, pr)
            if pr <= 0.05:
                DEV["popstrat_unexplained_rows_p_le_0.05_" + tag] = DEV.get("popstrat_unexplained_rows_p_le_0.05_" + tag, 0) + 1
    DEV["popstrat_rows"] = DEV.get("popstrat_rows", 0) + n
    DEV["popstrat_rows_stand" if stand else "popstrat_rows_nostand"] = DEV.get("popstrat_rows_stand" if stand else "popstrat_rows_nostand", 0) + n
    DEV["popstrat_designs_stand" if stand else "popstrat_designs_nostand"] = DEV.get("popstrat_designs_stand" if stand else "popstrat_designs_nostand", 0) + 1
    DEV["popstrat_libm_rows"] = DEV.get("popstrat_libm_rows", 0) + len(libm_rows)
    if libm_rows:
        key = "popstrat_libm_rows_stand" if stand else "popstrat_libm_rows_nostand"
        DEV[key] = DEV.get(key, 0) + len(libm_rows)


n_ps = 0
n_cases = 0
while time.time() - t0 < a.seconds and (a.cases == 0 or n_cases < a.cases):
    n_cases += 1
    kind = rng.integers(0, 19)
    if a.only == "popstrat":
        kind = 16
    if kind >= 16:
        popstrat_case()
        n_ps += 1
    elif kind >= 13:
        batch_case()
        n_batch += 1
    elif kind >= 10:
        k1_case()
        n_k1 += 1
    elif kind < 6:
        S = int(rng.choice([2, 3, 5, 8, 16, 40, 41, 64, 65, 100, 200, 260]))
        nc = int(rng.integers(1, S))
        n = int(rng.integers(0, 60000 if S <= 64 else 8000))
        mode = rng.integers(0, 4)
        if mode == 0:
            base = int(rng.integers(0, 1 << 60))
            universe = np.unique(np.uint64(base) + rng.integers(0, 3 * n + 8, n).astype(np.uint64))
        elif mode == 1:                                    # a few dense clusters
            starts = rng.integers(0, 1 << 61, 7, dtype=np.uint64)
            universe = np.unique((starts[:, None] + rng.integers(0, 4 * (n // 7 + 1), (7, n // 7 + 1)).astype(np.uint64)).ravel())
        else:
            universe = np.unique(rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64))
        if rng.random() < 0.15 and len(universe):
            universe = np.unique(np.concatenate([universe, np.array([0, 2 ** 64 - 1], dtype=np.uint64)]))
        pres = rng.uniform(0.01, 1.0, S) if rng.random() < 0.5 else np.full(S, rng.uniform(0.02, 0.9))
        empty = tuple(int(x) for x in rng.choice(S, size=int(rng.integers(0, max(1, S // 3))), replace=False)) if S > 2 else ()
        two = rng.random() < 0.25                          # two-limb k-mers (32 < k <= 64): a few high limbs, sorted (hi, lo)
        hi = None
        if two and len(universe):
            hi = np.sort(rng.integers(0, int(rng.choice([2, 5, 1 << 62])), len(universe), dtype=np.uint64))
            order = np.lexsort((universe, hi))
            universe, hi = universe[order], hi[order]
        else:
            two = False
        streams = make_streams(rng, universe, S, pres, count_hi=int(rng.integers(2, 3000)), empty=empty, hi=hi)
        if sum(len(t[0]) for t in streams) == 0:
            continue
        if rng.random() < 0.2:
            for s in range(S):
                km, cnt = streams[s][0], streams[s][1]
                if len(cnt):
                    cnt = cnt.copy()
                    cnt[rng.integers(0, len(cnt), 2)] = np.uint32(3_000_000_000)
                    streams[s] = (km, cnt) + tuple(streams[s][2:])
        want = oracle.merge_partition2(streams)[0] if two else oracle.merge_partition(streams)[0]
        tcs, tks = totals_of(want, nc)
        if int(tcs.sum()) == 0 or int(tks.sum()) == 0:
            continue
        thr, lf_n = float(rng.choice([1.0, 0.3, 1e-2, 1e-4, 1e-6, 1e-9])), int(rng.choice([10000, 300, 50]))
        try:
            fused_case(streams, nc, thr, lf_n, two=two)
        except AssertionError:
            np.savez("gpurun_out/soak_fail.npz", nc=nc, thr=thr, lf_n=lf_n, **{"k%d" % i: t[0] for i, t in enumerate(streams)},
                     **{"c%d" % i: t[1] for i, t in enumerate(streams)}, **({"h%d" % i: t[2] for i, t in enumerate(streams)} if two else {}))
            print("FAILED: S=%d nc=%d n=%d mode=%d thr=%g lf_n=%d count max %d" % (S, nc, n, mode, thr, lf_n, max(int(t[1].max(initial=0)) for t in streams)), flush=True)
            raise
        n_fused += 1
    elif kind < 8:
        streams = []
        for _ in range(int(rng.integers(1, 12))):
            n = int(rng.integers(0, 3000))
            bits = int(rng.integers(0, 65))
            if bits == 64:
                km = rng.integers(0, 2 ** 64, n, dtype=np.uint64)
            elif bits == 0:
                km = np.full(n, int(rng.integers(0, 2 ** 63)), dtype=np.uint64)
            else:
                km = np.cumsum(rng.integers(0, 1 << min(bits, 52), n, dtype=np.uint64), dtype=np.uint64)
            ct = rng.integers(0, int(rng.choice([3, 254, 256, 70000, 2 ** 32])), n).astype(np.uint32)
            streams.append((km, ct))
        packed, base, table, offs = K.pack_streams(streams)
        ss = K.unpack_streams(packed, base, table, offs)
        km, ct = ss.kmers.to_host(np.uint64, ss.total), ss.counts.to_host(np.uint32, ss.total)
        assert np.array_equal(km, np.concatenate([s[0] for s in streams])) and np.array_equal(ct, np.concatenate([s[1] for s in streams]))
        n_pack += 1
    else:
        n = int(rng.integers(0, 5000))
        p = rng.uniform(0, 1, n) ** int(rng.integers(1, 40))
        if n > 10 and rng.random() < 0.5:
            p[rng.integers(0, n, n // 5)] = p[rng.integers(0, n)]            # ties
        s = rng.integers(0, 3, n).astype(np.int32)
        world = int(rng.integers(2, 10))
        cuts = [0] + sorted(rng.integers(0, n + 1, world - 1).tolist()) + [n]
        parts = [(p[x:y], s[x:y]) for x, y in zip(cuts[:-1], cuts[1:])]
        total = int(rng.choice([max(n, 1), 10 * max(n, 1), 10 ** 7, 10 ** 10]))
        name = str(rng.choice(["benjamini", "holm", "bonferroni", "sidak", "nothing"]))
        thr = float(rng.choice([0.05, 0.5, 1e-3]))
        T = (N.Transport * world)()
        N.check(N.lib().kmd_transport_local_create(world, T))
        try:
            want, _, _ = K.aggregate(name, thr, total, K.DeviceBuffer.from_host(p), K.DeviceBuffer.from_host(s), n)
            per = [total // world + (1 if r < total % world else 0) for r in range(world)]
            res = sharded_decisions(K, T, name, per, parts, thr=thr)
            assert np.concatenate([o[1] for o in res]).tolist() == want.tolist(), (name, world, n, total, thr)
        finally:
            N.lib().kmd_transport_local_destroy(world, T)
        n_shard += 1
    if (n_fused + n_pack + n_shard + n_k1 + n_batch) % 50 == 0:
        print("  %.0f s: %d fused, %d pack, %d sharded, %d matrix, %d batch cases" % (time.time() - t0, n_fused, n_pack, n_shard, n_k1, n_batch), flush=True)
print("p-values vs the oracle:", DEV)
if a.tally:
    # --tally counts instead of stopping -- but it can still FAIL: on standardised designs (the only mode the
    # reference can reach) the rows outside the bars that one ulp of the oracle's own pow() does not explain stayed at
    # 0.04 % of 7.5 M rows in round 5, every one of them at p >= 0.33.  A regression of K3 shows up as more of them, or
    # as one at a p-value that decides something.
    rows_stand = DEV.get("popstrat_rows_stand", 0)
    unexplained = DEV.get("popstrat_unexplained_rows_stand", 0)
    assert unexplained <= max(3, 2e-3 * rows_stand), ("unexplained pop-strat rows on standardised designs", unexplained, rows_stand)
    assert DEV.get("popstrat_unexplained_rows_p_le_0.05_stand", 0) == 0, DEV.get("popstrat_unexplained_max_rel_at_p_le_0.05_stand")
    assert DEV.get("popstrat_unexplained_null_fits_stand", 0) <= max(1, DEV.get("popstrat_designs_stand", 0) // 100)
print("soak ok: %d fused merge cases, %d pack round trips, %d sharded corrections, %d matrix filters, %d batches, %d pop-strat designs in %.0f s (seed %d)"
      % (n_fused, n_pack, n_shard, n_k1, n_batch, n_ps, time.time() - t0, a.seed))
