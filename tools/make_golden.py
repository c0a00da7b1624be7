#!/usr/bin/env python3
"""Generate tests/golden/*.json from the REFERENCE's own arithmetic (oracle/_ref: alglib,
src/log_factorial_table.cpp, src/corrector.cpp compiled from /root/reference by
oracle/Makefile).  Runs only in the build container (needs oracle/_ref/libkmdiff_ref.so);
the JSON files are committed and are what travels.  Doubles are stored as C99 hex floats so
the fixtures are exact.

    make -C oracle ref && python3 tools/make_golden.py
"""
import ctypes as C
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = C.CDLL(os.path.join(ROOT, "oracle/_ref/libkmdiff_ref.so"))
R.kmdref_chisqc.restype = C.c_double
R.kmdref_chisqc.argtypes = [C.c_double, C.c_double]
R.kmdref_lf.argtypes = [C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
R.kmdref_corrector_new.restype = C.c_void_p
R.kmdref_corrector_new.argtypes = [C.c_int, C.c_double, C.c_uint64]
R.kmdref_corrector_apply.argtypes = [C.c_void_p, C.c_double]
R.kmdref_corrector_free.argtypes = [C.c_void_p]
R.kmdref_model_new.restype = C.c_void_p
R.kmdref_model_new.argtypes = [C.c_size_t] * 3 + [C.c_uint64] * 2
R.kmdref_model_process.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t] + [C.c_void_p] * 4
R.kmdref_model_free.argtypes = [C.c_void_p]


def hx(a):
    return [float(v).hex() for v in a]


def dump(name, obj):
    p = os.path.join(ROOT, "tests/golden", name)
    with open(p, "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print(name, os.path.getsize(p), "bytes")


def main():
    rng = np.random.default_rng(0x6B6D64696666)
    # R6: alglib::chisquarecdistribution(v, x)
    xs = np.concatenate([[0.0, 1e-300, 1e-12, 0.5, 1.0, 1.9999999, 2.0, 2.0000001, 25.2, 25.4, 30.0,
                          1418.0, 1419.5, 1419.6, 1430.0, 2000.0],
                         rng.uniform(0, 4, 300), rng.uniform(0, 120, 500), rng.uniform(0, 1500, 300),
                         10.0 ** rng.uniform(-15, 3.2, 300)])
    dump("chisqc_v1.json", {"source": "alglib::chisquarecdistribution(1,x), specialfunctions.cpp:2750",
                            "x": hx(xs), "p": hx([R.kmdref_chisqc(1.0, x) for x in xs])})
    vs = [2.0, 3.0, 5.0, 11.0, 30.0]
    xs2 = rng.uniform(0, 150, 200)
    dump("chisqc_vN.json", {"v": vs, "x": hx(xs2),
                            "p": [hx([R.kmdref_chisqc(v, x) for x in xs2]) for v in vs]})
    # R5: LogFactorialTable(size)[i]
    lf = {}
    for size in (0, 1, 2, 50, 1000):
        idx = np.unique(np.concatenate([[0, 1, 2, 3, 10, 49, 50, 51, 100, 999, 1000, 1001, 5000, 20000],
                                        rng.integers(0, 3000, 40)])).astype(np.uint64)
        out = np.zeros(len(idx))
        R.kmdref_lf(size, idx.ctypes.data, len(idx), out.ctypes.data)
        lf[str(size)] = {"i": [int(i) for i in idx], "lf": hx(out)}
    dump("log_factorial.json", {"source": "LogFactorialTable, src/log_factorial_table.cpp:5-22", "tables": lf})
    # R8: corrector decision streams (ascending, as the reference feeds the stateful ones)
    cor = []
    for ctype, name in ((0, "nothing"), (1, "bonferroni"), (2, "benjamini"), (3, "sidak"), (4, "holm")):
        for thr, total in ((0.05, 100), (0.05, 100000), (0.25, 25), (0.01, 3_000_000_000), (0.05, 7)):
            ps = np.sort(np.concatenate([10.0 ** rng.uniform(-14, 0, 60),
                                         [thr / total, thr, 0.0, 1 - (1 - thr) ** (1.0 / total)]]))
            h = R.kmdref_corrector_new(ctype, thr, total)
            dec = [int(R.kmdref_corrector_apply(C.c_void_p(h), float(p))) for p in ps]
            R.kmdref_corrector_free(C.c_void_p(h))
            cor.append({"type": ctype, "name": name, "threshold": float(thr).hex(), "total": int(total),
                        "p": hx(ps), "apply": dec})
    dump("correctors.json", {"source": "make_corrector/ICorrector::apply, src/corrector.cpp:6-116", "cases": cor})
    # R4: PoissonLikelihood::process glue around the real LogFactorialTable + alglib
    cases = []

    def model_case(pre, nc, nk, tcs, tks, rows):
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        m = R.kmdref_model_new(pre, nc, nk, int(sum(int(t) for t in tcs)), int(sum(int(t) for t in tks)))
        n = rows.shape[0]
        p = np.zeros(n)
        s = np.zeros(n, dtype=np.int32)
        mc = np.zeros(n)
        mk = np.zeros(n)
        R.kmdref_model_process(C.c_void_p(m), rows.ctypes.data, n, p.ctypes.data, s.ctypes.data,
                               mc.ctypes.data, mk.ctypes.data)
        R.kmdref_model_free(C.c_void_p(m))
        cases.append({"preload": pre, "nc": nc, "nk": nk, "total_controls": [int(t) for t in tcs],
                      "total_cases": [int(t) for t in tks], "rows": rows.tolist(), "p": hx(p),
                      "sign": s.tolist(), "mean_control": hx(mc), "mean_case": hx(mk)})

    # the reference's own test inputs (tests/model_test.cpp:45-81) and SURVEY 8c known answers
    model_case(10, 30, 30, [1] * 30, [1] * 30, [[200] * 30 + [100] * 30, [100] * 30 + [200] * 30, [100] * 60])
    model_case(10000, 4, 4, [10 ** 9] * 4, [12 * 10 ** 8] * 4, [[10, 12, 9, 11, 30, 28, 35, 31]])
    model_case(100, 2, 2, [1000] * 2, [1000] * 2,
               [[0, 0, 50, 70], [60, 70, 0, 0], [5, 5, 5, 5], [100, 150, 5, 5]])
    # random rows at four shapes, sums straddling the table size
    for (pre, nc, nk, lam) in ((10000, 4, 4, 6.0), (10000, 20, 20, 9.0), (64, 20, 20, 4.0), (500, 3, 5, 80.0)):
        S = nc + nk
        rows = rng.poisson(lam, size=(120, S)).astype(np.uint32)
        rows[::7, nc:] *= 3
        rows[::11, :nc] *= 4
        rows[5] = 0
        rows[5, 1] = 1
        tcs = rng.integers(10 ** 8, 2 * 10 ** 9, nc)
        tks = rng.integers(10 ** 8, 2 * 10 ** 9, nk)
        model_case(pre, nc, nk, tcs, tks, rows)
    dump("poisson_rows.json", {"source": "PoissonLikelihood::process (model.hpp:142-176) glue restated in "
                               "oracle/ref_shim.cpp around the reference's LogFactorialTable and alglib",
                               "cases": cases})


if __name__ == "__main__":
    main()
