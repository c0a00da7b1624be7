"""ctypes loader for the CPU oracle (oracle/liboracle.so) -- test infrastructure only.
Builds it with `make -C oracle liboracle.so` when missing (gcc is on every box)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "oracle", "liboracle.so")

LAYOUT_ROWS, LAYOUT_SOA = 0, 1


class Counters(C.Structure):
    _fields_ = [("total", C.c_uint64), ("n_sig", C.c_uint64), ("n_sig_control", C.c_uint64),
                ("n_sig_case", C.c_uint64)]


class Oracle:
    def __init__(self, L):
        self.L = L
        d, vp, sz, u64, i = C.c_double, C.c_void_p, C.c_size_t, C.c_uint64, C.c_int
        L.kmdo_lf_build.argtypes = [sz, vp]
        L.kmdo_lf_at.restype = d
        L.kmdo_lf_at.argtypes = [vp, sz, u64]
        L.kmdo_chisqc.restype = d
        L.kmdo_chisqc.argtypes = [d, d]
        L.kmdo_igamc.restype = d
        L.kmdo_igamc.argtypes = [d, d]
        L.kmdo_poisson_rows.argtypes = [vp, i, i, sz, sz, i, i, u64, u64, vp, sz, vp, vp, vp, vp]
        L.kmdo_diff_partition.restype = sz
        L.kmdo_diff_partition.argtypes = [vp, i, i, sz, sz, i, i, u64, u64, vp, sz, d, vp, vp, vp, vp, vp,
                                          sz, C.POINTER(Counters)]
        L.kmdo_bench_partitions.restype = d
        L.kmdo_bench_partitions.argtypes = [u64, i, sz, i, i, i, u64, u64, sz, d, i, C.POINTER(Counters)]
        L.kmdo_aggregate.restype = sz
        L.kmdo_aggregate.argtypes = [i, d, u64, vp, sz, vp]
        L.kmdo_synth_rows.argtypes = [u64, C.c_uint32, u64, sz, i, i, i, i, sz, vp, vp, vp]
        L.kmdo_sigmoid.restype = d
        L.kmdo_sigmoid.argtypes = [d]
        L.kmdo_sigmoid_jitter.restype = None
        L.kmdo_sigmoid_jitter.argtypes = [C.c_uint64, i]
        L.kmdo_lu.argtypes = [vp, i, vp, vp]
        L.kmdo_inverse.restype = i
        L.kmdo_inverse.argtypes = [vp, i, vp]
        L.kmdo_glm_irls.restype = i
        L.kmdo_glm_irls.argtypes = [vp, vp, i, i, i, vp, vp, vp]
        L.kmdo_popstrat_pvalue.restype = d
        L.kmdo_popstrat_pvalue.argtypes = [vp, i, i, vp, vp, vp, vp, i]

        L.kmdo_merge_partition.restype = sz
        L.kmdo_merge_partition.argtypes = [i, vp, vp, vp, vp, vp, sz]
        L.kmdo_merge_partition2.restype = sz
        L.kmdo_merge_partition2.argtypes = [i, vp, vp, vp, vp, vp, vp, vp, sz]
        L.kmdo_popstrat_features.argtypes = [i, i, vp, vp, vp, i, i, i, vp, vp, vp]

        class Corr(C.Structure):
            _fields_ = [("type", i), ("threshold", d), ("total", u64), ("rank", u64)]
        self.Corr = Corr
        L.kmdo_corrector_init.argtypes = [C.POINTER(Corr), i, d, u64]
        L.kmdo_corrector_apply.argtypes = [C.POINTER(Corr), d]

    # ---- R5
    def lf_table(self, size):
        t = np.zeros(max(size, 1))
        self.L.kmdo_lf_build(size, t.ctypes.data)
        return t[:size] if size else t[:0]

    def lf_at(self, table, k):
        t = np.ascontiguousarray(table, dtype=np.float64)
        return self.L.kmdo_lf_at(t.ctypes.data if len(t) else None, len(t), int(k))

    # ---- R6
    def chisqc(self, v, x):
        return self.L.kmdo_chisqc(float(v), float(x))

    # ---- R4
    def poisson_rows(self, counts, layout, nc, nk, tc, tk, lf):
        counts = np.ascontiguousarray(counts)
        if layout == LAYOUT_ROWS:
            n, ld = counts.shape[0], counts.shape[1]
        else:
            n, ld = counts.shape[1], counts.shape[1]
        lf = np.ascontiguousarray(lf, dtype=np.float64)
        p = np.zeros(n)
        s = np.zeros(n, dtype=np.int32)
        mc = np.zeros(n)
        mk = np.zeros(n)
        self.L.kmdo_poisson_rows(counts.ctypes.data, counts.dtype.itemsize, layout, ld, n, nc, nk, int(tc),
                                 int(tk), lf.ctypes.data if len(lf) else None, len(lf), p.ctypes.data,
                                 s.ctypes.data, mc.ctypes.data, mk.ctypes.data)
        return p, s, mc, mk

    # ---- R3
    def diff_partition(self, counts, layout, nc, nk, tc, tk, lf, threshold, cap=None):
        counts = np.ascontiguousarray(counts)
        if layout == LAYOUT_ROWS:
            n, ld = counts.shape[0], counts.shape[1]
        else:
            n, ld = counts.shape[1], counts.shape[1]
        cap = n if cap is None else cap
        lf = np.ascontiguousarray(lf, dtype=np.float64)
        row = np.zeros(max(cap, 1), dtype=np.uint64)
        p = np.zeros(max(cap, 1))
        s = np.zeros(max(cap, 1), dtype=np.int32)
        mc = np.zeros(max(cap, 1))
        mk = np.zeros(max(cap, 1))
        c = Counters()
        ns = self.L.kmdo_diff_partition(counts.ctypes.data, counts.dtype.itemsize, layout, ld, n, nc, nk,
                                        int(tc), int(tk), lf.ctypes.data if len(lf) else None, len(lf),
                                        float(threshold), row.ctypes.data, p.ctypes.data, s.ctypes.data,
                                        mc.ctypes.data, mk.ctypes.data, cap, C.byref(c))
        k = min(ns, cap)
        return {"row": row[:k], "pvalue": p[:k], "sign": s[:k], "mean_control": mc[:k], "mean_case": mk[:k],
                "counters": (c.total, c.n_sig, c.n_sig_control, c.n_sig_case)}

    # ---- R8
    def corrector(self, ctype, threshold, total):
        c = self.Corr()
        self.L.kmdo_corrector_init(C.byref(c), ctype, threshold, total)
        return c

    def corrector_apply(self, c, p):
        return self.L.kmdo_corrector_apply(C.byref(c), float(p))

    def aggregate(self, ctype, threshold, total, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        keep = np.zeros(max(len(p), 1), dtype=np.uint8)
        self.L.kmdo_aggregate(ctype, float(threshold), int(total), p.ctypes.data, len(p), keep.ctypes.data)
        return keep[:len(p)]

    # ---- R1
    def merge_partition(self, streams):
        S = len(streams)
        offs = np.zeros(S + 1, dtype=np.uint64)
        for s, (k, c) in enumerate(streams):
            offs[s + 1] = offs[s] + len(k)
        total = int(offs[-1])
        kmers = np.concatenate([np.asarray(k, dtype=np.uint64) for k, _ in streams]) if total else np.zeros(1, np.uint64)
        counts = np.concatenate([np.asarray(c, dtype=np.uint32) for _, c in streams]) if total else np.zeros(1, np.uint32)
        mat = np.zeros((max(total, 1), S), dtype=np.uint32)
        ko = np.zeros(max(total, 1), dtype=np.uint64)
        n = self.L.kmdo_merge_partition(S, kmers.ctypes.data, counts.ctypes.data, offs.ctypes.data,
                                        mat.ctypes.data, ko.ctypes.data, max(total, 1))
        return mat[:n].copy(), ko[:n].copy()

    def merge_partition2(self, streams):
        """streams of (kmers_lo, counts, kmers_hi)."""
        S = len(streams)
        offs = np.zeros(S + 1, dtype=np.uint64)
        for s, t in enumerate(streams):
            offs[s + 1] = offs[s] + len(t[0])
        total = max(int(offs[-1]), 1)
        cat = lambda i, dt: np.concatenate([np.asarray(t[i], dtype=dt) for t in streams] + [np.zeros(1, dt)])
        lo, ct, hi = cat(0, np.uint64), cat(1, np.uint32), cat(2, np.uint64)
        mat = np.zeros((total, S), dtype=np.uint32)
        klo = np.zeros(total, dtype=np.uint64)
        khi = np.zeros(total, dtype=np.uint64)
        n = self.L.kmdo_merge_partition2(S, lo.ctypes.data, hi.ctypes.data, ct.ctypes.data, offs.ctypes.data,
                                         mat.ctypes.data, klo.ctypes.data, khi.ctypes.data, total)
        return mat[:n].copy(), klo[:n].copy(), khi[:n].copy()

    # ---- R9
    def popstrat_setup(self, nc, nk, totals_c, totals_k, Z, npc, standardize=True, max_iter=100):
        """init_global_features + standardize + null-model fit.  Returns (alt, null_model, totals)."""
        n, fn = nc + nk, 2 + npc
        tc = np.ascontiguousarray(totals_c, dtype=np.uint64)
        tk = np.ascontiguousarray(totals_k, dtype=np.uint64)
        Z = np.ascontiguousarray(Z, dtype=np.float64)
        nul = np.zeros((n, fn))
        alt = np.zeros((n, fn + 1))
        tot = np.zeros(n)
        self.L.kmdo_popstrat_features(nc, nk, tc.ctypes.data, tk.ctypes.data, Z.ctypes.data, Z.shape[1], npc,
                                      int(standardize), nul.ctypes.data, alt.ctypes.data, tot.ctypes.data)
        y = np.concatenate([np.ones(nc), np.zeros(nk)])
        w = np.zeros(fn)
        self.L.kmdo_glm_irls(nul.ctypes.data, y.ctypes.data, n, fn, max_iter, w.ctypes.data, None, None)
        return alt, w, tot, y

    def popstrat_pvalues(self, alt, null_model, totals, y, counts_rows, max_iter=100):
        n, f = alt.shape
        out = np.zeros(len(counts_rows))
        for k, c in enumerate(np.ascontiguousarray(counts_rows, dtype=np.float64)):
            out[k] = self.L.kmdo_popstrat_pvalue(alt.ctypes.data, n, f, y.ctypes.data, totals.ctypes.data,
                                                 c.ctypes.data, null_model.ctypes.data, max_iter)
        return out

    # ---- synthetic
    def synth_rows(self, seed, part, row0, n, nc, nk, count_bytes=4, kmer_limbs=1):
        dt = {1: np.uint8, 2: np.uint16, 4: np.uint32}[count_bytes]
        counts = np.zeros((n, nc + nk), dtype=dt)
        lo = np.zeros(max(n, 1), dtype=np.uint64)
        hi = np.zeros(max(n, 1), dtype=np.uint64) if kmer_limbs == 2 else None
        self.L.kmdo_synth_rows(int(seed), int(part), int(row0), n, nc, nk, count_bytes, LAYOUT_ROWS, nc + nk,
                               counts.ctypes.data, lo.ctypes.data, hi.ctypes.data if hi is not None else None)
        return counts, lo[:n], (hi[:n] if hi is not None else None)


_inst = None


def load():
    global _inst
    if _inst is None:
        src = os.path.join(ROOT, "oracle", "kmd_oracle.c")
        if (not os.path.exists(PATH)) or os.path.getmtime(PATH) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"],
                                  stdout=subprocess.DEVNULL)
        _inst = Oracle(C.CDLL(PATH))
    return _inst
