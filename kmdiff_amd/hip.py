"""Host-side mirror of the reference's operator interface for the `diff` hot path, over the
C-ABI (include/kmdiff_hip.h).  Names and argument meaning follow the reference:

  PoissonLikelihood(nb_controls, nb_cases, total_controls, total_cases, preload)
                                              include/kmdiff/model.hpp:106-118
  diff_observer(model, acc, threshold, ...).process(matrix)
                                              include/kmdiff/merge.hpp:44-131 (whole tile at once)
  SurvivorAccumulator                         include/kmdiff/accumulator.hpp:36-54 (IAccumulator)
  aggregate(correction, threshold, total_kmers, acc)
                                              src/corrector.cpp:101-116 + include/kmdiff/aggregator.hpp:343-365

Everything numeric happens inside libkmdiff_hip.so; this module only owns buffers and
copies results out.  numpy is used for host arrays only.
"""
import ctypes as C

import numpy as np

from . import _native as N
from ._native import KmdError, check, lib

CORRECTION_BY_NAME = {"nothing": N.CORR_NOTHING, "disabled": N.CORR_NOTHING,
                      "bonferroni": N.CORR_BONFERRONI, "benjamini": N.CORR_BENJAMINI,
                      "sidak": N.CORR_SIDAK, "holm": N.CORR_HOLM}

_COUNT_DTYPES = {1: np.uint8, 2: np.uint16, 4: np.uint32}
TILED_BLOCK_ROWS = 4096     # T of LAYOUT_TILED: rows per block (multiple of 4096)


def device_count():
    n = C.c_int(0)
    lib().kmd_device_count(C.byref(n))
    return n.value


def device_name():
    buf = C.create_string_buffer(256)
    check(lib().kmd_device_name(buf, 256), "kmd_device_name")
    return buf.value.decode()


def _require_device():
    if device_count() < 1:
        raise KmdError("no HIP device visible: the kmdiff hot path has no CPU fallback")


class DeviceBuffer:
    """A block of HBM owned through kmd_malloc/kmd_free."""

    def __init__(self, nbytes):
        _require_device()
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(lib().kmd_malloc(C.byref(p), self.nbytes), "kmd_malloc(%d)" % self.nbytes)
        self.ptr = p.value

    @classmethod
    def from_host(cls, arr):
        arr = np.ascontiguousarray(arr)
        b = cls(arr.nbytes)
        if arr.nbytes:
            check(lib().kmd_memcpy_h2d(b.ptr, arr.ctypes.data, arr.nbytes, None), "h2d")
        return b

    def zero(self):
        check(lib().kmd_memset(self.ptr, 0, self.nbytes, None), "memset")
        check(lib().kmd_stream_sync(None), "sync")          # (the buffer may next be used on a stream that is not ordered against the null stream)
        return self

    def to_host(self, dtype, count=None, offset_bytes=0):
        dtype = np.dtype(dtype)
        if count is None:
            count = (self.nbytes - offset_bytes) // dtype.itemsize
        out = np.empty(count, dtype=dtype)
        if out.nbytes:
            check(lib().kmd_memcpy_d2h(out.ctypes.data, self.ptr + offset_bytes, out.nbytes, None), "d2h")
        return out

    def free(self):
        if getattr(self, "ptr", None):
            lib().kmd_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Event:
    """HIP event on a stream of the library (for timing kernels inside bench.py)."""

    def __init__(self):
        p = C.c_void_p()
        check(lib().kmd_event_create(C.byref(p)), "event_create")
        self.ptr = p.value

    def record(self, stream=None):
        check(lib().kmd_event_record(self.ptr, stream), "event_record")

    def elapsed_ms(self, stop):
        ms = C.c_float(0)
        check(lib().kmd_event_elapsed_ms(self.ptr, stop.ptr, C.byref(ms)), "event_elapsed")
        return ms.value

    def __del__(self):
        try:
            if self.ptr:
                lib().kmd_event_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass


class CountMatrix:
    """One partition tile of the merged count matrix, resident in HBM.

    layout LAYOUT_SOA: counts[sample][row] (device-native); LAYOUT_ROWS: counts[row][sample]
    (what km::MatrixReader / KmerMerger emit, merge.hpp:194-203)."""

    def __init__(self, n_rows, n_samples, count_bytes=4, layout=N.LAYOUT_SOA, ld=None,
                 with_kmers=True, kmer_limbs=1, row_base=0):
        self.n_rows, self.n_samples = int(n_rows), int(n_samples)
        self.count_bytes, self.layout = int(count_bytes), int(layout)
        if ld is None:
            if layout == N.LAYOUT_SOA:
                # column pitch: whole 256-byte lines, so that every column starts on an L2
                # line and no wave access straddles one (16 bytes is the kernel's minimum
                # for its vector loads; KMD_LD_ALIGN overrides for experiments)
                import os
                per = max(int(os.environ.get("KMD_LD_ALIGN", "256")), 16) // self.count_bytes
                ld = (self.n_rows + per - 1) // per * per
                ld += int(os.environ.get("KMD_LD_PAD", "0")) // self.count_bytes
            elif layout == N.LAYOUT_TILED:
                # 1-byte counts: 8192-row blocks give every lane an 8-byte load (4.95 -> 5.7 TB/s)
                ld = TILED_BLOCK_ROWS * (2 if self.count_bytes == 1 else 1)
            else:
                ld = self.n_samples
        self.ld = int(ld)
        if layout == N.LAYOUT_SOA:
            n_el = self.ld * self.n_samples
        elif layout == N.LAYOUT_TILED:
            n_el = (self.n_rows + self.ld - 1) // self.ld * self.ld * self.n_samples
        else:
            n_el = self.ld * self.n_rows
        self.counts = DeviceBuffer(max(n_el, 1) * self.count_bytes)
        self.kmer_lo = DeviceBuffer(max(self.n_rows, 1) * 8) if with_kmers else None
        self.kmer_hi = DeviceBuffer(max(self.n_rows, 1) * 8) if (with_kmers and kmer_limbs == 2) else None
        self.row_base = int(row_base)

    @classmethod
    def from_host(cls, counts, layout=N.LAYOUT_ROWS, kmer_lo=None, kmer_hi=None, row_base=0):
        """counts: 2-D numpy array; [row][sample] for LAYOUT_ROWS, [sample][row] for SOA."""
        counts = np.ascontiguousarray(counts)
        cb = counts.dtype.itemsize
        if layout == N.LAYOUT_ROWS:
            n_rows, n_samples = counts.shape
            m = cls(n_rows, n_samples, cb, layout, ld=n_samples, with_kmers=kmer_lo is not None,
                    kmer_limbs=2 if kmer_hi is not None else 1, row_base=row_base)
            if counts.nbytes:
                check(lib().kmd_memcpy_h2d(m.counts.ptr, counts.ctypes.data, counts.nbytes, None), "h2d")
        else:
            n_samples, n_rows = counts.shape
            m = cls(n_rows, n_samples, cb, layout, with_kmers=kmer_lo is not None,
                    kmer_limbs=2 if kmer_hi is not None else 1, row_base=row_base)
            if layout == N.LAYOUT_SOA:
                padded = np.zeros((n_samples, m.ld), dtype=counts.dtype)
                padded[:, :n_rows] = counts
            else:   # tiled: [block][sample][row in block]
                nb = (n_rows + m.ld - 1) // m.ld
                padded = np.zeros((n_samples, nb * m.ld), dtype=counts.dtype)
                padded[:, :n_rows] = counts
                padded = np.ascontiguousarray(padded.reshape(n_samples, nb, m.ld).transpose(1, 0, 2))
            if padded.nbytes:
                check(lib().kmd_memcpy_h2d(m.counts.ptr, padded.ctypes.data, padded.nbytes, None), "h2d")
        if kmer_lo is not None and n_rows:
            a = np.ascontiguousarray(kmer_lo, dtype=np.uint64)
            check(lib().kmd_memcpy_h2d(m.kmer_lo.ptr, a.ctypes.data, a.nbytes, None), "h2d")
        if kmer_hi is not None and n_rows:
            a = np.ascontiguousarray(kmer_hi, dtype=np.uint64)
            check(lib().kmd_memcpy_h2d(m.kmer_hi.ptr, a.ctypes.data, a.nbytes, None), "h2d")
        return m

    def tile(self):
        return N.Tile(self.counts.ptr, self.count_bytes, self.layout, self.ld,
                      self.kmer_lo.ptr if self.kmer_lo else None,
                      self.kmer_hi.ptr if self.kmer_hi else None, self.n_rows, self.row_base)

    def to_host(self):
        """counts back as a [row][sample] numpy array (test helper)."""
        dt = _COUNT_DTYPES[self.count_bytes]
        if self.layout == N.LAYOUT_ROWS:
            a = self.counts.to_host(dt, self.n_rows * self.ld).reshape(self.n_rows, self.ld)
            return a[:, :self.n_samples].copy()
        if self.layout == N.LAYOUT_TILED:
            nb = (self.n_rows + self.ld - 1) // self.ld
            a = self.counts.to_host(dt, nb * self.n_samples * self.ld).reshape(nb, self.n_samples, self.ld)
            a = a.transpose(1, 0, 2).reshape(self.n_samples, nb * self.ld)
            return a[:, :self.n_rows].T.copy()
        a = self.counts.to_host(dt, self.n_samples * self.ld).reshape(self.n_samples, self.ld)
        return a[:, :self.n_rows].T.copy()

    def kmers_to_host(self):
        lo = self.kmer_lo.to_host(np.uint64, self.n_rows) if self.kmer_lo else None
        hi = self.kmer_hi.to_host(np.uint64, self.n_rows) if self.kmer_hi else None
        return lo, hi


class PoissonLikelihood:
    """include/kmdiff/model.hpp:94-192 -- constructor arguments as in the reference."""

    def __init__(self, nb_controls, nb_cases, total_controls, total_cases, preload=10000):
        _require_device()
        tc = np.ascontiguousarray(total_controls, dtype=np.uint64)
        tk = np.ascontiguousarray(total_cases, dtype=np.uint64)
        if len(tc) != nb_controls or len(tk) != nb_cases:
            raise ValueError("total_controls/total_cases must have nb_controls/nb_cases entries")
        self.nb_controls, self.nb_cases, self.preload = int(nb_controls), int(nb_cases), int(preload)
        h = C.c_void_p()
        check(lib().kmd_model_create(C.byref(h), self.nb_controls, self.nb_cases, tc.ctypes.data,
                                     tk.ctypes.data, self.preload), "kmd_model_create")
        self.handle = h.value
        self.sum_controls, self.sum_cases = int(tc.sum(dtype=np.uint64)), int(tk.sum(dtype=np.uint64))

    def lf_table(self):
        out = np.empty(self.preload, dtype=np.float64)
        if self.preload:
            check(lib().kmd_model_lf_table(self.handle, out.ctypes.data, self.preload), "lf_table")
        return out

    def process(self, matrix):
        """IModel::process (imodel.hpp:36) for every row of `matrix`:
        returns (p_value, sign, mean_control, mean_case) arrays."""
        n = matrix.n_rows
        bp, bs, bc, bk = (DeviceBuffer(max(n, 1) * 8), DeviceBuffer(max(n, 1) * 4),
                          DeviceBuffer(max(n, 1) * 8), DeviceBuffer(max(n, 1) * 8))
        t = matrix.tile()
        check(lib().kmd_poisson_process(self.handle, C.byref(t), bp.ptr, bs.ptr, bc.ptr, bk.ptr, None),
              "kmd_poisson_process")
        check(lib().kmd_stream_sync(None), "sync")
        return (bp.to_host(np.float64, n), bs.to_host(np.int32, n), bc.to_host(np.float64, n),
                bk.to_host(np.float64, n))

    def close(self):
        if getattr(self, "handle", None):
            lib().kmd_model_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SurvivorAccumulator:
    """IAccumulator<KmerSign<KSIZE>> (accumulator.hpp:36-54) as device SoA arrays."""

    FIELDS = (("row", np.uint64), ("kmer_lo", np.uint64), ("kmer_hi", np.uint64),
              ("pvalue", np.float64), ("sign", np.int32), ("mean_control", np.float64),
              ("mean_case", np.float64))

    def __init__(self, capacity, kmer_limbs=1):
        self.capacity = int(capacity)
        cap = max(self.capacity, 1)
        self.bufs = {}
        for name, dt in self.FIELDS:
            if name == "kmer_hi" and kmer_limbs < 2:
                self.bufs[name] = None
                continue
            self.bufs[name] = DeviceBuffer(cap * np.dtype(dt).itemsize)
        self.counters = DeviceBuffer(2 * N.NCOUNTERS * 8).zero()   # [NCOUNTERS..] dev-only timing slots
        self._size = None

    def struct(self):
        g = lambda k: self.bufs[k].ptr if self.bufs[k] else None
        return N.Survivors(g("row"), g("kmer_lo"), g("kmer_hi"), g("pvalue"), g("sign"),
                           g("mean_control"), g("mean_case"), self.capacity)

    def read_counters(self):
        check(lib().kmd_stream_sync(None), "sync")
        return self.counters.to_host(np.uint64, N.NCOUNTERS)

    def finish(self, sort=True, by_kmer=False, refine=None, allow_unresolved=False):
        """IAccumulator::finish: returns the number of survivors stored (sorted by row -- or by k-mer,
        for survivors of merge_filter, which have no row index -- the reference's push order).
        refine = the PoissonLikelihood the survivors were tested with: their p-values are recomputed with correctly
        rounded log / exp (kmd_pvalues_refine: the bits a glibc-built reference prints).
        Raises KmdError(KMD_E_OVERFLOW) if records were dropped, and KmdError if the guard of `p <= threshold` could not list
        every row within 1e-8 of the threshold (more than 4096 in one launch, KMD_CNT_NEAR_UNRESOLVED): zero the counters and
        run the matrix again with diff_observer.process_in_pieces -- or pass allow_unresolved=True to keep the device
        libm's decision for the rows beyond the list (a caller inside a multi-rank job should do one or the other BEFORE
        the exchange: a rank that raises here leaves its peers to the collective's timeout)."""
        c = self.read_counters()
        n = int(c[N.CNT_SIG])
        if n > self.capacity:
            raise KmdError("survivor capacity exceeded: %d > %d (status %d)" % (n, self.capacity, N.KMD_E_OVERFLOW))
        if len(c) > N.CNT_NEAR_UNRESOLVED and int(c[N.CNT_NEAR_UNRESOLVED]) and not allow_unresolved:
            # (include/kmdiff_hip.h, KMD_CNT_NEAR_UNRESOLVED: the guard of `p <= threshold` could not list them all)
            raise KmdError("%d rows within 1e-8 of the threshold were left with the device libm's decision (more than 4096 "
                           "in one launch): run the partition again in smaller pieces" % int(c[N.CNT_NEAR_UNRESOLVED]))
        if refine is not None and n:
            check(lib().kmd_pvalues_refine(refine.handle, n, self.bufs["mean_control"].ptr, self.bufs["mean_case"].ptr,
                                           self.bufs["pvalue"].ptr, None), "pvalues_refine")
            check(lib().kmd_stream_sync(None), "sync")
        if sort and n > 1:
            s = self.struct()
            if by_kmer:
                check(lib().kmd_survivors_sort_by_kmer(C.byref(s), n, None), "sort_by_kmer")
            else:
                check(lib().kmd_survivors_sort_by_row(C.byref(s), n, None), "sort_by_row")
        self._size = n
        return n

    def size(self):
        if self._size is None:
            self.finish()
        return self._size

    def get(self):
        """All stored survivors as a dict of numpy arrays."""
        n = self.size()
        out = {}
        for name, dt in self.FIELDS:
            out[name] = self.bufs[name].to_host(dt, n) if self.bufs[name] else None
        return out


class diff_observer:
    """include/kmdiff/merge.hpp:44-131, applied to a whole tile per call instead of a row."""

    def __init__(self, model, acc, threshold, nb_controls=None, nb_cases=None, partition=0):
        self.model, self.acc, self.threshold, self.partition = model, acc, float(threshold), partition
        if nb_controls is not None and nb_controls != model.nb_controls:
            raise ValueError("nb_controls differs from the model's")
        if nb_cases is not None and nb_cases != model.nb_cases:
            raise ValueError("nb_cases differs from the model's")

    def process(self, matrix, stream=None):
        if matrix.n_samples != self.model.nb_controls + self.model.nb_cases:
            raise ValueError("matrix has %d samples, model expects %d" %
                             (matrix.n_samples, self.model.nb_controls + self.model.nb_cases))
        t = matrix.tile()
        s = self.acc.struct()
        check(lib().kmd_poisson_filter(self.model.handle, C.byref(t), self.threshold, C.byref(s),
                                       self.acc.counters.ptr, stream), "kmd_poisson_filter")
        self.acc._size = None

    def process_in_pieces(self, matrix, rows_per_piece=4096, stream=None):
        """process() over row windows of `rows_per_piece` rows (rounded up to whole blocks of the tiled layout): the way
        to run a matrix again when finish() reports rows within 1e-8 of the threshold beyond the 4096 one launch can
        list (KMD_CNT_NEAR_UNRESOLVED) -- a piece of 4096 rows cannot overflow the list, so every such row gets the
        correctly rounded second look.  What `kmdiff-hip diff` does on that counter (kmdiff_amd/host/main.cpp).  The
        caller zeroes the accumulator's counters (and forgets its records: acc.counters.zero()) before the rerun."""
        if matrix.n_samples != self.model.nb_controls + self.model.nb_cases:
            raise ValueError("matrix has %d samples, model expects %d" % (matrix.n_samples, self.model.nb_controls + self.model.nb_cases))
        unit = matrix.ld if matrix.layout == N.LAYOUT_TILED else 1
        per = max(unit, (int(rows_per_piece) + unit - 1) // unit * unit)
        full = matrix.tile()
        s = self.acc.struct()
        for r0 in range(0, matrix.n_rows, per):
            if matrix.layout == N.LAYOUT_TILED:
                off = (r0 // matrix.ld) * matrix.n_samples * matrix.ld * matrix.count_bytes
            elif matrix.layout == N.LAYOUT_ROWS:
                off = r0 * matrix.ld * matrix.count_bytes
            else:
                off = r0 * matrix.count_bytes
            t = N.Tile(full.d_counts + off, matrix.count_bytes, matrix.layout, matrix.ld,
                       (full.d_kmer_lo + 8 * r0) if full.d_kmer_lo else None, (full.d_kmer_hi + 8 * r0) if full.d_kmer_hi else None,
                       min(per, matrix.n_rows - r0), matrix.row_base + r0)
            check(lib().kmd_poisson_filter(self.model.handle, C.byref(t), self.threshold, C.byref(s), self.acc.counters.ptr, stream),
                  "kmd_poisson_filter")
        self.acc._size = None

    def process_sums(self, sums, stream=None):
        """The rows as merge_sums leaves them: (k-mer, control sum, case sum).  Survivor `row` = index
        into those arrays."""
        s = self.acc.struct()
        check(lib().kmd_poisson_filter_sums(self.model.handle, sums.kmers.ptr, sums.sum_c.ptr, sums.sum_k.ptr, sums.n_rows,
                                            self.threshold, C.byref(s), self.acc.counters.ptr, stream), "kmd_poisson_filter_sums")
        self.acc._size = None

    def _c(self, i):
        return int(self.acc.read_counters()[i])

    def total(self):
        return self._c(N.CNT_TOTAL)

    def nb_sign(self):
        return self._c(N.CNT_SIG)

    def nb_signs(self):
        c = self.acc.read_counters()
        return int(c[N.CNT_SIG_CONTROL]), int(c[N.CNT_SIG_CASE])


def merge_partition(streams, n_samples=None, count_bytes=4, layout=N.LAYOUT_TILED, row_capacity=None):
    """km::KmerMerger over one partition (merge.hpp:265-289): `streams` is a list, one entry per
    sample in fof order, of (kmers uint64[], counts uint32[]) -- or (kmers_lo, counts, kmers_hi)
    for 32 < k <= 64 -- sorted by k-mer.  Returns a CountMatrix (with its k-mer column(s))
    resident on the device."""
    n_samples = len(streams) if n_samples is None else n_samples
    two = any(len(t) == 3 for t in streams)
    offs = np.zeros(n_samples + 1, dtype=np.uint64)
    for s, t in enumerate(streams):
        offs[s + 1] = offs[s] + len(t[0])
    total = int(offs[-1])
    cat = lambda i, dt: (np.concatenate([np.asarray(t[i], dtype=dt) for t in streams]) if total else np.zeros(0, dt))
    kmers, counts = cat(0, np.uint64), cat(1, np.uint32)
    cap = total if row_capacity is None else int(row_capacity)
    m = CountMatrix(max(cap, 1), n_samples, count_bytes, layout, with_kmers=True, kmer_limbs=2 if two else 1)
    dk, dc = DeviceBuffer.from_host(kmers), DeviceBuffer.from_host(counts)
    dh = DeviceBuffer.from_host(cat(2, np.uint64)) if two else None
    n_rows = C.c_uint64(0)
    check(lib().kmd_merge_partition(n_samples, dk.ptr if total else None, dh.ptr if (two and total) else None,
                                    dc.ptr if total else None, offs.ctypes.data, count_bytes, layout, m.ld, cap,
                                    m.counts.ptr, m.kmer_lo.ptr, m.kmer_hi.ptr if two else None,
                                    C.byref(n_rows), None), "kmd_merge_partition")
    m.n_rows = int(n_rows.value)
    return m


class StreamSet:
    """One partition's per-sample streams resident on the device, concatenated in sample order:
    what km::KmerMerger opens (merge.hpp:265-266).  `streams` as in merge_partition."""

    def __init__(self, streams):
        self.n_samples = len(streams)
        self.two = any(len(t) == 3 for t in streams)
        self.offs = np.zeros(self.n_samples + 1, dtype=np.uint64)
        for s, t in enumerate(streams):
            self.offs[s + 1] = self.offs[s] + len(t[0])
        self.total = int(self.offs[-1])
        cat = lambda i, dt: (np.concatenate([np.asarray(t[i], dtype=dt) for t in streams]) if self.total else np.zeros(0, dt))
        self.kmers, self.counts = DeviceBuffer.from_host(cat(0, np.uint64)), DeviceBuffer.from_host(cat(1, np.uint32))
        self.kmers_hi = DeviceBuffer.from_host(cat(2, np.uint64)) if self.two else None

    def ptrs(self):
        t = self.total
        return (self.kmers.ptr if t else None, self.kmers_hi.ptr if (self.two and t) else None, self.counts.ptr if t else None)


def synth_streams(seed, partition, n_rows, nb_controls, nb_cases, kmer_limbs=1, row0=0, profile=0):
    """The synthetic partition as the per-sample streams kmtricks would write, generated on the
    device (kmd_synth_streams): returns (StreamSet, per-sample totals).  `profile`: as synth_matrix."""
    S = nb_controls + nb_cases
    partition = int(partition) | (int(profile) << 8)
    ss = StreamSet.__new__(StreamSet)
    ss.n_samples, ss.two = S, kmer_limbs == 2
    ss.offs = np.zeros(S + 1, dtype=np.uint64)
    tot = DeviceBuffer(S * 8).zero()
    check(lib().kmd_synth_streams(int(seed), int(partition), int(row0), int(n_rows), nb_controls, nb_cases, ss.offs.ctypes.data,
                                  None, None, None, tot.ptr, None), "kmd_synth_streams (offsets)")
    ss.total = int(ss.offs[-1])
    ss.kmers, ss.counts = DeviceBuffer(max(ss.total, 1) * 8), DeviceBuffer(max(ss.total, 1) * 4)
    ss.kmers_hi = DeviceBuffer(max(ss.total, 1) * 8) if ss.two else None
    check(lib().kmd_synth_streams(int(seed), int(partition), int(row0), int(n_rows), nb_controls, nb_cases, ss.offs.ctypes.data,
                                  ss.kmers.ptr, ss.kmers_hi.ptr if ss.two else None, ss.counts.ptr, None, None), "kmd_synth_streams (fill)")
    return ss, tot.to_host(np.uint64, S)


PACK_BLOCK = 256


def pack_streams(streams):
    """The compact transfer format (kmd_pack_block, kmd_pack.hip) of a list of one-limb streams [(kmers, counts)]:
    returns (packed bytes uint8[], stream_base uint64[S + 1], block_off8 uint32[], offsets uint64[S + 1]).  Test / tool
    helper: the blocks are packed one by one through the C-ABI (the CLI packs while it decodes the files)."""
    L = lib()
    bound = int(L.kmd_pack_block_bound())
    S = len(streams)
    offs = np.zeros(S + 1, dtype=np.uint64)
    chunks, base, tables, at = [], np.zeros(S + 1, dtype=np.uint64), [], 0
    buf = np.zeros(bound, dtype=np.uint8)
    for s, (km, ct) in enumerate(streams):
        km = np.ascontiguousarray(km, dtype=np.uint64)
        ct = np.ascontiguousarray(ct, dtype=np.uint32)
        offs[s + 1] = offs[s] + len(km)
        base[s] = at
        within = 0
        for b in range(0, len(km), PACK_BLOCK):
            n = min(PACK_BLOCK, len(km) - b)
            got = int(L.kmd_pack_block(km[b:].ctypes.data, ct[b:].ctypes.data, n, buf.ctypes.data))
            assert got > 0 and got % 8 == 0 and got <= bound
            tables.append(within // 8)
            chunks.append(buf[:got].copy())
            within += got
        at += within
        base[s + 1] = at
    packed = np.concatenate(chunks) if chunks else np.zeros(8, dtype=np.uint8)
    return packed, base, np.asarray(tables, dtype=np.uint32), offs


def unpack_streams(packed, stream_base, block_off8, offs, stream=None):
    """kmd_unpack_streams: the packed bytes go to the device and come back as a StreamSet (the arrays kmd_merge_filter reads)."""
    S = len(stream_base) - 1
    ss = StreamSet.__new__(StreamSet)
    ss.n_samples, ss.two, ss.offs = S, False, np.ascontiguousarray(offs, dtype=np.uint64)
    ss.total = int(ss.offs[-1])
    ss.kmers, ss.counts, ss.kmers_hi = DeviceBuffer(max(ss.total, 1) * 8), DeviceBuffer(max(ss.total, 1) * 4), None
    dp = DeviceBuffer.from_host(np.ascontiguousarray(packed, dtype=np.uint8))
    dt = DeviceBuffer.from_host(np.ascontiguousarray(block_off8, dtype=np.uint32)) if len(block_off8) else DeviceBuffer(8)
    sb = np.ascontiguousarray(stream_base, dtype=np.uint64)
    check(lib().kmd_unpack_streams(S, dp.ptr, sb.ctypes.data, dt.ptr, ss.offs.ctypes.data, ss.kmers.ptr, ss.counts.ptr, stream), "kmd_unpack_streams")
    check(lib().kmd_stream_sync(stream), "sync")
    return ss


class RowSums:
    """What kmd_merge_sums leaves on the device: n_rows rows (k-mer, control sum, case sum), compact, in no
    particular order."""

    def __init__(self, capacity, two=False):
        self.kmers = DeviceBuffer(max(capacity, 1) * 8)
        self.kmers_hi = DeviceBuffer(max(capacity, 1) * 8) if two else None
        self.sum_c = DeviceBuffer(max(capacity, 1) * 8)
        self.sum_k = DeviceBuffer(max(capacity, 1) * 8)
        self.capacity, self.n_rows = int(capacity), 0

    def to_host(self):
        """(k-mers, control sums, case sums, entry index) of the rows (+ high limbs last, two-limb k-mers)."""
        n = self.n_rows
        km, sc, sk = self.kmers.to_host(np.uint64, n), self.sum_c.to_host(np.uint64, n), self.sum_k.to_host(np.uint64, n)
        rows = np.arange(n)
        if self.kmers_hi is not None:
            return km, sc, sk, rows, self.kmers_hi.to_host(np.uint64, n)
        return km, sc, sk, rows


def merge_sums(streams, nb_controls, row_capacity=None):
    """The merge of one partition for a consumer that only needs every k-mer's two count sums
    (PoissonLikelihood::process reads nothing else of a row, model.hpp:144-145): no matrix."""
    ss = streams if isinstance(streams, StreamSet) else StreamSet(streams)
    out = RowSums(max(ss.total, 1) if row_capacity is None else int(row_capacity), ss.two)
    n_rows = C.c_uint64(0)
    dk, dh, dc = ss.ptrs()
    rc = lib().kmd_merge_sums(ss.n_samples, int(nb_controls), dk, dh, dc, ss.offs.ctypes.data, out.capacity, out.kmers.ptr,
                              out.kmers_hi.ptr if ss.two else None, out.sum_c.ptr, out.sum_k.ptr, C.byref(n_rows), None)
    out.n_rows_needed = int(n_rows.value)
    check(rc, "kmd_merge_sums")
    out.n_rows = int(n_rows.value)
    out.streams = ss                                         # kept for gather_counts_streams
    return out


def merge_filter(streams, observer, stream=None):
    """km::KmerMerger::merge(diff_observer) for one partition (merge.hpp:265-289, 68-103): streams in,
    survivors into the observer's accumulator; returns the number of distinct k-mers.  Survivor `row` =
    low limb of the k-mer (SurvivorAccumulator.finish(by_kmer=True) gives the reference's order).  `stream`: a
    handle from kmd_stream_create (partitions on different streams, from different host threads, overlap)."""
    ss = streams if isinstance(streams, StreamSet) else StreamSet(streams)
    n_rows = C.c_uint64(0)
    s = observer.acc.struct()
    dk, dh, dc = ss.ptrs()
    check(lib().kmd_merge_filter(observer.model.handle, ss.n_samples, dk, dh, dc, ss.offs.ctypes.data, observer.threshold,
                                 C.byref(s), observer.acc.counters.ptr, C.byref(n_rows), stream), "kmd_merge_filter")
    observer.acc._size = None
    return int(n_rows.value)


def merge_filter_batch(stream_sets, observers, stream=None):
    """km::KmerMerger::merge(diff_observer) for a batch of partitions (global_merge's loop over the partitions,
    merge.hpp:259-307): kmd_merge_filter_batch keeps two or three of them in flight on streams of the library's own.
    `stream_sets`: one StreamSet per partition; `observers`: one diff_observer per partition (several may share an
    accumulator: survivors and counters add up).  Returns the partitions' numbers of distinct k-mers."""
    P = len(stream_sets)
    assert len(observers) == P
    if P == 0:
        return []
    model = observers[0].model
    S = stream_sets[0].n_samples
    assert all(o.model is model and o.threshold == observers[0].threshold for o in observers) and all(ss.n_samples == S for ss in stream_sets)
    vp = C.c_void_p * P
    ptrs = [ss.ptrs() for ss in stream_sets]
    any_hi = any(pt[1] is not None for pt in ptrs)
    dk = vp(*[pt[0] for pt in ptrs]); dh = vp(*[pt[1] for pt in ptrs]); dc = vp(*[pt[2] for pt in ptrs])
    offs = vp(*[ss.offs.ctypes.data for ss in stream_sets])
    ctr = vp(*[o.acc.counters.ptr for o in observers])
    sinks = (N.Survivors * P)(*[o.acc.struct() for o in observers])
    n_rows = (C.c_uint64 * P)()
    check(lib().kmd_merge_filter_batch(model.handle, P, S, dk, dh if any_hi else None, dc, offs, observers[0].threshold, sinks, ctr,
                                       n_rows, stream), "kmd_merge_filter_batch")
    for o in observers:
        o.acc._size = None
    return [int(v) for v in n_rows]


def gather_counts_streams(sums, rows_buf, n, row_kmers=None, row_kmers_hi=None):
    """KmerSign::m_counts_ratio for n survivors of the fused paths: host array [n][S] of doubles, every
    count looked up in the per-sample streams.  `sums`: a RowSums (rows_buf = the survivors' `row` on the
    device) or a StreamSet with row_kmers(/_hi) = the survivors' own k-mer columns (rows_buf None)."""
    ss = sums if isinstance(sums, StreamSet) else sums.streams
    if row_kmers is None:
        row_kmers, row_kmers_hi = sums.kmers, sums.kmers_hi
    out = DeviceBuffer(max(n, 1) * ss.n_samples * 8)
    dk, dh, dc = ss.ptrs()
    check(lib().kmd_survivors_gather_counts_streams(ss.n_samples, dk, dh, dc, ss.offs.ctypes.data, row_kmers.ptr,
                                                    row_kmers_hi.ptr if row_kmers_hi is not None else None,
                                                    rows_buf.ptr if rows_buf is not None else None, n, out.ptr, None),
          "kmd_survivors_gather_counts_streams")
    return out.to_host(np.float64, n * ss.n_samples).reshape(n, ss.n_samples)


class pop_strat_corrector:
    """include/kmdiff/popstrat.hpp:148-367 -- constructor arguments as in the reference plus the
    data its load_Z / load_Y read from files (Z: n x z_cols principal components, Y: 1.0 for
    controls / 0.0 for cases)."""

    def __init__(self, nb_controls, nb_cases, control_totals, case_totals, npc, Z, Y=None,
                 stand=True, max_iter=0, epsilon=0.0):
        _require_device()
        n = nb_controls + nb_cases
        tc = np.ascontiguousarray(control_totals, dtype=np.uint64)
        tk = np.ascontiguousarray(case_totals, dtype=np.uint64)
        Z = np.ascontiguousarray(Z, dtype=np.float64)
        if Z.shape[0] != n:
            raise ValueError("Z must have one row per sample")
        if Y is None:
            Y = np.concatenate([np.ones(nb_controls), np.zeros(nb_cases)])       # popstrat.cpp:168
        Y = np.ascontiguousarray(Y, dtype=np.float64)
        h = C.c_void_p()
        check(lib().kmd_popstrat_create(C.byref(h), nb_controls, nb_cases, tc.ctypes.data, tk.ctypes.data,
                                        Z.ctypes.data, Z.shape[1], int(npc), Y.ctypes.data, int(bool(stand)),
                                        int(max_iter)), "kmd_popstrat_create")
        self.handle, self.n = h.value, n
        if epsilon:                                             # set_params (popstrat.hpp:162-175): only a non-zero value counts
            check(lib().kmd_popstrat_set_epsilon(self.handle, float(epsilon)), "kmd_popstrat_set_epsilon")
        nf = C.c_int(0)
        check(lib().kmd_popstrat_info(self.handle, None, C.byref(nf), None, None, None), "popstrat_info")
        self.n_features = nf.value

    def info(self):
        alt = np.zeros((self.n, self.n_features))
        nm = np.zeros(self.n_features - 1)
        nl = C.c_double(0)
        check(lib().kmd_popstrat_info(self.handle, None, None, alt.ctypes.data, nm.ctypes.data, C.byref(nl)),
              "popstrat_info")
        return alt, nm, nl.value

    def apply(self, counts_buf, n, sample_major=False, ld=0):
        """apply(KmerSign&) over n survivors; counts_buf holds their count vectors as doubles
        (kmd_survivors_gather_counts order).  Returns the corrected p-values (numpy)."""
        out = DeviceBuffer(max(n, 1) * 8)
        check(lib().kmd_popstrat_apply(self.handle, counts_buf.ptr if n else None, int(sample_major), int(ld),
                                       int(n), out.ptr, None), "kmd_popstrat_apply")
        check(lib().kmd_stream_sync(None), "sync")
        self.last_pvalues = out
        return out.to_host(np.float64, n)

    def close(self):
        if getattr(self, "handle", None):
            lib().kmd_popstrat_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gather_counts(matrix, rows_buf, n):
    """KmerSign::m_counts_ratio of n survivors (merge.hpp:91-92): [n][S] doubles on the device."""
    out = DeviceBuffer(max(n * matrix.n_samples, 1) * 8)
    t = matrix.tile()
    check(lib().kmd_survivors_gather_counts(C.byref(t), matrix.n_samples, rows_buf.ptr, n, out.ptr, None),
          "gather_counts")
    return out


def aggregate(correction, threshold, total_kmers, pvalue_buf, sign_buf, n):
    """make_corrector + make_aggregator()->run() decisions (corrector.cpp:101-116,
    aggregator.hpp:343-365) over n device-resident survivors.
    Returns (keep mask as numpy uint8, n_control, n_case)."""
    if isinstance(correction, str):
        correction = CORRECTION_BY_NAME[correction.lower()]
    keep = DeviceBuffer(max(n, 1))
    nk, nc, nca = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    check(lib().kmd_correct(int(correction), float(threshold), int(total_kmers),
                            pvalue_buf.ptr if n else None, sign_buf.ptr if (n and sign_buf) else None,
                            n, keep.ptr, C.byref(nk), C.byref(nc), C.byref(nca), None), "kmd_correct")
    return keep.to_host(np.uint8, n), int(nc.value), int(nca.value)


SYNTH_MIXED = 1          # KMD_SYNTH_MIXED: every second row in one or two samples, the others in 95 % of them


def synth_matrix(seed, partition, n_rows, nb_controls, nb_cases, count_bytes=4,
                 layout=N.LAYOUT_SOA, kmer_limbs=1, row0=0, with_kmers=True, stream=None, profile=0):
    """Synthetic partition generated on the device; `profile`: the rows' presence profile
    (include/kmdiff_hip.h, kmd_synth_fill)."""
    m = CountMatrix(n_rows, nb_controls + nb_cases, count_bytes, layout, with_kmers=with_kmers,
                    kmer_limbs=kmer_limbs, row_base=row0)
    check(lib().kmd_synth_fill(int(seed), int(partition) | (int(profile) << 8), int(row0), m.n_rows, nb_controls, nb_cases,
                               count_bytes, layout, m.ld, m.counts.ptr,
                               m.kmer_lo.ptr if m.kmer_lo else None,
                               m.kmer_hi.ptr if m.kmer_hi else None, stream), "kmd_synth_fill")
    return m


def column_sums(matrix, totals_buf=None, stream=None):
    """Per-sample totals of a device matrix (the role of get_total_kmer for synthetic data)."""
    own = totals_buf is None
    if own:
        totals_buf = DeviceBuffer(matrix.n_samples * 8).zero()
    check(lib().kmd_column_sums(matrix.counts.ptr, matrix.count_bytes, matrix.layout, matrix.ld,
                                matrix.n_rows, matrix.n_samples, totals_buf.ptr, stream), "column_sums")
    if own:
        check(lib().kmd_stream_sync(None), "sync")
        return totals_buf.to_host(np.uint64, matrix.n_samples)
    return None


class PopulationPCA:
    """Sampler + smartpca (popstrat.hpp:55-146, src/popstrat.cpp:97-134) on the device:
    sample rows of the partitions as they pass, then gram() and pca_eigen()."""

    def __init__(self, n_samples, rate, seed=0, diploid=True, capacity=1 << 20):
        self.n_samples = int(n_samples)
        h = C.c_void_p()
        check(lib().kmd_pca_create(C.byref(h), self.n_samples, float(rate), int(seed), 1 if diploid else 0, int(capacity)),
              "kmd_pca_create")
        self.handle = h

    def sample(self, mat):
        if mat.n_samples != self.n_samples:
            raise ValueError("matrix has %d samples, the PCA %d" % (mat.n_samples, self.n_samples))
        t = mat.tile()
        check(lib().kmd_pca_sample(self.handle, C.byref(t), None), "kmd_pca_sample")

    def sample_streams(self, streams):
        """The same rows for the fused merge (no matrix): sampled k-mers are looked up in the per-sample
        streams (a StreamSet, or a list of streams as merge_partition takes)."""
        ss = streams if isinstance(streams, StreamSet) else StreamSet(streams)
        if ss.n_samples != self.n_samples:
            raise ValueError("%d streams, the PCA has %d samples" % (ss.n_samples, self.n_samples))
        dk, dh, dc = ss.ptrs()
        check(lib().kmd_pca_sample_streams(self.handle, ss.n_samples, dk, dh, dc, ss.offs.ctypes.data, None), "kmd_pca_sample_streams")

    def count(self):
        n = C.c_uint64(0)
        check(lib().kmd_pca_count(self.handle, C.byref(n)), "kmd_pca_count")
        return int(n.value)

    def gram(self):
        xtx = np.zeros((self.n_samples, self.n_samples), dtype=np.float64)
        check(lib().kmd_pca_gram(self.handle, xtx.ctypes.data, None), "kmd_pca_gram")
        return xtx

    def close(self):
        if self.handle:
            lib().kmd_pca_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pca_eigen(xtx, n_out=10):
    """(evec [S][n_out] unit-norm columns, eigenvalues[n_out] decreasing) of the summed Gram matrix."""
    xtx = np.ascontiguousarray(xtx, dtype=np.float64)
    S = xtx.shape[0]
    n_out = min(int(n_out), S)
    evec = np.zeros((S, n_out), dtype=np.float64)
    evals = np.zeros(n_out, dtype=np.float64)
    check(lib().kmd_pca_eigen(S, xtx.ctypes.data, n_out, evec.ctypes.data, evals.ctypes.data), "kmd_pca_eigen")
    return evec, evals

