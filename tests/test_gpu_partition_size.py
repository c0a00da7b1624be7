"""The fused merge + test (kmd_merge_filter, kmd_merge_filter_batch: the default product path) on partitions of the
size the reference hands a task: km::KmerMerger merges a WHOLE partition per task (include/kmdiff/merge.hpp:265-289);
a partition of BASELINE.json configs[2] is 39 062 500 rows, ~10^9 records, 12 GB of streams.  The streams are built on
the device (kmd_synth_streams: the same synthetic partition kmd_synth_fill writes as a matrix), so the checks are
  * bit for bit against K1 (kmd_poisson_filter on the matrix of the same partition: oracle-checked in test_gpu_parity),
  * exact oracle replay of sampled row windows (test_gpu_parity.replay_windows),
  * the 32-bit limits of the kernel (2^32 - 129 records, 2^29 per sample): a run of >= 2^28 records of one sample passes,
    anything beyond the limits is refused with KMD_E_INVALID -- rows are never lost silently.
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as OL
from test_gpu_parity import replay_windows, SEED, THR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    import kmdiff_amd as K
    assert K.device_count() >= 1, "no GPU: the HIP path has no CPU fallback"
    return K


def row_of_kmer(got, limbs):
    """kmd_synth_fill: k-mer = part * 2^54 + row * 2^21 + 1 + 20 random bits (the high limb, with two)"""
    k = got["kmer_hi"] if limbs == 2 else got["kmer_lo"]
    return (k >> np.uint64(21)) & np.uint64((1 << 33) - 1)


def same_survivors(a, b, limbs=1):
    assert a["kmer_lo"].tolist() == b["kmer_lo"].tolist()
    if limbs == 2:
        assert a["kmer_hi"].tolist() == b["kmer_hi"].tolist()
    for f in ("pvalue", "sign", "mean_control", "mean_case"):
        assert a[f].tolist() == b[f].tolist(), f


@pytest.mark.parametrize("nc,nk,limbs,rows,chunk", [(20, 20, 1, 150_001, 0), (20, 20, 1, 150_001, 8192), (3, 2, 2, 70_000, 4096),
                                                    (1, 1, 1, 1, 0), (5, 4, 1, 1023, 1024)])
def test_synth_streams_are_the_matrix_column_by_column(K, nc, nk, limbs, rows, chunk):
    """kmd_synth_streams == the non-zero cells of kmd_synth_fill's matrix, sample by sample, in row order -- also when the
    partition is built in many chunks."""
    S = nc + nk
    mat = K.synth_matrix(SEED, 5, rows, nc, nk, 4, K.LAYOUT_ROWS, kmer_limbs=limbs)
    host = mat.to_host()
    lo, hi = mat.kmers_to_host()
    if chunk:
        os.environ["KMD_SYNTH_CHUNK"] = str(chunk)
    try:
        ss, tot = K.synth_streams(SEED, 5, rows, nc, nk, kmer_limbs=limbs)
    finally:
        os.environ.pop("KMD_SYNTH_CHUNK", None)
    assert tot.tolist() == host.sum(axis=0, dtype=np.uint64).tolist()
    km, cn = ss.kmers.to_host(np.uint64, ss.total), ss.counts.to_host(np.uint32, ss.total)
    kh = ss.kmers_hi.to_host(np.uint64, ss.total) if limbs == 2 else None
    for s in range(S):
        sel = host[:, s] > 0
        a, b = int(ss.offs[s]), int(ss.offs[s + 1])
        assert b - a == int(sel.sum())
        assert np.array_equal(km[a:b], lo[sel]) and np.array_equal(cn[a:b], host[sel, s])
        if limbs == 2:
            assert np.array_equal(kh[a:b], hi[sel])


def run_both(K, part, rows, nc, nk, limbs, thr=THR, profile=0):
    """K1 on the matrix and K2t (single call, then a batch of two) on the streams of one synthetic partition"""
    ss, tot = K.synth_streams(SEED, part, rows, nc, nk, kmer_limbs=limbs, profile=profile)
    model = K.PoissonLikelihood(nc, nk, tot[:nc], tot[nc:], 10000)
    cap = max(rows // 200, 1 << 16)
    mat = K.synth_matrix(SEED, part, rows, nc, nk, 4, K.LAYOUT_TILED, kmer_limbs=limbs, profile=profile)
    assert K.column_sums(mat).tolist() == tot.tolist()
    a = K.SurvivorAccumulator(cap, kmer_limbs=limbs)
    K.diff_observer(model, a, thr).process(mat)
    na = a.finish()
    ga, ca = a.get(), a.read_counters()
    del mat
    b = K.SurvivorAccumulator(cap, kmer_limbs=limbs)
    n_rows = K.merge_filter(ss, K.diff_observer(model, b, thr))
    nb = b.finish(by_kmer=True)
    gb, cb = b.get(), b.read_counters()
    assert n_rows == rows
    assert [int(x) for x in cb[:4]] == [int(x) for x in ca[:4]] and int(cb[6]) == int(ca[6]) and int(cb[7]) == 0
    assert na == nb
    same_survivors(ga, gb, limbs)
    return ss, tot, model, gb, cb


def test_fused_merge_on_a_config3_partition(K, oracle):
    """configs[2]: one of the 256 partitions, 39 062 500 rows of 20v20, ~10^9 records (12 GB) -- through
    kmd_merge_filter and kmd_merge_filter_batch."""
    nc, nk, rows, part = 20, 20, 39_062_500, 17
    ss, tot, model, got, c = run_both(K, part, rows, nc, nk, 1)
    assert ss.total > 900_000_000 and int(c[0]) == rows and 1000 < int(c[1]) < rows // 200
    assert int(got["kmer_lo"].min()) >> 54 == part and int(got["kmer_lo"].max()) >> 54 == part
    rng = np.random.default_rng(17)
    got_rows = dict(got)
    got_rows["row"] = row_of_kmer(got, 1)
    assert (np.diff(got_rows["row"].astype(np.int64)) > 0).all()
    replay_windows(K, oracle, part, rows, nc, nk, 1, tot, got_rows, np.concatenate([[0, rows - 2048], rng.integers(0, rows - 2048, 10)]))
    # the batch entry point on two copies of the partition (the same streams twice: they are only read), sinks of their own
    accs = [K.SurvivorAccumulator(1 << 17), K.SurvivorAccumulator(1 << 17)]
    assert K.merge_filter_batch([ss, ss], [K.diff_observer(model, x, THR) for x in accs]) == [rows, rows]
    for x in accs:
        assert x.finish(by_kmer=True) == len(got["sign"])
        same_survivors(x.get(), got)
        assert [int(v) for v in x.read_counters()[:4]] == [int(v) for v in c[:4]]


def test_mixed_partition_rare_and_common_rows(K, oracle):
    """The MIXED presence profile (KMD_SYNTH_MIXED: bench.py's pipeline.sparse): every second row in one or two samples,
    the others in ~95 % of them.  Small: the device-built streams are the device-built matrix column by column, and the
    fused merge's survivors are the ORACLE's on that matrix (bit-exact identity, sign, means, counters; p within 1e-10).
    Whole configs[2] size (39 062 500 rows, ~7.7 x 10^8 records): K2t == K1 bit for bit, single call and batch."""
    import oracle_lib as OL
    from test_gpu_parity import assert_p_close
    nc, nk, rows = 20, 20, 120_000
    S = nc + nk
    mat = K.synth_matrix(SEED, 7, rows, nc, nk, 4, K.LAYOUT_ROWS, profile=K.SYNTH_MIXED)
    host, (lo, _) = mat.to_host(), mat.kmers_to_host()
    present = (host > 0).sum(axis=1)
    rare = present <= 2
    assert 0.45 < rare.mean() < 0.55 and present[rare].min() >= 1 and present[~rare].mean() > 0.93 * S and present[~rare].min() > 0.7 * S
    ss, tot = K.synth_streams(SEED, 7, rows, nc, nk, profile=K.SYNTH_MIXED)
    assert tot.tolist() == host.sum(axis=0, dtype=np.uint64).tolist() and ss.total == int(present.sum())
    km, cn = ss.kmers.to_host(np.uint64, ss.total), ss.counts.to_host(np.uint32, ss.total)
    for s in range(S):
        sel = host[:, s] > 0
        a, b = int(ss.offs[s]), int(ss.offs[s + 1])
        assert np.array_equal(km[a:b], lo[sel]) and np.array_equal(cn[a:b], host[sel, s])
    thr = 1e-4
    model = K.PoissonLikelihood(nc, nk, tot[:nc], tot[nc:], 10000)
    acc = K.SurvivorAccumulator(rows)
    assert K.merge_filter(ss, K.diff_observer(model, acc, thr)) == rows
    n = acc.finish(by_kmer=True)
    got, c = acc.get(), acc.read_counters()
    want = oracle.diff_partition(host, OL.LAYOUT_ROWS, nc, nk, int(tot[:nc].sum()), int(tot[nc:].sum()), oracle.lf_table(10000), thr)
    rr = want["row"].astype(np.int64)
    assert n == len(rr) > 20 and tuple(int(v) for v in c[:4]) == want["counters"]
    assert got["kmer_lo"].tolist() == lo[rr].tolist() and got["sign"].tolist() == want["sign"].tolist()
    assert got["mean_control"].tolist() == want["mean_control"].tolist() and got["mean_case"].tolist() == want["mean_case"].tolist()
    assert_p_close(got["pvalue"], want["pvalue"])
    del mat, ss, acc
    ss, tot, model, got, c = run_both(K, 7, 39_062_500, nc, nk, 1, profile=K.SYNTH_MIXED)
    assert 600_000_000 < ss.total < 900_000_000 and int(c[0]) == 39_062_500 and int(c[1]) > 100
    accs = [K.SurvivorAccumulator(1 << 17), K.SurvivorAccumulator(1 << 17)]
    assert K.merge_filter_batch([ss, ss], [K.diff_observer(model, x, THR) for x in accs]) == [39_062_500] * 2
    for x in accs:
        assert x.finish(by_kmer=True) == len(got["sign"])
        same_survivors(x.get(), got)


def test_fused_merge_on_a_config4_partition(K, oracle):
    """configs[3]'s shape: k = 63 (two limbs), 50v50, 16 M rows -- ~10^9 records of 20 bytes."""
    nc, nk, rows, part = 50, 50, 16_000_000, 3
    ss, tot, model, got, c = run_both(K, part, rows, nc, nk, 2)
    assert ss.total > 900_000_000 and int(c[5]) > 0                        # sums beyond the log-factorial table
    got_rows = dict(got)
    got_rows["row"] = row_of_kmer(got, 2)
    rng = np.random.default_rng(3)
    replay_windows(K, oracle, part, rows, nc, nk, 2, tot, got_rows, np.concatenate([[0, rows - 2048], rng.integers(0, rows - 2048, 6)]))


@pytest.mark.parametrize("nc,nk,rows,part", [(4, 4, 100_000_000, 0), (100, 100, 8_000_000, 1)])
def test_fused_merge_on_config2_and_config5_partitions(K, oracle, nc, nk, rows, part):
    """configs[1] (10^8 rows of 4v4: 5 x 10^8 records) and configs[4]'s shape (100v100, 8 M rows: 1.05 x 10^9 records)
    through the fused merge: K1's survivors bit for bit, oracle replay of row windows."""
    ss, tot, model, got, c = run_both(K, part, rows, nc, nk, 1)
    assert ss.total > 400_000_000 and int(c[0]) == rows and int(c[1]) > 500
    got_rows = dict(got)
    got_rows["row"] = row_of_kmer(got, 1)
    rng = np.random.default_rng(nc)
    replay_windows(K, oracle, part, rows, nc, nk, 1, tot, got_rows, np.concatenate([[0, rows - 2048], rng.integers(0, rows - 2048, 4)]))


def test_runs_of_more_than_2_to_the_28_records(K, oracle):
    """Two samples v two, each with more than 2^28 records in the partition (the kernel's positions and run extents are
    32-bit: the limit is 2^29 - 1 per sample): same survivors as K1 on the matrix, every row counted."""
    nc, nk, rows, part = 2, 2, 460_000_000, 9
    ss, tot, model, got, c = run_both(K, part, rows, nc, nk, 1, thr=1e-9)
    per = np.diff(ss.offs.astype(np.int64))
    assert per.min() > (1 << 28) and per.max() < (1 << 29), per
    got_rows = dict(got)
    got_rows["row"] = row_of_kmer(got, 1)
    rng = np.random.default_rng(9)
    replay_windows(K, oracle, part, rows, nc, nk, 1, tot, got_rows, np.concatenate([[0, rows - 2048], rng.integers(0, rows - 2048, 4)]), thr=1e-9)


def test_beyond_the_32_bit_limits_is_refused_not_truncated(K):
    """Offsets that claim 2^29 records of one sample, or 2^32 - 129 in all (nine samples, each inside its own limit):
    KMD_E_INVALID from every entry point that takes streams, before anything is read (the buffers here are a few
    bytes) -- and nothing reaches the counters."""
    lib = K._native.lib()
    model = K.PoissonLikelihood(5, 4, [10] * 5, [10] * 4, 100)
    acc = K.SurvivorAccumulator(16)
    buf = K.DeviceBuffer(64)
    s = acc.struct()
    n_rows = C.c_uint64(7)
    small = list(range(9))
    cases = [[0, 1 << 29] + [(1 << 29) + i for i in range(1, 9)],                    # one sample too long
             [i * ((1 << 29) - 1) for i in range(10)],                                # every sample inside its limit, too many in all
             [0, 5, 3] + [8 + i for i in range(7)]]                                   # not ascending
    assert cases[1][-1] >= (1 << 32) - 129 and small
    for offs in cases:
        o = np.array(offs, dtype=np.uint64)
        rc = lib.kmd_merge_filter(model.handle, 9, buf.ptr, None, buf.ptr, o.ctypes.data, 1e-3, C.byref(s), acc.counters.ptr, C.byref(n_rows), None)
        assert rc == -1, (offs, rc)
        rc = lib.kmd_merge_sums(9, 5, buf.ptr, None, buf.ptr, o.ctypes.data, 4, buf.ptr, None, buf.ptr, buf.ptr, C.byref(n_rows), None)
        assert rc == -1, (offs, rc)
        vp = C.c_void_p * 1
        rc = lib.kmd_merge_filter_batch(model.handle, 1, 9, vp(buf.ptr), None, vp(buf.ptr), vp(o.ctypes.data), 1e-3, C.byref(s), vp(acc.counters.ptr),
                                        C.byref(n_rows), None)
        assert rc == -1, (offs, rc)
    assert [int(x) for x in acc.read_counters()[:4]] == [0, 0, 0, 0]
