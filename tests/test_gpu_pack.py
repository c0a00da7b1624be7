"""kmd_unpack_streams (kmdiff_amd/csrc/kmd_pack.hip): the device decoder of the compact transfer format gives back the
(k-mer, count) arrays the host packed, bit for bit -- every delta width, partial and empty streams, escapes -- and
kmd_merge_filter on the unpacked streams finds the survivors it finds on the plain arrays."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    import kmdiff_amd as K
    assert K.device_count() >= 1, "no GPU: the HIP path has no CPU fallback"
    return K


def check_roundtrip(K, streams):
    packed, base, table, offs = K.pack_streams(streams)
    ss = K.unpack_streams(packed, base, table, offs)
    km, ct = ss.kmers.to_host(np.uint64, ss.total), ss.counts.to_host(np.uint32, ss.total)
    want_k = np.concatenate([np.asarray(s[0], dtype=np.uint64) for s in streams]) if ss.total else np.zeros(0, np.uint64)
    want_c = np.concatenate([np.asarray(s[1], dtype=np.uint32) for s in streams]) if ss.total else np.zeros(0, np.uint32)
    assert np.array_equal(km, want_k) and np.array_equal(ct, want_c)
    return len(packed) / max(ss.total, 1), ss


def test_every_width_and_ragged_streams(K):
    rng = np.random.default_rng(2)
    streams = []
    for bits in (0, 1, 9, 21, 22, 23, 31, 32, 33, 44, 55, 63, 64):
        n = int(rng.integers(1, 3000))
        if bits == 64:
            km = rng.integers(0, 2 ** 64, n, dtype=np.uint64)
        elif bits == 0:
            km = np.full(n, 2 ** 63 + 5, dtype=np.uint64)
        else:
            km = np.cumsum(rng.integers(0, 1 << min(bits, 50), n, dtype=np.uint64), dtype=np.uint64)
        ct = rng.integers(1, 600, n).astype(np.uint32)
        ct[rng.integers(0, n, 3)] = np.uint32(2 ** 32 - 1)
        streams.append((km, ct))
    streams.insert(3, (np.zeros(0, np.uint64), np.zeros(0, np.uint32)))            # an empty stream in the middle
    for n in (1, 255, 256, 257, 512):                                               # block boundaries
        streams.append((np.arange(n, dtype=np.uint64) * np.uint64(977) + np.uint64(5), np.full(n, 255, dtype=np.uint32)))
    check_roundtrip(K, streams)
    check_roundtrip(K, [(np.zeros(0, np.uint64), np.zeros(0, np.uint32))] * 3)      # nothing at all


def test_unpacked_partition_through_the_fused_merge(K, oracle):
    """A synthetic 20v20 partition: packed on the host, unpacked on the device, merged and tested -- the survivors of the
    plain arrays, and ~4 bytes per record across the link instead of 12."""
    import oracle_lib as OL
    seed, n, nc, nk, thr = 0x6B6D64696666, 60_000, 20, 20, 1e-4
    host, lo, _ = oracle.synth_rows(seed, 2, 0, n, nc, nk, 4)
    streams = [(lo[host[:, s] > 0], host[host[:, s] > 0, s]) for s in range(nc + nk)]
    per_record, ss = check_roundtrip(K, streams)
    assert 3.5 < per_record < 5.0, per_record
    tot = host.sum(axis=0, dtype=np.uint64)
    model = K.PoissonLikelihood(nc, nk, tot[:nc], tot[nc:], 10000)
    a, b = K.SurvivorAccumulator(n), K.SurvivorAccumulator(n)
    assert K.merge_filter(ss, K.diff_observer(model, a, thr)) == n
    assert K.merge_filter(K.StreamSet(streams), K.diff_observer(model, b, thr)) == n
    na, nb = a.finish(by_kmer=True), b.finish(by_kmer=True)
    ga, gb = a.get(), b.get()
    assert na == nb > 20
    for f in ("kmer_lo", "pvalue", "sign", "mean_control", "mean_case"):
        assert ga[f].tolist() == gb[f].tolist(), f
    want = oracle.diff_partition(host, OL.LAYOUT_ROWS, nc, nk, int(tot[:nc].sum()), int(tot[nc:].sum()), oracle.lf_table(10000), thr)
    assert ga["kmer_lo"].tolist() == lo[want["row"].astype(np.int64)].tolist()


def test_bad_arguments(K):
    lib = K._native.lib()
    buf = K.DeviceBuffer(64)
    base = np.array([4, 16], dtype=np.uint64)                                        # not a multiple of 8
    offs = np.array([0, 10], dtype=np.uint64)
    assert lib.kmd_unpack_streams(1, buf.ptr, base.ctypes.data, buf.ptr, offs.ctypes.data, buf.ptr, buf.ptr, None) == -1
    base[0] = 0
    offs[:] = [10, 3]
    assert lib.kmd_unpack_streams(1, buf.ptr, base.ctypes.data, buf.ptr, offs.ctypes.data, buf.ptr, buf.ptr, None) == -1
    offs[:] = [0, 10]
    assert lib.kmd_unpack_streams(1, None, base.ctypes.data, buf.ptr, offs.ctypes.data, buf.ptr, buf.ptr, None) == -1


def test_damaged_block_headers_stay_inside_the_block(K):
    """A delta width beyond 64 or an escape count beyond the block cannot come from kmd_pack_block; the device decoder
    sees that the header does not describe the block's bytes and writes zero records for it -- no access outside the
    block (the call returns, the block's neighbours and the other stream are untouched)."""
    rng = np.random.default_rng(4)
    a = (np.cumsum(rng.integers(1, 1 << 20, 600, dtype=np.uint64), dtype=np.uint64), rng.integers(1, 300, 600).astype(np.uint32))
    b = (np.cumsum(rng.integers(1, 1 << 30, 300, dtype=np.uint64), dtype=np.uint64), rng.integers(1, 9, 300).astype(np.uint32))
    packed, base, table, offs = K.pack_streams([a, b])
    bad = packed.copy()
    bad[8] = 200                                           # width byte of stream 0's first block
    bad[10] = 0xFF; bad[11] = 0xFF                         # ... and its escape count
    ss = K.unpack_streams(bad, base, table, offs)
    km, ct = ss.kmers.to_host(np.uint64, ss.total), ss.counts.to_host(np.uint32, ss.total)
    assert np.array_equal(km[600:], b[0]) and np.array_equal(ct[600:], b[1])          # the other stream as it was
    assert np.array_equal(km[256:600], a[0][256:]) and np.array_equal(ct[256:600], a[1][256:])   # and stream 0's other blocks
    assert not km[:256].any() and not ct[:256].any()                                               # the damaged one: zeros
    # one damaged byte at a time, in every header field and in the block table's view of the lengths
    for pos, val in ((8, 65), (8, 63), (10, 1), (11, 1)):
        bad = packed.copy()
        if bad[pos] == val:
            continue
        bad[pos] = val
        ss = K.unpack_streams(bad, base, table, offs)
        km = ss.kmers.to_host(np.uint64, ss.total)
        assert np.array_equal(km[256:], np.concatenate([a[0][256:], b[0]])) and not km[:256].any(), (pos, val)
