"""The compact transfer format of a stream (kmd_pack_block, kmdiff_amd/csrc/kmd_pack.hip) against an independent
decoder written from the format's description (numpy, no GPU): every block is its first k-mer, 256 bit-packed deltas of
the block's widest delta, one-byte counts with 255 as the escape into a list of 32-bit counts."""
import os

import numpy as np
import pytest

import kmdiff_amd as K


def decode_block(b, n):
    """(k-mers, counts, bytes of the block) of a block of n records at the start of the uint8 array b"""
    anchor = int(b[0:8].view("<u8")[0])
    w, n_esc = int(b[8]), int(b[10:12].view("<u2")[0])
    assert b[9] == 0 and not b[12:16].any()
    n_words = 4 * w + 1
    words = [int(x) for x in b[16:16 + 8 * n_words].view("<u8")]
    big = sum(v << (64 * i) for i, v in enumerate(words))                     # the bit string as one integer
    deltas = [(big >> (i * w)) & ((1 << w) - 1) if w else 0 for i in range(256)]
    assert deltas[0] == 0 and not any(deltas[n:])
    keys, k = [], anchor
    for i in range(n):
        k = (k + deltas[i]) & (2 ** 64 - 1)
        keys.append(k)
    cb = b[16 + 8 * n_words:16 + 8 * n_words + 256]
    esc = b[16 + 8 * n_words + 256:16 + 8 * n_words + 256 + 4 * n_esc].view("<u4")
    counts, e = [], 0
    for i in range(n):
        if cb[i] == 255:
            counts.append(int(esc[e])); e += 1
        else:
            counts.append(int(cb[i]))
    assert e == n_esc and not cb[n:].any()
    size = 16 + 8 * n_words + 256 + 4 * n_esc
    return keys, counts, (size + 7) // 8 * 8


def roundtrip(km, ct):
    packed, base, table, offs = K.pack_streams([(km, ct)])
    assert base.tolist() == [0, len(packed) if len(km) else 0] and offs.tolist() == [0, len(km)] and len(table) == (len(km) + 255) // 256
    keys, counts, at = [], [], 0
    for j, b in enumerate(range(0, len(km), 256)):
        assert int(table[j]) * 8 == at
        k, c, size = decode_block(packed[at:], min(256, len(km) - b))
        keys += k; counts += c; at += size
    assert at == len(packed) or len(km) == 0
    assert keys == [int(x) for x in km] and counts == [int(x) for x in ct]
    return len(packed)


@pytest.mark.parametrize("bits", [0, 1, 7, 22, 33, 45, 62, 63, 64])
@pytest.mark.parametrize("n", [1, 2, 255, 256, 257, 1000])
def test_blocks_of_every_delta_width(bits, n):
    rng = np.random.default_rng(bits * 1000 + n)
    if bits == 64:                                       # an unsorted stream: differences wrap around 2^64
        km = rng.integers(0, 2 ** 64, n, dtype=np.uint64)
    elif bits == 0:
        km = np.full(n, 12345678901234567, dtype=np.uint64)
    else:
        d = rng.integers(0, 1 << bits, n, dtype=np.uint64) >> np.uint64(max(0, 2 + int(np.log2(n)) - (64 - bits)) if bits > 40 else 0)
        km = np.cumsum(d, dtype=np.uint64) + np.uint64(17)
    ct = rng.integers(1, 400, n).astype(np.uint32)
    ct[rng.integers(0, n, max(1, n // 50))] = np.uint32(2 ** 32 - 1)
    ct[rng.integers(0, n, max(1, n // 50))] = np.uint32(255)
    ct[rng.integers(0, n, max(1, n // 50))] = np.uint32(254)
    size = roundtrip(km, ct)
    assert size <= ((n + 255) // 256) * K._native.lib().kmd_pack_block_bound()


def test_size_on_streams_like_the_synthetic_partitions():
    """A sample's stream of a 2 M-row partition: k-mers every ~1.5 rows of 2^21, counts mostly below 255."""
    rng = np.random.default_rng(1)
    n = 100_000
    km = np.cumsum(rng.geometric(0.65, n).astype(np.uint64) << np.uint64(21), dtype=np.uint64) + rng.integers(0, 1 << 20, n, dtype=np.uint64)
    km.sort()
    ct = rng.poisson(8, n).astype(np.uint32) + 1
    ct[rng.random(n) < 0.01] = 3000
    size = roundtrip(km, ct)
    assert 3.5 < size / n < 5.0, size / n                # (12 as plain arrays)


def test_bad_arguments_pack_nothing():
    L = K._native.lib()
    buf = np.zeros(int(L.kmd_pack_block_bound()), dtype=np.uint8)
    km, ct = np.arange(300, dtype=np.uint64), np.ones(300, dtype=np.uint32)
    assert L.kmd_pack_block(km.ctypes.data, ct.ctypes.data, 0, buf.ctypes.data) == 0
    assert L.kmd_pack_block(km.ctypes.data, ct.ctypes.data, 257, buf.ctypes.data) == 0
    assert L.kmd_pack_block(None, ct.ctypes.data, 5, buf.ctypes.data) == 0
    assert L.kmd_pack_block_bound() == 16 + 257 * 8 + 256 + 1024


@pytest.mark.parametrize("count_bytes", [1, 2, 4])
def test_pack_records_takes_a_kmer_files_records_as_they_are(count_bytes):
    """kmd_pack_records (what `kmdiff-hip diff`'s decoder threads call on the LZ4 decoder's output: [k-mer, 8 bytes][count,
    1 / 2 / 4 bytes] one behind the other) writes exactly what kmd_pack_block writes for the same records taken apart --
    every block length from 1 to 256 (the vector path's groups of eight and what is left over), records at any byte offset,
    nothing read beyond the n records (the buffer ends with them), bad arguments pack nothing."""
    L = K._native.lib()
    rng = np.random.default_rng(50 + count_bytes)
    rec = 8 + count_bytes
    bound = int(L.kmd_pack_block_bound())
    top = (1 << (8 * count_bytes)) - 1
    for n in list(range(1, 41)) + [63, 64, 65, 127, 128, 200, 255, 256]:
        km = np.cumsum(rng.integers(1, 1 << int(rng.integers(3, 50)), n, dtype=np.uint64)).astype(np.uint64)
        ct = rng.integers(1, min(top, 300) + 1, n, dtype=np.uint64).astype(np.uint32)
        if count_bytes == 4: ct[rng.random(n) < 0.05] = 0xFFFFFFF0
        raw = np.zeros(n * rec, dtype=np.uint8)
        r2 = raw.reshape(n, rec)
        r2[:, :8] = km.view(np.uint8).reshape(n, 8)
        r2[:, 8:] = ct.astype("<u4").view(np.uint8).reshape(n, 4)[:, :count_bytes]
        want, got = np.zeros(bound, dtype=np.uint8), np.zeros(bound, dtype=np.uint8)
        nw = int(L.kmd_pack_block(km.ctypes.data, ct.ctypes.data, n, want.ctypes.data))
        for shift in (0, 1, 3):                                  # (the decoder's chunks start anywhere)
            buf = np.zeros(shift + n * rec, dtype=np.uint8)
            buf[shift:] = raw
            got[:] = 0
            ng = int(L.kmd_pack_records(buf.ctypes.data + shift, count_bytes, n, got.ctypes.data))
            assert ng == nw and (got[:ng] == want[:nw]).all(), (n, count_bytes, shift)
    buf = np.zeros(300 * rec, dtype=np.uint8)
    out = np.zeros(bound, dtype=np.uint8)
    assert L.kmd_pack_records(buf.ctypes.data, count_bytes, 0, out.ctypes.data) == 0
    assert L.kmd_pack_records(buf.ctypes.data, count_bytes, 257, out.ctypes.data) == 0
    assert L.kmd_pack_records(buf.ctypes.data, 3, 5, out.ctypes.data) == 0
    assert L.kmd_pack_records(None, count_bytes, 5, out.ctypes.data) == 0


@pytest.mark.parametrize("n", [1, 255, 256, 257, 5000])
def test_pack_stream_is_its_blocks_one_behind_the_other(n):
    """kmd_pack_stream (a whole stream in one call: what bench.py's feed-inclusive leg and a host's decoder thread use) writes
    exactly what kmd_pack_block writes block by block, with the block table beside it; a buffer that is too small is
    refused (0 bytes), one that is just large enough is filled to the byte."""
    from kmdiff_amd import _native as N
    L = N.lib()
    rng = np.random.default_rng(n)
    km = np.cumsum(rng.integers(1, 1 << 23, n, dtype=np.uint64), dtype=np.uint64)
    ct = rng.integers(1, 400, n).astype(np.uint32)
    packed, base, table, offs = K.pack_streams([(km, ct)])                  # block by block (kmd_pack_block)
    nb = (n + 255) // 256
    bound = int(L.kmd_pack_block_bound())
    out = np.zeros(nb * bound, dtype=np.uint8)
    tab = np.zeros(nb, dtype=np.uint32)
    got = int(L.kmd_pack_stream(km.ctypes.data, ct.ctypes.data, n, out.ctypes.data, out.nbytes, tab.ctypes.data))
    assert got == len(packed) and np.array_equal(out[:got], packed) and tab.tolist() == table.tolist()
    tight = np.zeros(got, dtype=np.uint8)
    assert int(L.kmd_pack_stream(km.ctypes.data, ct.ctypes.data, n, tight.ctypes.data, tight.nbytes, tab.ctypes.data)) == got
    assert np.array_equal(tight, packed)
    small = np.zeros(max(got - 8, 1), dtype=np.uint8)
    assert int(L.kmd_pack_stream(km.ctypes.data, ct.ctypes.data, n, small.ctypes.data, small.nbytes, tab.ctypes.data)) == 0
    assert int(L.kmd_pack_stream(km.ctypes.data, ct.ctypes.data, 0, out.ctypes.data, out.nbytes, tab.ctypes.data)) == 0


def test_host_packer_under_the_sanitizers():
    """tests/pack_asan.cpp: kmd_pack_host.cpp alone, CPU build with AddressSanitizer + UBSan -- 6000 random blocks through
    kmd_pack_records (every count width, any alignment, heap buffers that end with the last record: its AVX2 path reads 96
    bytes at a time) == kmd_pack_block; kmd_pack_stream into a buffer of exactly the bytes it takes, refused 8 bytes short."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "kmdiff_amd", "host"), "../bin/pack_asan"], check=True, capture_output=True)
    r = subprocess.run([os.path.join(root, "kmdiff_amd", "bin", "pack_asan"), "6000", "11"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "pack_asan ok: 6000 blocks" in r.stdout, r.stdout + r.stderr
