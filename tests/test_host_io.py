"""Host-side file formats (kmdiff_amd/host/kmtricks_io.cpp) against the independent Python
reader/writer of tests/kmtricks_files.py, both directions, no GPU:
  matrices/*                      the alternate feed (matrix_proxy, merge.hpp:194-203) and --save-sk
  partitions/p<i>_uncorrected     FileAccumulator<KmerSign> records (kmer.hpp:113-127)
  options.bin                     dump_opt / load_opt / compare_opt (cmd/diff_opt.hpp:78-133)
The matrix header is by analogy with the k-mer file header (no fixture, kmtricks absent): these
tests pin the C++ and Python sides to each other, not to kmtricks."""
import os
import struct
import subprocess

import numpy as np
import pytest

import kmtricks_files as KF

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "kmdiff_amd", "bin", "io_roundtrip")


def tool(*args):
    if not os.path.exists(TOOL):
        subprocess.run(["make", "-C", os.path.join(ROOT, "kmdiff_amd", "host"), "../bin/io_roundtrip"], check=True,
                       capture_output=True)
    r = subprocess.run([TOOL] + [str(a) for a in args], capture_output=True, text=True, timeout=120)
    return r.returncode, r.stdout, r.stderr


@pytest.mark.parametrize("count_bytes", [1, 2, 4])
def test_matrix_file_both_directions(tmp_path, count_bytes):
    rng = np.random.default_rng(3)
    n, S = 70_000, 7                                    # several 64 KB LZ4 blocks
    km = np.sort(rng.integers(0, 1 << 62, n, dtype=np.uint64))
    cnt = rng.integers(0, 1 << (8 * count_bytes), (n, S), dtype=np.uint64).astype(np.uint32)
    KF.write_matrix_file(str(tmp_path / "in.lz4"), 31, 5, km, cnt, count_bytes)
    rc, out, err = tool("matrix", tmp_path / "in.lz4", tmp_path / "out.lz4")
    assert rc == 0, err
    assert out.split() == ["rows=%d" % n, "k=31", "count_bytes=%d" % count_bytes, "nb_counts=%d" % S, "partition=5"]
    hdr, km2, cnt2 = KF.read_matrix_file(str(tmp_path / "out.lz4"))
    assert hdr == {"k": 31, "count_bytes": count_bytes, "nb_counts": S, "partition": 5}
    assert (km2 == km).all() and (cnt2 == cnt).all()


def test_matrix_file_rejects_other_kmtricks_files(tmp_path):
    KF.write_kmer_file(str(tmp_path / "x.kmer.lz4"), 31, 0, 0, [1, 2, 3], [1, 1, 1])
    rc, _, err = tool("matrix", tmp_path / "x.kmer.lz4", tmp_path / "o")
    assert rc == 1 and "not a kmtricks count matrix" in err


@pytest.mark.parametrize("n,S", [(0, 9), (1, 0), (5000, 40)])
def test_survivor_file_both_directions(tmp_path, n, S):
    rng = np.random.default_rng(4)
    raw = b""
    recs = []
    for i in range(n):
        rec = (int(rng.integers(0, 1 << 62)), float(rng.random() * 1e-7), int(rng.integers(0, 3)), float(rng.integers(0, 500)),
               float(rng.integers(0, 500)))
        counts = rng.integers(0, 300, S).astype("<f8")
        raw += struct.pack("<QdiddH", *rec, S) + counts.tobytes()
        recs.append((rec, counts))
    with open(tmp_path / "p0_uncorrected", "wb") as f:
        f.write(KF.lz4_frame_encode(raw))
    rc, out, err = tool("survivors", tmp_path / "p0_uncorrected", tmp_path / "back")
    assert rc == 0, err
    assert out.split()[0] == "records=%d" % n
    got = KF.read_survivor_file(str(tmp_path / "back"))
    assert len(got["p"]) == n
    for i, (rec, counts) in enumerate(recs):
        assert (got["kmer"][i], got["p"][i], got["sign"][i], got["mc"][i], got["mk"][i]) == rec
        assert (got["counts"][i] == counts).all()


def test_two_limb_kmers_in_kmer_matrix_and_survivor_files(tmp_path):
    """32 < k <= 64: two 64-bit slots per k-mer, low limb first (unpinned: no fixture with k > 32)."""
    rng = np.random.default_rng(6)
    n, S = 1000, 3
    lo = rng.integers(0, 1 << 63, n, dtype=np.uint64); hi = np.sort(rng.integers(0, 1 << 30, n, dtype=np.uint64))
    cnt = rng.integers(0, 1000, (n, S)).astype(np.uint32)
    rec = np.zeros((n, 16 + 4 * S), dtype=np.uint8)
    rec[:, :8] = lo.view(np.uint8).reshape(n, 8); rec[:, 8:16] = hi.view(np.uint8).reshape(n, 8)
    rec[:, 16:] = cnt.view(np.uint8).reshape(n, 4 * S)
    hdr = struct.pack("<8sIB8sIIIIII", b"kmtricks", 0, 1, b"matrix\0\0", 63, 2, 4, S, 0, 1)
    open(tmp_path / "m.lz4", "wb").write(hdr + KF.lz4_frame_encode(rec.tobytes()))
    rc, out, err = tool("matrix", tmp_path / "m.lz4", tmp_path / "m2.lz4")
    assert rc == 0 and out.startswith("two-limb rows=%d k=63" % n), err
    d = open(tmp_path / "m2.lz4", "rb").read()
    assert d[:45] == hdr and KF.lz4_frame_decode(d[45:]) == rec.tobytes()
    raw = b"".join(struct.pack("<QQdiddH", int(lo[i]), int(hi[i]), 1e-9 * i, i % 3, 1.0, 2.0, 2) + struct.pack("<dd", 3.0, 4.0)
                   for i in range(50))
    open(tmp_path / "s16", "wb").write(KF.lz4_frame_encode(raw))
    rc, out, err = tool("survivors16", tmp_path / "s16", tmp_path / "s16b")
    assert rc == 0 and out.split()[0] == "records=50", err
    got = KF.read_survivor_file(str(tmp_path / "s16b"), kmer_bytes=16)
    assert got["kmer"] == lo[:50].tolist() and got["kmer_hi"] == hi[:50].tolist() and got["sign"] == [i % 3 for i in range(50)]


def test_survivor_file_truncated_record_is_an_error(tmp_path):
    raw = struct.pack("<QdiddH", 5, 1e-9, 1, 2.0, 3.0, 4) + b"\0" * 8       # says 4 counts, holds 1
    with open(tmp_path / "bad", "wb") as f:
        f.write(KF.lz4_frame_encode(raw))
    rc, _, err = tool("survivors", tmp_path / "bad", tmp_path / "o")
    assert rc == 1 and "malformed" in err


def test_options_bin_layout(tmp_path):
    """threshold f64, cutoff f64, correction i32, pop_correction u8, kmer_pca f64, npc u64 = 37 bytes."""
    blob = struct.pack("<ddi?dQ", 0.05, 100000.0, 2, True, 0.001, 3)
    assert len(blob) == 37
    open(tmp_path / "options.bin", "wb").write(blob)
    rc, out, err = tool("options", tmp_path / "options.bin", tmp_path / "back.bin")
    assert rc == 0, err
    assert "correction=2 pop=1" in out and "npc=3" in out and "action_same=0" in out
    assert open(tmp_path / "back.bin", "rb").read() == blob
    open(tmp_path / "short.bin", "wb").write(blob[:20])
    assert tool("options", tmp_path / "short.bin", tmp_path / "x")[0] == 1


@pytest.mark.parametrize("count_bytes,two,compressed,n", [(4, False, True, 300_000), (2, False, True, 70_001), (1, True, True, 90_000),
                                                          (4, True, True, 200_003), (4, False, False, 100_000), (4, False, True, 0),
                                                          (4, False, True, 1)])
def test_streaming_kmer_reader(tmp_path, count_bytes, two, compressed, n):
    """stream_kmer_file (LZ4 chunk -> records -> caller's arrays; records straddle the 1 MB chunks: 12,
    10, 17 and 20 bytes do not divide 2^20) against the whole-file reader and the Python reader."""
    rng = np.random.default_rng(n + count_bytes)
    km = np.sort(rng.integers(0, 1 << 62, n, dtype=np.uint64))
    hi = np.sort(rng.integers(0, 1 << 60, n, dtype=np.uint64)) if two else None
    ct = rng.integers(1, 1 << (8 * count_bytes), n, dtype=np.uint64).astype(np.uint32)
    KF.write_kmer_file(str(tmp_path / "s.kmer.lz4"), 63 if two else 31, 3, 7, km, ct, count_bytes, compressed, hi)
    rc, out, err = tool("kmers", tmp_path / "s.kmer.lz4", tmp_path / "dump")
    assert rc == 0, err
    assert out.split() == ["records=%d" % n, "slots=%d" % (2 if two else 1), "count_bytes=%d" % count_bytes, "same=1"]
    raw = open(tmp_path / "dump", "rb").read()
    assert len(raw) == n * (8 * (2 if two else 1) + 4)
    assert (np.frombuffer(raw[:8 * n], "<u8") == km).all()
    if two:
        assert (np.frombuffer(raw[8 * n:16 * n], "<u8") == hi).all()
    assert (np.frombuffer(raw[-4 * n:] if n else b"", "<u4") == ct).all()


def test_streaming_kmer_reader_rejects_a_truncated_frame(tmp_path):
    rng = np.random.default_rng(9)
    n = 50_000
    km = np.sort(rng.integers(0, 1 << 62, n, dtype=np.uint64))
    KF.write_kmer_file(str(tmp_path / "s.kmer.lz4"), 31, 0, 0, km, np.ones(n, dtype=np.uint32))
    whole = open(tmp_path / "s.kmer.lz4", "rb").read()
    open(tmp_path / "cut.kmer.lz4", "wb").write(whole[:len(whole) // 2])
    rc, out, err = tool("kmers", tmp_path / "cut.kmer.lz4", tmp_path / "dump")
    assert rc == 1 and "truncated LZ4 frame" in err
    open(tmp_path / "nomark.kmer.lz4", "wb").write(whole[:-4])          # the 4-byte end mark cut off
    rc, out, err = tool("kmers", tmp_path / "nomark.kmer.lz4", tmp_path / "dump")
    assert rc == 1 and "truncated LZ4 frame" in err


@pytest.mark.parametrize("k", [20, 31, 32, 33, 47, 63, 64])
def test_kff_writer_round_trip(tmp_path, k):
    """-f / --kff-output (KffWriter, include/kmdiff/kff_utils.hpp:32-107): header, encoding {0, 1, 3, 2} = 0x1e,
    the k / max = 1 / data_size = 0 variables, one raw section of 2-bit sequences, footer -- read back by the
    independent Python reader.  UNPINNED against kff-cpp-api (absent from the reference tree)."""
    rng = np.random.default_rng(k)
    n = 200
    bits = 2 * k
    vals = [int(rng.integers(0, 1 << 62)) | (int(rng.integers(0, 1 << 62)) << 62) | (int(rng.integers(0, 16)) << 124) for _ in range(n)]
    vals = [v & ((1 << bits) - 1) for v in vals] + [0, (1 << bits) - 1]
    src = tmp_path / "kmers.txt"
    src.write_text("%d\n" % k + "".join("%d %d\n" % (v & (2 ** 64 - 1), v >> 64) for v in vals))
    rc, out, err = tool("kff", src, tmp_path / "out.kff")
    assert rc == 0, err
    variables, enc, kmers = KF.read_kff(tmp_path / "out.kff")
    assert enc == 0b00011110 and variables["k"] == k and variables["max"] == 1 and variables["data_size"] == 0
    want = [KF.kmer_to_string2(v >> 64, v & (2 ** 64 - 1), k) if k > 32 else KF.kmer_to_string(v, k) for v in vals]
    assert kmers == want
    assert os.path.getsize(tmp_path / "out.kff") == 12 + (1 + 8 + (2 + 8) + (4 + 8) + (10 + 8)) + (1 + 8) + len(vals) * ((k + 3) // 4) + 49 + 3


def test_case_sum_is_printed_like_fmt_braces():
    """FASTA header `case={}` (aggregator.hpp:51-55): fmt prints a double with the shortest digits that round-trip,
    in fixed notation -- no trailing .0 -- for 1e-4 <= |v| < 1e16 and in exponent notation (two exponent digits at
    least) outside.  Literal expectations, independent of the implementation."""
    exe = os.path.join(ROOT, "kmdiff_amd", "bin", "kmdiff-hip")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "kmdiff_amd", "host")], check=True, stdout=subprocess.DEVNULL)
    cases = [("10", "10"), ("100", "100"), ("1200", "1200"), ("1000000", "1000000"), ("1e15", "1000000000000000"),
             ("1e16", "1e+16"), ("3e22", "3e+22"), ("0.0001", "0.0001"), ("0.00001", "1e-05"), ("2.5e-5", "2.5e-05"),
             ("1e-7", "1e-07"), ("0.1", "0.1"), ("0.3", "0.3"), ("1.5", "1.5"), ("123456789.125", "123456789.125"),
             ("1234567", "1234567"), ("0.3333333333333333", "0.3333333333333333"), ("5e-324", "5e-324"),
             ("1.7976931348623157e308", "1.7976931348623157e+308"), ("0", "0"), ("4294967296", "4294967296"),
             ("9007199254740993", "9007199254740992"), ("-2.5", "-2.5")]
    out = subprocess.run([exe, "fmt"] + [c[0] for c in cases], check=True, capture_output=True, text=True).stdout.split()
    assert out == [c[1] for c in cases]


FUZZ = os.path.join(ROOT, "kmdiff_amd", "bin", "io_fuzz")


@pytest.mark.timeout(600)
@pytest.mark.parametrize("what", ["kmers", "kmers16", "matrix", "survivors"])
def test_readers_on_damaged_files_under_sanitizers(tmp_path, what):
    """stream_kmer_file / read_kmer_file, stream_matrix_file / read_matrix_file, read_survivor_file on 1 000 truncated,
    bit-flipped, overwritten files each, in a CPU build with AddressSanitizer + UBSan (tests/io_fuzz.cpp): a reader may
    refuse a file or deliver fewer records; any out-of-bounds access, overflow or abort fails the run.  (Round 4's
    finding: a matrix header whose sample count was overwritten sized the decoder's row buffer -- gigabytes, half a minute under the
    sanitizer; the header is now checked, kMaxMatrixSamples.)"""
    if not os.path.exists(FUZZ):
        subprocess.run(["make", "-C", os.path.join(ROOT, "kmdiff_amd", "host"), "../bin/io_fuzz"], check=True, capture_output=True)
    rng = np.random.default_rng(11)
    n = 12_000                                          # three 64 KB LZ4 blocks
    km = np.sort(rng.integers(0, 1 << 62, n, dtype=np.uint64))
    seed = tmp_path / "seed"
    if what == "kmers":
        KF.write_kmer_file(str(seed), 31, 3, 1, km, rng.integers(1, 300, n).astype(np.uint32))
    elif what == "kmers16":
        KF.write_kmer_file(str(seed), 63, 3, 1, rng.integers(0, 1 << 62, n, dtype=np.uint64), rng.integers(1, 70000, n).astype(np.uint32),
                           kmers_hi=km)
    elif what == "matrix":
        KF.write_matrix_file(str(seed), 31, 2, km, rng.integers(0, 1 << 16, (n, 6), dtype=np.uint64).astype(np.uint32), 2)
    else:
        raw = b""
        for i in range(1500):
            S = 5
            raw += struct.pack("<QdiddH", int(km[i]), 1e-9 * i, i % 3, float(i), float(2 * i), S) + np.arange(S, dtype="<f8").tobytes()
        open(seed, "wb").write(KF.lz4_frame_encode(raw))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([FUZZ, "kmers" if what == "kmers16" else what, str(seed), str(tmp_path / "work"), "1000", "42"], capture_output=True, text=True,
                       timeout=550, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "1000 damaged files" in r.stdout and "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    refused = int(r.stdout.split("damaged files, ")[1].split(" refused")[0])
    assert 0 < refused < 1000                           # some mutations are caught, some leave a readable file
