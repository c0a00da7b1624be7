"""kmtricks on-disk formats (test infrastructure): the byte layouts SURVEY.md 8f derives from the
reference's fixture tests/data_test/km_out_dir (written by kmtricks v1.1.1).  Read AND write, so
that tests can fabricate run directories from synthetic matrices.

  <id>.kmer.lz4 : 41-byte header  "kmtricks" u32 0  u8 compressed  "kmer\\0\\0\\0\\0"  u32 k
                  u32 kmer_slots  u32 count_bytes  u32 sample_id  u32 partition ; then one LZ4
                  frame of records [u64 kmer LE x slots][count LE], ascending by k-mer
  <id>.hist     : "kmtricks" u32 0 u8 0 "khist\\0\\0\\0" u32 k u32 id  u64 lower u64 upper
                  u64 uniq u64 total  4 x u64 out-of-bounds  u64 hist_u[upper-lower+1]
                  u64 hist_n[upper-lower+1]
"""
import ctypes as C
import os
import struct

import numpy as np

_L = None


def _lz4():
    global _L
    if _L is None:
        L = C.CDLL("liblz4.so.1")
        L.LZ4F_createDecompressionContext.restype = C.c_size_t
        L.LZ4F_createDecompressionContext.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
        L.LZ4F_freeDecompressionContext.argtypes = [C.c_void_p]
        L.LZ4F_decompress.restype = C.c_size_t
        L.LZ4F_decompress.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p,
                                      C.POINTER(C.c_size_t), C.c_void_p]
        L.LZ4F_isError.argtypes = [C.c_size_t]
        L.LZ4F_compressFrameBound.restype = C.c_size_t
        L.LZ4F_compressFrameBound.argtypes = [C.c_size_t, C.c_void_p]
        L.LZ4F_compressFrame.restype = C.c_size_t
        L.LZ4F_compressFrame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        _L = L
    return _L


def lz4_frame_decode(src):
    L = _lz4()
    ctx = C.c_void_p()
    assert L.LZ4F_createDecompressionContext(C.byref(ctx), 100) == 0
    out = bytearray()
    buf = C.create_string_buffer(1 << 20)
    pos = 0
    src = bytes(src)
    while pos < len(src):
        dn = C.c_size_t(len(buf))
        sn = C.c_size_t(len(src) - pos)
        r = L.LZ4F_decompress(ctx, buf, C.byref(dn), src[pos:], C.byref(sn), None)
        assert not L.LZ4F_isError(r), "LZ4F_decompress failed"
        out += buf.raw[:dn.value]
        pos += sn.value
        if r == 0 and sn.value == 0:
            break
    L.LZ4F_freeDecompressionContext(ctx)
    return bytes(out)


def lz4_frame_encode(raw):
    L = _lz4()
    cap = L.LZ4F_compressFrameBound(len(raw), None)
    buf = C.create_string_buffer(cap)
    n = L.LZ4F_compressFrame(buf, cap, raw, len(raw), None)
    assert not L.LZ4F_isError(n)
    return buf.raw[:n]


def read_kmer_file(path):
    """Returns (header dict, kmers uint64[n] (low limb), counts uint32[n])."""
    d = open(path, "rb").read()
    magic, _, comp, kind, k, slots, cbytes, sid, part = struct.unpack("<8sIB8sIIIII", d[:41])
    assert magic == b"kmtricks" and kind.rstrip(b"\0") == b"kmer"
    raw = lz4_frame_decode(d[41:]) if comp else d[41:]
    rec = 8 * slots + cbytes
    n = len(raw) // rec
    a = np.frombuffer(raw, dtype=np.uint8, count=n * rec).reshape(n, rec)
    kmers = a[:, :8].copy().view("<u8").reshape(n)
    cdt = {1: "<u1", 2: "<u2", 4: "<u4"}[cbytes]
    counts = a[:, 8 * slots:8 * slots + cbytes].copy().view(cdt).reshape(n).astype(np.uint32)
    return {"k": k, "slots": slots, "count_bytes": cbytes, "sample_id": sid, "partition": part}, kmers, counts


def write_kmer_file(path, k, sample_id, partition, kmers, counts, count_bytes=4, compressed=True, kmers_hi=None):
    """kmers_hi: high limbs for 32 < k <= 64 (two slots, low limb first -- unpinned, see kmtricks_io.hpp)."""
    kmers = np.ascontiguousarray(kmers, dtype="<u8")
    counts = np.ascontiguousarray(np.asarray(counts).astype({1: "<u1", 2: "<u2", 4: "<u4"}[count_bytes]))
    n = len(kmers)
    slots = 1 if kmers_hi is None else 2
    rec = np.zeros((n, 8 * slots + count_bytes), dtype=np.uint8)
    rec[:, :8] = kmers.view(np.uint8).reshape(n, 8)
    if kmers_hi is not None:
        rec[:, 8:16] = np.ascontiguousarray(kmers_hi, dtype="<u8").view(np.uint8).reshape(n, 8)
    rec[:, 8 * slots:] = counts.view(np.uint8).reshape(n, count_bytes)
    raw = rec.tobytes()
    hdr = struct.pack("<8sIB8sIIIII", b"kmtricks", 0, 1 if compressed else 0, b"kmer\0\0\0\0", k, slots,
                      count_bytes, sample_id, partition)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "wb") as f:
        f.write(hdr + (lz4_frame_encode(raw) if compressed else raw))


def read_hist(path):
    d = open(path, "rb").read()
    magic, _, comp, kind = struct.unpack("<8sIB8s", d[:21])
    assert magic == b"kmtricks" and kind.rstrip(b"\0") == b"khist"
    k, sid = struct.unpack("<II", d[21:29])
    lower, upper, uniq, total, oob_lu, oob_uu, oob_ln, oob_un = struct.unpack("<8Q", d[29:93])
    n = upper - lower + 1
    hist_u = np.frombuffer(d, dtype="<u8", count=n, offset=93)
    hist_n = np.frombuffer(d, dtype="<u8", count=n, offset=93 + 8 * n)
    return {"k": k, "id": sid, "lower": lower, "upper": upper, "uniq": uniq, "total": total,
            "hist_u": hist_u.copy(), "hist_n": hist_n.copy()}


def write_hist(path, k, sample_id, counts, lower=1, upper=255):
    counts = np.asarray(counts, dtype=np.uint64)
    counts = counts[counts > 0]
    n = upper - lower + 1
    hu = np.zeros(n, dtype="<u8")
    hn = np.zeros(n, dtype="<u8")
    inb = counts[(counts >= lower) & (counts <= upper)]
    np.add.at(hu, (inb - lower).astype(np.int64), 1)
    np.add.at(hn, (inb - lower).astype(np.int64), inb)
    hi = counts[counts > upper]
    hdr = struct.pack("<8sIB8sII8Q", b"kmtricks", 0, 0, b"khist\0\0\0", k, sample_id, lower, upper,
                      len(counts), int(counts.sum()), 0, len(hi), 0, int(hi.sum()))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "wb") as f:
        f.write(hdr + hu.tobytes() + hn.tobytes())


def read_fof(path):
    ids = []
    for line in open(path):
        line = line.strip()
        if line:
            ids.append(line.split(":")[0].strip())
    return ids


def write_run_dir(root, k, sample_ids, partitions, abundance_min=1):
    """partitions: list over partitions of a list over samples of (kmers, counts[, kmers_hi])."""
    os.makedirs(root, exist_ok=True)
    with open(os.path.join(root, "kmtricks.fof"), "w") as f:
        for s in sample_ids:
            f.write("%s : ./fasta/%s.fasta\n" % (s, s))
    with open(os.path.join(root, "kmdiff-count.opt"), "w") as f:
        f.write("Options: verbosity=info,nb_threads=8,file=fof.txt,dir=%s,kmer_size=%d,abundance_min=%d,"
                "recurrence_min=1,memory=0,minimizer_type=0,minimizer_size=10,repartition_type=0,"
                "nb_partitions=0,\n" % (root, k, abundance_min))
    per_sample = [[] for _ in sample_ids]
    for p, streams in enumerate(partitions):
        for s, stream in enumerate(streams):
            km, ct = stream[0], stream[1]
            write_kmer_file(os.path.join(root, "counts", "partition_%d" % p, "%s.kmer.lz4" % sample_ids[s]),
                            k, s, p, km, ct, kmers_hi=stream[2] if len(stream) > 2 else None)
            per_sample[s].append(np.asarray(ct, dtype=np.uint64))
    for s, name in enumerate(sample_ids):
        allc = np.concatenate(per_sample[s]) if per_sample[s] else np.zeros(0, np.uint64)
        write_hist(os.path.join(root, "histograms", "%s.hist" % name), k, s, allc)
    os.makedirs(os.path.join(root, "matrices"), exist_ok=True)


def write_matrix_file(path, k, partition, kmers, counts, count_bytes=4):
    """<run>/matrices/*: header BY ANALOGY with the k-mer file header (no fixture, kmtricks absent):
    13-byte base header, "matrix\\0\\0", u32 k, u32 slots, u32 count_bytes, u32 nb_counts, u32 id,
    u32 partition; then one LZ4 frame of rows [u64 kmer][count x nb_counts]."""
    kmers = np.asarray(kmers, dtype="<u8")
    counts = np.asarray(counts).astype({1: "<u1", 2: "<u2", 4: "<u4"}[count_bytes])
    n, S = counts.shape
    rec = np.zeros((n, 8 + S * count_bytes), dtype=np.uint8)
    rec[:, :8] = kmers.view(np.uint8).reshape(n, 8)
    rec[:, 8:] = np.ascontiguousarray(counts).view(np.uint8).reshape(n, S * count_bytes)
    hdr = struct.pack("<8sIB8sIIIIII", b"kmtricks", 0, 1, b"matrix\0\0", k, 1, count_bytes, S, 0, partition)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "wb") as f:
        f.write(hdr + lz4_frame_encode(rec.tobytes()))


def read_matrix_file(path):
    d = open(path, "rb").read()
    magic, _, comp, kind, k, slots, cbytes, S, sid, part = struct.unpack("<8sIB8sIIIIII", d[:45])
    assert magic == b"kmtricks" and kind.rstrip(b"\0") == b"matrix" and slots == 1
    raw = lz4_frame_decode(d[45:]) if comp else d[45:]
    rec = 8 + S * cbytes
    n = len(raw) // rec
    a = np.frombuffer(raw, dtype=np.uint8, count=n * rec).reshape(n, rec)
    kmers = a[:, :8].copy().view("<u8").reshape(n)
    counts = a[:, 8:].copy().view({1: "<u1", 2: "<u2", 4: "<u4"}[cbytes]).reshape(n, S).astype(np.uint32)
    return {"k": k, "count_bytes": cbytes, "nb_counts": S, "partition": part}, kmers, counts


def read_survivor_file(path, kmer_bytes=8):
    """partitions/p<i>_uncorrected (FileAccumulator<KmerSign<32>>, WITH_POPSTRAT): one LZ4 frame of
    [kmer u64][p f64][sign i32][mean_control f64][mean_case f64][n u16][n x f64]."""
    d = open(path, "rb").read()
    raw = lz4_frame_decode(d) if d else b""
    out = {"kmer": [], "kmer_hi": [], "p": [], "sign": [], "mc": [], "mk": [], "counts": []}
    pos = 0
    while pos < len(raw):
        if kmer_bytes == 16:
            out["kmer_hi"].append(struct.unpack_from("<Q", raw, pos + 8)[0])
        km = struct.unpack_from("<Q", raw, pos)[0]
        p, sg, mc, mk, n = struct.unpack_from("<diddH", raw, pos + kmer_bytes)
        pos += kmer_bytes + 30
        out["kmer"].append(km); out["p"].append(p); out["sign"].append(sg); out["mc"].append(mc); out["mk"].append(mk)
        out["counts"].append(np.frombuffer(raw, dtype="<f8", count=n, offset=pos).copy())
        pos += 8 * n
    assert pos == len(raw)
    return out


def kmer_to_string2(hi, lo, k):
    v = (int(hi) << 64) | int(lo)
    return "".join("ACTG"[(v >> (2 * (k - 1 - i))) & 3] for i in range(k))


def kmer_to_string(v, k):
    return "".join("ACTG"[(int(v) >> (2 * (k - 1 - i))) & 3] for i in range(k))


def read_kff(path):
    """A KFF 1.0 file as kmdiff writes it (include/kmdiff/kff_utils.hpp:32-107: one global-variable section
    with k / max / data_size, one raw section, max = 1 and data_size = 0): (variables, encoding byte, k-mers
    as strings).  Written from the format description -- kff-cpp-api is not in the tree: UNPINNED."""
    with open(path, "rb") as f:
        b = f.read()
    assert b[:3] == b"KFF" and b[-3:] == b"KFF", "signature"
    major, minor, enc, uniq, canon = b[3], b[4], b[5], b[6], b[7]
    assert (major, minor) == (1, 0) and uniq == 0 and canon == 0
    free = struct.unpack(">I", b[8:12])[0]
    pos = 12 + free
    letters = [None] * 4
    for nt, shift in zip("ACGT", (6, 4, 2, 0)):
        letters[(enc >> shift) & 3] = nt
    assert None not in letters, "encoding is not a permutation"
    variables, kmers = {}, []
    while pos < len(b) - 3:
        kind = chr(b[pos]); pos += 1
        if kind == "v":
            nv = struct.unpack(">Q", b[pos:pos + 8])[0]; pos += 8
            for _ in range(nv):
                end = b.index(b"\0", pos)
                name = b[pos:end].decode(); pos = end + 1
                variables[name] = struct.unpack(">Q", b[pos:pos + 8])[0]; pos += 8
        elif kind == "r":
            k, mx, ds = variables["k"], variables["max"], variables["data_size"]
            assert mx == 1 and ds == 0
            nb = struct.unpack(">Q", b[pos:pos + 8])[0]; pos += 8
            nbytes = (k + 3) // 4
            for _ in range(nb):
                v = int.from_bytes(b[pos:pos + nbytes], "big"); pos += nbytes
                assert v >> (2 * k) == 0, "bits above the sequence"
                kmers.append("".join(letters[(v >> (2 * (k - 1 - i))) & 3] for i in range(k)))
        else:
            raise AssertionError("unknown section %r at %d" % (kind, pos - 1))
    assert pos == len(b) - 3
    assert variables.get("first_index") == 0 and variables.get("footer_size") == 9 + 2 * (12 + 8)
    return variables, enc, kmers
