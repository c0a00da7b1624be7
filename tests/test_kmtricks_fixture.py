"""The reference's own integration fixture (tests/data_test/km_out_dir, copied as DATA under
tests/golden/km_out_dir): what tests/merge_test.cpp:12-46 asserts about it -- 4 partitions,
1 control + 1 case, k = 20, per-sample totals 160/160, 320 merged rows, 0 significant at
0.05/10000 -- checked for the oracle's merge + model and for the format reader/writer."""
import os

import numpy as np

import kmtricks_files as KF
import oracle_lib as OL

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "km_out_dir")


def load_fixture():
    ids = KF.read_fof(os.path.join(FIX, "kmtricks.fof"))
    parts = []
    for p in range(4):
        streams = []
        for s in ids:
            h, km, ct = KF.read_kmer_file(os.path.join(FIX, "counts", "partition_%d" % p, "%s.kmer.lz4" % s))
            assert h["k"] == 20 and h["partition"] == p and h["count_bytes"] == 4
            assert (np.diff(km.astype(np.int64)) > 0).all()
            streams.append((km, ct))
        parts.append(streams)
    totals = [KF.read_hist(os.path.join(FIX, "histograms", "%s.hist" % s))["total"] for s in ids]
    return ids, parts, totals


def test_fixture_facts_of_merge_test(oracle):
    ids, parts, totals = load_fixture()
    assert ids == ["Control1", "Case1"]
    assert totals == [160, 160]                      # merge_test.cpp:39-41 (get_total_kmer)
    rows = n_sig = 0
    lf = oracle.lf_table(10000)
    for streams in parts:
        mat, kmers = oracle.merge_partition(streams)
        assert (np.diff(kmers.astype(np.int64)) > 0).all()
        out = oracle.diff_partition(mat, OL.LAYOUT_ROWS, 1, 1, totals[0], totals[1], lf, 0.05 / 10000)
        rows += mat.shape[0]
        n_sig += out["counters"][1]
    assert rows == 320                               # merge_test.cpp:43
    assert n_sig == 0                                # merge_test.cpp:44


def test_format_writer_round_trips(tmp_path, oracle):
    ids, parts, totals = load_fixture()
    KF.write_run_dir(str(tmp_path / "run"), 20, ids, parts)
    for p in range(4):
        for s, name in enumerate(ids):
            h, km, ct = KF.read_kmer_file(str(tmp_path / "run" / "counts" / ("partition_%d" % p) / (name + ".kmer.lz4")))
            assert (km == parts[p][s][0]).all() and (ct == parts[p][s][1]).all() and h["sample_id"] == s
    for s, name in enumerate(ids):
        h = KF.read_hist(str(tmp_path / "run" / "histograms" / (name + ".hist")))
        ref = KF.read_hist(os.path.join(FIX, "histograms", name + ".hist"))
        assert (h["total"], h["uniq"]) == (ref["total"], ref["uniq"]) == (160, 160)
        assert (h["hist_u"] == ref["hist_u"]).all() and (h["hist_n"] == ref["hist_n"]).all()
    assert KF.kmer_to_string(parts[0][0][0][0], 20) == "AATATACTATATAATATATA"


def test_oracle_merge_random_streams(oracle):
    rng = np.random.default_rng(3)
    universe = np.unique(rng.integers(0, 1 << 40, 5000, dtype=np.uint64))
    streams = []
    dense = np.zeros((len(universe), 7), dtype=np.uint32)
    for s in range(7):
        pick = rng.random(len(universe)) < (0.1 + 0.1 * s)
        cnt = rng.integers(1, 1000, pick.sum()).astype(np.uint32)
        streams.append((universe[pick], cnt))
        dense[pick, s] = cnt
    streams[3] = (np.zeros(0, np.uint64), np.zeros(0, np.uint32))      # an empty sample
    dense[:, 3] = 0
    keep = dense.sum(axis=1) > 0
    mat, kmers = oracle.merge_partition(streams)
    assert (kmers == universe[keep]).all() and (mat == dense[keep]).all()
