"""oracle/cpu_pipeline -- the CPU doing `kmdiff-hip diff`'s stage 1 on a run directory (liblz4 decode + the oracle's merge +
the oracle's test, one task per partition): the like-for-like baseline of tools/cli_throughput.py --cpu-baseline and
bench.py --e2e.  Held here to the reference's fixture (tests/merge_test.cpp:39-45: 320 merged rows, none significant) and
to the oracle called directly on a fabricated run directory."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "cpu_pipeline")


@pytest.fixture(scope="module")
def exe():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "cpu_pipeline"], stdout=subprocess.DEVNULL)
    return EXE


def run(exe, *args):
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return json.loads(r.stdout)


def test_reference_fixture(exe, golden_dir):
    out = run(exe, os.path.join(golden_dir, "km_out_dir"), 1, 1, 2)
    assert (out["partitions"], out["rows"], out["records"], out["survivors"]) == (4, 320, 320, 0)


def test_wrong_sample_split_is_refused(exe, golden_dir):
    r = subprocess.run([exe, os.path.join(golden_dir, "km_out_dir"), "2", "2", "1"], capture_output=True, text=True)
    assert r.returncode != 0 and "samples" in r.stderr


def test_fabricated_run_directory_equals_the_oracle_called_directly(exe, oracle, tmp_path):
    import kmtricks_files as KF
    import oracle_lib as OL
    seed, nc, nk, n, parts_n, thr = 0x6B6D64696666, 5, 4, 6000, 3, 0.01
    S = nc + nk
    parts, hosts = [], []
    for p in range(parts_n):
        host, lo, _ = oracle.synth_rows(seed, p, 0, n, nc, nk, 4)
        parts.append([(lo[host[:, s] > 0], host[host[:, s] > 0, s]) for s in range(S)])
        hosts.append(host)
    ids = ["C%d" % i for i in range(nc)] + ["K%d" % i for i in range(nk)]
    KF.write_run_dir(str(tmp_path / "km"), 31, ids, parts)
    tot = sum(h.sum(axis=0, dtype=np.uint64) for h in hosts)
    tc, tk = int(tot[:nc].sum()), int(tot[nc:].sum())
    want_rows, want_surv = 0, 0
    lf = oracle.lf_table(10000)
    for h in hosts:
        rows = h[(h > 0).any(axis=1)]                      # (a row nobody holds is no row of the merge)
        want_rows += len(rows)
        want_surv += len(oracle.diff_partition(rows, OL.LAYOUT_ROWS, nc, nk, tc, tk, lf, thr)["row"])
    for threads in (1, 3):
        out = run(exe, tmp_path / "km", nc, nk, threads, thr)
        assert (out["partitions"], out["rows"], out["survivors"], out["threads"]) == (parts_n, want_rows, want_surv, threads)
        assert out["records"] == sum(int((h > 0).sum()) for h in hosts)
