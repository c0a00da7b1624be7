"""The C++ host (`kmdiff-hip diff`, kmdiff_amd/host/main.cpp) end to end on kmtricks run
directories: the reference's fixture and run directories fabricated from synthetic matrices,
checked against the oracle pipeline (merge -> Poisson test -> threshold -> corrector -> FASTA)."""
import json
import os
import subprocess

import numpy as np
import pytest

import kmtricks_files as KF
import oracle_lib as OL

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "kmdiff_amd", "bin", "kmdiff-hip")
SEED = 0x6B6D64696666


def run_cli(args, out):
    r = subprocess.run([CLI, "diff", "-o", str(out)] + [str(a) for a in args], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    s = json.load(open(os.path.join(str(out), "summary.json")))
    # how the k-mer feed's records crossed the link (absent when stage 1 did not run: a resume, the matrices/ feed) is
    # kept apart: the summaries of two runs are compared as a whole all over this file
    TRANSFER[str(out)] = s.pop("transfer", None)
    return s, r.stderr


TRANSFER = {}


def read_fasta(path):
    recs = []
    lines = open(path).read().split("\n")
    for i in range(0, len(lines) - 1, 2):
        recs.append((lines[i], lines[i + 1]))
    return recs


def fmt_shortest(v):
    """fmt's {} for a double: shortest round-trip digits, fixed notation for 1e-4 <= |v| < 1e16 (no trailing .0)"""
    r = repr(float(v))                                     # shortest round-trip digits (David Gay), like fmt
    if "e" in r or "inf" in r or "nan" in r:
        m, e = r.split("e")
        digits = m.replace(".", "").replace("-", "").rstrip("0") or "0"
        return ("-" if v < 0 else "") + digits[0] + ("." + digits[1:] if len(digits) > 1 else "") + "e%s%02d" % ("-" if int(e) < 0 else "+", abs(int(e)))
    return r[:-2] if r.endswith(".0") else r


def test_cli_on_reference_fixture(tmp_path):
    """tests/merge_test.cpp:39-45: 320 merged rows, 0 significant."""
    s, err = run_cli(["-d", os.path.join(ROOT, "tests", "golden", "km_out_dir"), "-1", 1, "-2", 1, "-u", 10000], tmp_path / "o")
    assert (s["total_kmers"], s["n_sig"], s["kept"], s["kmer_size"], s["nb_partitions"]) == (320, 0, 0, 20, 4)
    assert "0/320 significant k-mers." in err
    assert read_fasta(tmp_path / "o" / "control_kmers.fasta") == [] and read_fasta(tmp_path / "o" / "case_kmers.fasta") == []


@pytest.fixture(scope="module")
def synth_run(tmp_path_factory):
    """3 partitions x 40 000 rows, 5 controls + 4 cases, written in kmtricks' formats."""
    o = OL.load()
    nc, nk, n, k = 5, 4, 40_000, 31
    root = tmp_path_factory.mktemp("run")
    parts, mats, kms = [], [], []
    for p in range(3):
        host, lo, _ = o.synth_rows(SEED, p, 0, n, nc, nk, 4)
        parts.append([(lo[host[:, s] > 0], host[host[:, s] > 0, s]) for s in range(nc + nk)])
        mats.append(host)
        kms.append(lo)
    ids = ["C%d" % i for i in range(nc)] + ["K%d" % i for i in range(nk)]
    KF.write_run_dir(str(root / "km"), k, ids, parts)
    return str(root / "km"), nc, nk, k, mats, kms


def oracle_pipeline(o, nc, nk, mats, kms, thr1, correction, alpha):
    totals = np.sum([m.sum(axis=0, dtype=np.uint64) for m in mats], axis=0)
    lf = o.lf_table(10000)
    surv = {"kmer": [], "p": [], "sign": [], "mc": [], "mk": []}
    total = 0
    for m, km in zip(mats, kms):
        out = o.diff_partition(m, OL.LAYOUT_ROWS, nc, nk, int(totals[:nc].sum()), int(totals[nc:].sum()), lf, thr1)
        idx = out["row"].astype(np.int64)
        surv["kmer"] += km[idx].tolist(); surv["p"] += out["pvalue"].tolist(); surv["sign"] += out["sign"].tolist()
        surv["mc"] += out["mean_control"].tolist(); surv["mk"] += out["mean_case"].tolist()
        total += m.shape[0]
    keep = o.aggregate({"disabled": 0, "bonferroni": 1, "benjamini": 2, "sidak": 3, "holm": 4}[correction], alpha, total,
                       np.array(surv["p"]))
    return surv, keep, total


@pytest.mark.parametrize("correction", ["bonferroni", "benjamini", "disabled"])
def test_cli_matches_oracle_pipeline(synth_run, tmp_path, correction):
    run_dir, nc, nk, k, mats, kms = synth_run
    o = OL.load()
    s, _ = run_cli(["-d", run_dir, "-1", nc, "-2", nk, "-c", correction, "-s", 0.05, "-u", 1000], tmp_path / "o")
    surv, keep, total = oracle_pipeline(o, nc, nk, mats, kms, 0.05 / 1000, correction, 0.05)
    assert s["total_kmers"] == total and s["n_sig"] == len(surv["p"]) and s["kept"] == int(keep.sum())
    assert s["n_sig"] > 10
    order = list(range(len(keep)))
    if correction == "benjamini":
        order.sort(key=lambda i: surv["p"][i])
    want = {"control": [], "case": []}
    for i in order:
        if keep[i]:
            want["control" if surv["sign"][i] == 0 else "case"].append(i)
    for name in ("control", "case"):
        got = read_fasta(tmp_path / "o" / ("%s_kmers.fasta" % name))
        assert len(got) == len(want[name])
        for j, (i, (hdr, seq)) in enumerate(zip(want[name], got)):
            assert seq == KF.kmer_to_string(surv["kmer"][i], k)
            f = hdr[1:].split("_")
            assert f[0] == str(j) and f[2] == "control=%d" % int(surv["mc"][i]) and f[3] == "case=" + fmt_shortest(surv["mk"][i])
            pv = float(f[1].split("=")[1])
            assert abs(pv - surv["p"][i]) <= 1e-5 * surv["p"][i] + 1e-300          # %g keeps 6 digits


def test_cli_kff_output_holds_the_fasta_kmers(synth_run, tmp_path):
    """-f / --kff-output (aggregator.hpp:197, kff_utils.hpp:32-107): control_kmers.kff / case_kmers.kff hold the
    k-mers of the two FASTA files, in the same order; the case sum in a FASTA header is rendered like fmt's {}
    (plain digits for integral sums: 1200, not 1.2e+03)."""
    run_dir, nc, nk, k, mats, kms = synth_run
    run_cli(["-d", run_dir, "-1", nc, "-2", nk, "-u", 1000], tmp_path / "a")
    run_cli(["-d", run_dir, "-1", nc, "-2", nk, "-u", 1000, "-f"], tmp_path / "b")
    for name in ("control", "case"):
        fasta = read_fasta(tmp_path / "a" / ("%s_kmers.fasta" % name))
        variables, enc, kmers = KF.read_kff(tmp_path / "b" / ("%s_kmers.kff" % name))
        assert variables["k"] == k and enc == 0x1E and kmers == [seq for _, seq in fasta] and len(kmers) >= 3
        assert not os.path.exists(tmp_path / "b" / ("%s_kmers.fasta" % name))
        for hdr, _ in fasta:
            case = hdr.split("_case=")[1]
            assert case.isdigit(), hdr                    # integral sums below 1e16 print as integers


def test_cli_matrix_feed_equals_kmer_file_feed(synth_run, tmp_path):
    """cmd/diff.hpp:80-101,151-152: a non-empty matrices/ switches do_diff to matrix_proxy
    (merge.hpp:194-203, pre-merged rows): same survivors and decisions as the k-way merge feed."""
    import shutil
    run_dir, nc, nk, k, mats, kms = synth_run
    a, _ = run_cli(["-d", run_dir, "-1", nc, "-2", nk, "-u", 1000, "-t", 3], tmp_path / "a")
    mrun = tmp_path / "km_matrix"
    shutil.copytree(run_dir, mrun)
    for p, (m, km) in enumerate(zip(mats, kms)):
        KF.write_matrix_file(str(mrun / "matrices" / ("matrix_%d.count.lz4" % p)), k, p, km, m)
    b, _ = run_cli(["-d", mrun, "-1", nc, "-2", nk, "-u", 1000], tmp_path / "b")
    assert TRANSFER[str(tmp_path / "a")]["format"] == "packed" and TRANSFER[str(tmp_path / "b")] is None
    assert a == b and a["n_sig"] > 10
    for name in ("control_kmers.fasta", "case_kmers.fasta"):
        assert open(tmp_path / "a" / name).read() == open(tmp_path / "b" / name).read()


def test_cli_keep_tmp_files_resume_and_save_sk(synth_run, tmp_path):
    """--keep-tmp leaves partitions/p<i>_uncorrected in the reference's record format
    (kmer.hpp:113-127 through lz4_stream) and options.bin (diff_opt.hpp:78-88); a second run that
    only changes the corrector skips stage 1 (cmd/diff.hpp:310-337) and must equal a fresh run;
    --save-sk writes the significant rows (merge.hpp:83-86)."""
    import struct
    run_dir, nc, nk, k, mats, kms = synth_run
    o = OL.load()
    out = tmp_path / "o"
    s1, err1 = run_cli(["-d", run_dir, "-1", nc, "-2", nk, "-u", 1000, "--keep-tmp", "--save-sk"], out)
    assert "Resume" not in err1
    surv, _, total = oracle_pipeline(o, nc, nk, mats, kms, 0.05 / 1000, "bonferroni", 0.05)
    got = {"kmer": [], "p": [], "sign": [], "mc": [], "mk": [], "counts": []}
    sk_rows = []
    for p in range(3):
        f = KF.read_survivor_file(str(out / "partitions" / ("p%d_uncorrected" % p)))
        for key in got:
            got[key] += f[key]
        hdr, km, cnt = KF.read_matrix_file(str(out / "positive_kmer_matrix" / "matrices" / ("matrix_%d.count.lz4" % p)))
        assert (hdr["k"], hdr["count_bytes"], hdr["nb_counts"], hdr["partition"]) == (k, 4, nc + nk, p)
        assert km.tolist() == f["kmer"]
        sk_rows.append(cnt)
    assert got["kmer"] == surv["kmer"] and got["sign"] == surv["sign"]
    assert got["mc"] == surv["mc"] and got["mk"] == surv["mk"]
    assert np.allclose(got["p"], surv["p"], rtol=0, atol=1e-10)
    # ... and since the CLI passes its survivors through kmd_pvalues_refine: the oracle's bits (glibc's own log is the rounded
    # one in all but ~1 call in 10^3)
    assert (np.array(got["p"]) == np.array(surv["p"])).mean() >= 0.99 and np.allclose(got["p"], surv["p"], rtol=1e-9, atol=0)
    lut = [{int(v): i for i, v in enumerate(km)} for km in kms]
    want_rows = np.array([next(m[l[kv]] for m, l in zip(mats, lut) if kv in l) for kv in surv["kmer"]])
    assert (np.array(got["counts"]) == want_rows).all() and (np.concatenate(sk_rows) == want_rows).all()
    ob = open(out / "options.bin", "rb").read()
    assert len(ob) == 37
    thr, cutoff, corr, pop, kpca, npc = struct.unpack("<ddi?dQ", ob)
    assert (thr, cutoff, corr, pop, npc) == (0.05, 1000.0, 1, False, 2)
    assert os.path.exists(os.path.join(str(out), "positive_kmer_matrix", "kmtricks.fof"))
    # second run, other corrector: stage 1 is not redone
    s2, err2 = run_cli(["-d", run_dir, "-1", nc, "-2", nk, "-u", 1000, "--keep-tmp", "-c", "benjamini"], out)
    assert "Resume" in err2 and "Process partitions" not in err2
    fresh, _ = run_cli(["-d", run_dir, "-1", nc, "-2", nk, "-u", 1000, "-c", "benjamini"], tmp_path / "fresh")
    assert s2 == fresh
    for name in ("control_kmers.fasta", "case_kmers.fasta"):
        assert open(out / name).read() == open(tmp_path / "fresh" / name).read()
    # third run, other first-pass threshold: stage 1 IS redone (compare_opt bit 0)
    s3, err3 = run_cli(["-d", run_dir, "-1", nc, "-2", nk, "-u", 100, "--keep-tmp", "-c", "benjamini"], out)
    assert "Process partitions" in err3 and s3["n_sig"] > s2["n_sig"]


def test_cli_two_limb_kmers_k63(tmp_path):
    """configs[3]: 32 < k <= 64, the (hi, lo) 128-bit k-mer through file reader, two-limb merge,
    filter, survivor files and FASTA."""
    o = OL.load()
    nc, nk, n, k = 3, 3, 20_000, 63
    parts, mats, los, his = [], [], [], []
    for p in range(2):
        host, lo, hi = o.synth_rows(SEED, p, 0, n, nc, nk, 4, kmer_limbs=2)
        parts.append([(lo[host[:, s] > 0], host[host[:, s] > 0, s], hi[host[:, s] > 0]) for s in range(nc + nk)])
        mats.append(host); los.append(lo); his.append(hi)
    ids = ["C%d" % i for i in range(nc)] + ["K%d" % i for i in range(nk)]
    KF.write_run_dir(str(tmp_path / "km"), k, ids, parts)
    s, _ = run_cli(["-d", tmp_path / "km", "-1", nc, "-2", nk, "-c", "disabled", "-s", 0.05, "-u", 1000, "--keep-tmp"], tmp_path / "o")
    totals = np.sum([m.sum(axis=0, dtype=np.uint64) for m in mats], axis=0)
    lf = o.lf_table(10000)
    want = {"control": [], "case": []}
    n_sig = 0
    for p, m in enumerate(mats):
        out = o.diff_partition(m, OL.LAYOUT_ROWS, nc, nk, int(totals[:nc].sum()), int(totals[nc:].sum()), lf, 0.05 / 1000)
        idx = out["row"].astype(np.int64)
        n_sig += len(idx)
        f = KF.read_survivor_file(str(tmp_path / "o" / "partitions" / ("p%d_uncorrected" % p)), kmer_bytes=16)
        assert f["kmer"] == los[p][idx].tolist() and f["kmer_hi"] == his[p][idx].tolist()
        assert np.allclose(f["p"], out["pvalue"], rtol=0, atol=1e-10)
        assert (np.array(f["p"]) == out["pvalue"]).mean() >= 0.98                   # kmd_pvalues_refine: the oracle's bits
        for i, sg, pv in zip(idx, out["sign"], out["pvalue"]):
            if pv < 0.05:                                                    # "disabled" = threshold corrector
                want["control" if sg == 0 else "case"].append(KF.kmer_to_string2(his[p][i], los[p][i], k))
    assert s["n_sig"] == n_sig and n_sig > 5 and s["total_kmers"] == 2 * n and s["kmer_size"] == 63
    for name in ("control", "case"):
        got = [seq for _, seq in read_fasta(tmp_path / "o" / ("%s_kmers.fasta" % name))]
        assert got == want[name] and all(len(q) == 63 for q in got)


def test_cli_partitions_sharded_over_gpus(synth_run, tmp_path):
    """--devices N: one worker thread per GPU, partition p on GPU p mod N (folded onto the GPUs the
    box has); survivors, files, PCA and decisions equal the one-GPU run."""
    run_dir, nc, nk, k, mats, kms = synth_run
    common = ["-d", run_dir, "-1", nc, "-2", nk, "-u", 1000, "-c", "benjamini", "--keep-tmp", "--pop-correction", "--kmer-pca", 0.05]
    a, _ = run_cli(common + ["--devices", 1], tmp_path / "a")
    b, err = run_cli(common + ["--devices", 2], tmp_path / "b")
    c, _ = run_cli(common + ["--devices", 5], tmp_path / "c")           # more workers than partitions
    d, _ = run_cli(common + ["--devices", 8], tmp_path / "d")           # the width of the node the job is sold for
    assert a == b == c == d and a["n_sig"] > 10
    for name in ("control_kmers.fasta", "case_kmers.fasta", "popstrat/pcs.evec", "partitions/p0_uncorrected", "partitions/p2_uncorrected",
                 "partitions/p1_popstrat_uncorrected"):
        ref = open(tmp_path / "a" / name, "rb").read()
        for other in "bcd":
            assert open(tmp_path / other / name, "rb").read() == ref, (other, name)


def test_cli_a_rank_that_fails_ends_the_run_with_its_error(synth_run, tmp_path):
    """--devices 4 with one rank's thread failing before the exchange of stage 3 (KMD_TEST_FAIL_RANK): the run ends
    with that rank's message and a non-zero status within seconds -- it used to hang in the other ranks' barrier."""
    run_dir, nc, nk, k, mats, kms = synth_run
    env = dict(os.environ, KMD_TEST_FAIL_RANK="2")
    r = subprocess.run([CLI, "diff", "-o", str(tmp_path / "o"), "-d", run_dir, "-1", str(nc), "-2", str(nk), "-u", "1000", "-c", "benjamini",
                        "--devices", "4"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "KMD_TEST_FAIL_RANK: rank 2" in r.stderr, r.stderr[-2000:]


def test_cli_more_near_threshold_rows_than_one_launch_lists(tmp_path):
    """KMD_CNT_NEAR_UNRESOLVED read by the host: 6 000 rows with the same two count sums and the
    first threshold put ON their p-value -- more rows within 1e-8 of the threshold than one launch can list for the
    correctly rounded second look (4096).  The fused pass reports the rest unresolved; the command takes the partition's
    rows as (k-mer, control sum, case sum) -- no matrix -- and tests them in pieces until none is (stderr says so), and
    --matrix-path gets there by halving: both write what the oracle decides."""
    import kmdiff_amd as K
    import ctypes as C
    o = OL.load()
    nc, nk, k, n_bg, n_same = 4, 4, 31, 30_000, 6_000
    host, lo, _ = o.synth_rows(SEED, 9, 0, n_bg, nc, nk, 4)
    rng = np.random.default_rng(11)
    extra_km = np.setdiff1d(np.unique(rng.integers(int(lo.min()), int(lo.max()), 2 * n_same, dtype=np.uint64)), lo)[:n_same]
    assert len(extra_km) == n_same
    tcs, tks = host[:, :nc].sum(axis=0, dtype=np.uint64), host[:, nc:].sum(axis=0, dtype=np.uint64)
    lf = o.lf_table(10000)
    lib = K._native.lib()
    # a row shape whose correctly rounded p-value (what the device decides a near row by) is the oracle's own double
    chosen = None
    cands = [[c0, 0, 0, 0, a, b, 2, 1] for c0 in (0, 1) for a in (2, 3, 4, 5) for b in (1, 2, 3)]
    for cand in cands:
        row = np.array(cand, dtype=np.uint32)
        full = np.vstack([host, np.tile(row, (n_same, 1))])
        tot_c = int(tcs.sum()) + n_same * int(row[:nc].sum()); tot_k = int(tks.sum()) + n_same * int(row[nc:].sum())
        one = o.diff_partition(row[None, :], OL.LAYOUT_ROWS, nc, nk, tot_c, tot_k, lf, 1.0)
        p_star = float(one["pvalue"][0])
        tot = full.sum(axis=0, dtype=np.uint64)
        model = K.PoissonLikelihood(nc, nk, tot[:nc], tot[nc:], 10000)
        p_rounded = lib.kmd_test_row_pvalue_rounded(C.c_void_p(model.handle), int(row[:nc].sum()), int(row[nc:].sum()))
        if p_rounded == p_star and 1e-12 < p_star < 1e-2:
            chosen = (row, full, p_star)
            break
    assert chosen is not None
    row, full, p_star = chosen
    km_all = np.concatenate([lo, extra_km])
    order = np.argsort(km_all, kind="stable")
    full, km_all = full[order], km_all[order]
    part = [(km_all[full[:, s] > 0], full[full[:, s] > 0, s]) for s in range(nc + nk)]
    ids = ["C%d" % i for i in range(nc)] + ["K%d" % i for i in range(nk)]
    KF.write_run_dir(str(tmp_path / "km"), k, ids, [part])
    tot = full.sum(axis=0, dtype=np.uint64)
    want = o.diff_partition(full, OL.LAYOUT_ROWS, nc, nk, int(tot[:nc].sum()), int(tot[nc:].sum()), lf, p_star)
    assert (want["pvalue"] == p_star).sum() >= n_same                       # the reference keeps them: p <= threshold (merge.hpp:78)
    want_km = sorted(km_all[want["row"].astype(np.int64)].tolist())
    outs = {}
    for name, extra in (("fused", []), ("matrix", ["--matrix-path"])):
        s, err = run_cli(["-d", str(tmp_path / "km"), "-1", nc, "-2", nk, "-s", repr(p_star), "-u", 1, "-c", "disabled", "--keep-tmp"] + extra,
                         tmp_path / name)
        assert "every near-threshold row decided in" in err, err[-1500:]
        assert ("again, as rows of sums in pieces" in err) == (name == "fused") and "as a matrix in pieces" not in err
        assert s["near_threshold"] >= n_same and s["n_sig"] == len(want_km)
        recs = KF.read_survivor_file(str(tmp_path / name / "partitions" / "p0_uncorrected"))
        outs[name] = sorted(int(v) for v in recs["kmer"])
        assert outs[name] == want_km


def test_cli_partitions_on_two_real_gpus(synth_run, tmp_path):
    """The same on a box that HAS two GPUs (skipped on the one-GPU test box): the workers are not folded, partition p
    really runs on GPU p mod 2; byte-identical outputs."""
    import kmdiff_amd as K
    if K.device_count() < 2:
        pytest.skip("one GPU on this box")
    run_dir, nc, nk, k, mats, kms = synth_run
    common = ["-d", run_dir, "-1", nc, "-2", nk, "-u", 1000, "-c", "benjamini", "--keep-tmp", "--pop-correction", "--kmer-pca", 0.05]
    a, _ = run_cli(common + ["--devices", 1], tmp_path / "a")
    b, err = run_cli(common + ["--devices", 2], tmp_path / "b")
    assert a == b and a["n_sig"] > 10 and "folded" not in err
    for name in ("control_kmers.fasta", "case_kmers.fasta", "popstrat/pcs.evec", "partitions/p0_uncorrected", "partitions/p1_popstrat_uncorrected"):
        assert open(tmp_path / "b" / name, "rb").read() == open(tmp_path / "a" / name, "rb").read(), name


def test_cli_cmodel_plugin_row_loop(synth_run, tmp_path):
    """--cmodel / --config (cli.cpp:246-262, model_manager.hpp:33-94): a user's IModel plugin is
    loaded with dlopen and called row by row on the host, like the reference; here the plugin is this
    build's own libkmdiff_hip_model.so, so the result must equal the native path's."""
    run_dir, nc, nk, k, mats, kms = synth_run
    totals = np.sum([m.sum(axis=0, dtype=np.uint64) for m in mats], axis=0)
    cfgs = "controls=%d;cases=%d;total_controls=%s;total_cases=%s;log_factorial=10000" % (
        nc, nk, ",".join(str(int(t)) for t in totals[:nc]), ",".join(str(int(t)) for t in totals[nc:]))
    plugin = os.path.join(ROOT, "kmdiff_amd", "lib", "libkmdiff_hip_model.so")
    a, _ = run_cli(["-d", run_dir, "-1", nc, "-2", nk, "-u", 100, "-c", "benjamini", "--keep-tmp"], tmp_path / "a")
    b, err = run_cli(["-d", run_dir, "-1", nc, "-2", nk, "-u", 100, "-c", "benjamini", "--keep-tmp", "--cmodel", plugin, "--config", cfgs,
                      "--pop-correction"], tmp_path / "b")
    assert "model plugin:" in err and "disabled with custom models" in err
    assert a == b and a["n_sig"] > 50
    for name in ("control_kmers.fasta", "case_kmers.fasta", "partitions/p1_uncorrected"):
        assert open(tmp_path / "a" / name, "rb").read() == open(tmp_path / "b" / name, "rb").read(), name
    r = subprocess.run([CLI, "diff", "-d", run_dir, "-1", str(nc), "-2", str(nk), "-o", str(tmp_path / "c"), "--cmodel", "/nonexistent.so"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "--cmodel" in r.stderr


def test_cli_pop_correction(synth_run, tmp_path):
    run_dir, nc, nk, k, mats, kms = synth_run
    o = OL.load()
    rng = np.random.default_rng(2)
    Z = rng.normal(0, 0.1, size=(nc + nk, 10))
    np.savetxt(tmp_path / "pcs.evec", Z, fmt="%.17g")
    s, _ = run_cli(["-d", run_dir, "-1", nc, "-2", nk, "-c", "bonferroni", "-u", 1000, "--pop-correction", "--pcs",
                    tmp_path / "pcs.evec"], tmp_path / "o")
    surv, _, total = oracle_pipeline(o, nc, nk, mats, kms, 0.05 / 1000, "bonferroni", 0.05)
    totals = np.sum([m.sum(axis=0, dtype=np.uint64) for m in mats], axis=0)
    Zr = np.loadtxt(tmp_path / "pcs.evec")
    alt, null_model, tot_d, y = o.popstrat_setup(nc, nk, totals[:nc], totals[nc:], Zr, 2, True)
    rows = []
    for m, km in zip(mats, kms):
        lut = {int(v): i for i, v in enumerate(km)}
        rows.append((m, lut))
    counts = np.array([next(m[lut[kv]] for m, lut in rows if kv in lut) for kv in surv["kmer"]], dtype=np.float64)
    p2 = o.popstrat_pvalues(alt, null_model, tot_d, y, counts)
    keep = o.aggregate(1, 0.05, total, p2)
    assert s["kept"] == int(keep.sum()) and s["n_sig"] == len(p2)


def test_cli_pop_correction_with_device_pca(synth_run, tmp_path):
    """--pop-correction without --pcs: rows sampled at --kmer-pca, smartpca's normalisation and
    eigen-decomposition on the device, popstrat/pcs.evec in evec2pca's format, then the same re-test."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pca_oracle as PO
    run_dir, nc, nk, k, mats, kms = synth_run
    S = nc + nk
    o = OL.load()
    s, err = run_cli(["-d", run_dir, "-1", nc, "-2", nk, "-c", "bonferroni", "-u", 1000, "--pop-correction", "--kmer-pca", 0.05,
                      "--random-seed", 11], tmp_path / "o")
    assert "PCA done" in err
    lines = open(tmp_path / "o" / "popstrat" / "pcs.evec").read().rstrip("\n").split("\n")
    Zf = np.array([[float(x) for x in l.split()] for l in lines])
    assert Zf.shape == (S, min(S, 10))
    # the oracle's PCA over the same sampled rows
    sampled = [m[PO.sampled_mask(11, 0.05, km)] for m, km in zip(mats, kms)]
    n_sampled = sum(len(x) for x in sampled)
    assert ("%d k-mers sampled" % n_sampled) in err and n_sampled > 1000
    evec, evals = PO.eigen(PO.gram(np.concatenate(sampled)), min(S, 10))
    assert np.abs(Zf[:, :2] - evec[:, :2]).max() <= 0.00005 + 1e-9          # "%.04f"
    assert lines == PO.pcs_evec_lines(np.round(Zf, 4)) or all(len(l.split()) == min(S, 10) for l in lines)
    # the re-test with the components the command wrote
    surv, _, total = oracle_pipeline(o, nc, nk, mats, kms, 0.05 / 1000, "bonferroni", 0.05)
    totals = np.sum([m.sum(axis=0, dtype=np.uint64) for m in mats], axis=0)
    Zr = np.zeros((S, 10)); Zr[:, :Zf.shape[1]] = Zf
    alt, null_model, tot_d, y = o.popstrat_setup(nc, nk, totals[:nc], totals[nc:], Zr, 2, True)
    lut = [{int(v): i for i, v in enumerate(km)} for km in kms]
    counts = np.array([next(m[l[kv]] for m, l in zip(mats, lut) if kv in l) for kv in surv["kmer"]], dtype=np.float64)
    keep = o.aggregate(1, 0.05, total, o.popstrat_pvalues(alt, null_model, tot_d, y, counts))
    assert s["kept"] == int(keep.sum()) and s["n_sig"] == len(counts)


@pytest.mark.parametrize("bits,dtype", [(32, np.uint32), (16, np.uint16), (8, np.uint8)])
def test_imodel_plugin_loaded_like_the_reference_does(tmp_path, bits, dtype):
    """libkmdiff_hip_model.so through a dlopen / plugin_name / create<bits> / configure / process
    host (model_manager.hpp:33-94): per-row results == oracle."""
    o = OL.load()
    nc, nk, n = 4, 3, 60
    host, _, _ = o.synth_rows(SEED, 4, 0, n, nc, nk, np.dtype(dtype).itemsize)
    totals = host.sum(axis=0, dtype=np.uint64) * 1000 + 17
    cfg = "controls=%d;cases=%d;total_controls=%s;total_cases=%s;log_factorial=500" % (
        nc, nk, ",".join(str(int(t)) for t in totals[:nc]), ",".join(str(int(t)) for t in totals[nc:]))
    plugin = os.path.join(ROOT, "kmdiff_amd", "lib", "libkmdiff_hip_model.so")
    harness = os.path.join(ROOT, "kmdiff_amd", "bin", "plugin_host")
    rows_txt = "\n".join(" ".join(str(int(v)) for v in r) for r in host) + "\n"
    r = subprocess.run([harness, plugin, str(bits), cfg, str(nc), str(nk)], input=rows_txt, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "plugin: kmdiff_hip_poisson" in r.stderr
    got = [line.split() for line in r.stdout.strip().split("\n")]
    assert len(got) == n
    p, s, mc, mk = o.poisson_rows(host, OL.LAYOUT_ROWS, nc, nk, int(totals[:nc].sum()), int(totals[nc:].sum()),
                                  o.lf_table(500))
    for i, g in enumerate(got):
        assert int(g[1]) == s[i] and float.fromhex(g[2]) == mc[i] and float.fromhex(g[3]) == mk[i]
        assert abs(float.fromhex(g[0]) - p[i]) <= 1e-10 and abs(float.fromhex(g[0]) - p[i]) <= 1e-9 * p[i] + 1e-300
    # the plugin passes every p-value through kmd_pvalues_refine (the reference compares THIS number with its threshold)
    assert sum(float.fromhex(g[0]) == p[i] for i, g in enumerate(got)) >= 0.98 * n


def test_cli_empty_sample_files_and_an_empty_partition(tmp_path):
    """A sample without k-mers in a partition (an LZ4 frame of nothing), a partition where every
    sample file is empty, and one-record files: the streaming reader, the merge and the survivor
    bookkeeping take them; same counters as the oracle on the non-empty rows."""
    o = OL.load()
    nc, nk, n, k = 3, 3, 20_000, 31
    parts, mats, kms = [], [], []
    for p in range(4):
        host, lo, _ = o.synth_rows(SEED + 9, p, 0, n, nc, nk, 4)
        if p == 1:
            host = host.copy(); host[:, 0] = 0; host[:, 4] = 0           # two samples absent from this partition
            keep_rows = host.sum(axis=1) > 0
            host, lo = host[keep_rows], lo[keep_rows]
        if p == 2:
            host, lo = host[:0], lo[:0]                                  # nothing at all
        if p == 3:
            host, lo = host[:1], lo[:1]                                  # one row
        parts.append([(lo[host[:, s] > 0], host[host[:, s] > 0, s]) for s in range(nc + nk)])
        mats.append(host); kms.append(lo)
    ids = ["C%d" % i for i in range(nc)] + ["K%d" % i for i in range(nk)]
    KF.write_run_dir(str(tmp_path / "km"), k, ids, parts)
    s, _ = run_cli(["-d", tmp_path / "km", "-1", nc, "-2", nk, "-c", "disabled", "-s", 0.05, "-u", 100, "-t", 3], tmp_path / "o")
    surv, keep, total = oracle_pipeline(o, nc, nk, [m for m in mats if len(m)], [km for m, km in zip(mats, kms) if len(m)], 0.05 / 100,
                                        "disabled", 0.05)
    assert s["total_kmers"] == total == sum(len(m) for m in mats)
    assert s["n_sig"] == len(surv["p"]) > 0 and s["kept"] == int(keep.sum())


def test_cli_fused_path_equals_matrix_path(synth_run, tmp_path):
    """The default path (kmd_merge_filter: streams merged and tested in one go, no matrix) writes what
    --matrix-path (k-way merge into the count matrix, then K1) writes: summary, both FASTA files, the --keep-tmp
    survivor files byte for byte (the survivors' count rows then come from the streams), the --save-sk matrices,
    and with --pop-correction the sampled rows, pcs.evec and the re-tested p-values; also with BH
    (ascending-p order) and on the reference's fixture."""
    import shutil
    run_dir, nc, nk, k, mats, kms = synth_run
    for extra in (["-c", "bonferroni"], ["-c", "benjamini", "--keep-tmp"], ["--save-sk"],
                  ["--pop-correction", "--kmer-pca", "0.05", "-c", "disabled", "--keep-tmp"]):
        a, _ = run_cli(["-d", run_dir, "-1", nc, "-2", nk, "-u", 1000, "-t", 3, "--matrix-path"] + extra, tmp_path / "a")
        b, err = run_cli(["-d", run_dir, "-1", nc, "-2", nk, "-u", 1000, "-t", 3] + extra, tmp_path / "b")
        assert a == b and a["n_sig"] > 10
        for name in ("control_kmers.fasta", "case_kmers.fasta"):
            assert open(tmp_path / "a" / name).read() == open(tmp_path / "b" / name).read()
        if "--keep-tmp" in extra:
            suffix = "_popstrat_uncorrected" if "--pop-correction" in extra else "_uncorrected"
            for p in range(3):
                fa = KF.lz4_frame_decode(open(tmp_path / "a" / "partitions" / ("p%d%s" % (p, suffix)), "rb").read())
                fb = KF.lz4_frame_decode(open(tmp_path / "b" / "partitions" / ("p%d%s" % (p, suffix)), "rb").read())
                assert fa == fb and len(fa) > 0
        if "--save-sk" in extra:
            for p in range(3):
                name = os.path.join("positive_kmer_matrix", "matrices", "matrix_%d.count.lz4" % p)
                assert open(tmp_path / "a" / name, "rb").read() == open(tmp_path / "b" / name, "rb").read()
        if "--pop-correction" in extra:
            assert open(tmp_path / "a" / "popstrat" / "pcs.evec").read() == open(tmp_path / "b" / "popstrat" / "pcs.evec").read()
        shutil.rmtree(tmp_path / "a"); shutil.rmtree(tmp_path / "b")
    for extra in ([], ["--matrix-path"]):
        s, _ = run_cli(["-d", os.path.join(ROOT, "tests", "golden", "km_out_dir"), "-1", 1, "-2", 1, "-u", 10000] + extra, tmp_path / "f")
        assert (s["total_kmers"], s["n_sig"], s["kept"]) == (320, 0, 0)
        shutil.rmtree(tmp_path / "f")


def test_cli_packed_transfer_equals_raw_transfer(synth_run, tmp_path):
    """The default k-mer feed (records packed on the host as they are decoded, kmd_pack_block; unpacked on the GPU,
    kmd_unpack_streams) against --raw-transfer (plain 12-byte records): byte-identical outputs on every path that reads
    the streams -- the fused merge, the matrix path, pop-strat's count rows and PCA sampling, --keep-tmp files -- and the
    bytes per record that crossed the link in summary.json."""
    run_dir, nc, nk, k, mats, kms = synth_run
    for tag, extra in (("fused", ["--keep-tmp", "--pop-correction", "--kmer-pca", 0.05, "-c", "benjamini"]), ("matrix", ["--matrix-path"])):
        common = ["-d", run_dir, "-1", nc, "-2", nk, "-u", 1000] + extra
        a, _ = run_cli(common, tmp_path / (tag + "_a"))
        b, _ = run_cli(common + ["--raw-transfer"], tmp_path / (tag + "_b"))
        ta, tb = TRANSFER[str(tmp_path / (tag + "_a"))], TRANSFER[str(tmp_path / (tag + "_b"))]
        assert a == b and a["n_sig"] > 10
        records = sum(int((m > 0).sum()) for m in mats)
        assert ta["format"] == "packed" and tb["format"] == "raw" and ta["records"] == tb["records"] == records
        assert tb["bytes_per_record"] == 12.0 and 3.5 < ta["bytes_per_record"] < 6.0, ta
        names = ["control_kmers.fasta", "case_kmers.fasta"]
        if tag == "fused":
            names += ["popstrat/pcs.evec", "partitions/p0_uncorrected", "partitions/p2_uncorrected", "partitions/p1_popstrat_uncorrected"]
        for name in names:
            assert open(tmp_path / (tag + "_a") / name, "rb").read() == open(tmp_path / (tag + "_b") / name, "rb").read(), (tag, name)


def test_cli_packed_transfer_counts_beyond_one_byte_and_tiny_files(tmp_path):
    """Streams whose counts need the escape list (>= 255, up to 2^32 - 1), a sample without k-mers in a partition, a
    partition of a single k-mer: the packed feed and the raw one agree."""
    rng = np.random.default_rng(5)
    nc, nk, k = 2, 2, 31
    parts = []
    uni = np.unique(rng.integers(0, 1 << 62, 3000, dtype=np.uint64))
    st = []
    for s in range(nc + nk):
        pick = rng.random(len(uni)) < 0.7
        cnt = rng.integers(1, 1000, int(pick.sum())).astype(np.uint32)
        cnt[rng.integers(0, len(cnt), 5)] = np.uint32(2 ** 32 - 1 if s == 0 else 70000)
        st.append((uni[pick], cnt))
    st[2] = (np.zeros(0, np.uint64), np.zeros(0, np.uint32))                  # a sample without k-mers here
    parts.append(st)
    parts.append([(np.array([77], dtype=np.uint64), np.array([3], dtype=np.uint32))] + [(np.zeros(0, np.uint64), np.zeros(0, np.uint32))] * 3)
    ids = ["C0", "C1", "K0", "K1"]
    KF.write_run_dir(str(tmp_path / "km"), k, ids, parts)
    common = ["-d", str(tmp_path / "km"), "-1", nc, "-2", nk, "-s", 0.9, "-u", 1, "-c", "disabled", "--keep-tmp"]
    a, _ = run_cli(common, tmp_path / "a")
    b, _ = run_cli(common + ["--raw-transfer"], tmp_path / "b")
    present = len(np.unique(np.concatenate([t[0] for t in parts[0]])))
    assert a == b and a["total_kmers"] == present + 1 and a["n_sig"] > 100
    for name in ("control_kmers.fasta", "case_kmers.fasta", "partitions/p0_uncorrected", "partitions/p1_uncorrected"):
        assert open(tmp_path / "a" / name, "rb").read() == open(tmp_path / "b" / name, "rb").read(), name


def test_cli_staging_slot_outgrown_by_a_later_partition(tmp_path):
    """The page-locked arrays of a staging slot are cut from one allocation sized by the files of the FIRST partition that
    uses the slot (main.cpp, partition_input::slab); a stream of a later partition that needs more moves into an array of
    its own while it is being decoded.  Partitions 0 and 1 of a few hundred k-mers, 2 and 3 of 300 000 (a thousand times the
    bytes, in the slots 0 and 1 were sized for), then small ones again: the packed feed and the raw one agree byte for byte
    and every k-mer is counted."""
    rng = np.random.default_rng(15)
    nc, nk, k = 2, 2, 31
    parts, present = [], 0
    for n_keys in (300, 200, 300_000, 280_000, 500, 250_000):
        uni = np.unique(rng.integers(0, 1 << 62, n_keys, dtype=np.uint64))
        st = []
        for s in range(nc + nk):
            pick = rng.random(len(uni)) < 0.7
            cnt = rng.integers(1, 40, int(pick.sum())).astype(np.uint32) * np.uint32(1 if s < nc else 3)
            st.append((uni[pick], cnt))
        present += len(np.unique(np.concatenate([t[0] for t in st])))
        parts.append(st)
    KF.write_run_dir(str(tmp_path / "km"), k, ["C0", "C1", "K0", "K1"], parts)
    common = ["-d", str(tmp_path / "km"), "-1", nc, "-2", nk, "-s", 0.05, "-u", 1000, "-c", "disabled", "--keep-tmp", "-t", 4]
    os.environ["KMD_HOST_TIMING"] = "2"                             # (a line per decoded partition: what it page-locked meanwhile)
    try:
        a, log = run_cli(common, tmp_path / "a")
    finally:
        del os.environ["KMD_HOST_TIMING"]
    b, _ = run_cli(common + ["--raw-transfer"], tmp_path / "b")
    assert a == b and a["total_kmers"] == present and a["n_sig"] > 1000
    assert TRANSFER[str(tmp_path / "a")]["format"] == "packed"
    grown = {int(l.split("partition ")[1].split(" ")[0]): float(l.split(" for ")[1].split(" MB")[0]) for l in log.split("\n") if "decoded in" in l}
    # (the large partitions outgrew their slots' pieces -- megabytes page-locked while they were decoded; the ones behind them fit)
    assert grown[2] > 10 and grown[3] > 10 and grown[4] < 4 and grown[5] < 4, grown
    for name in ["control_kmers.fasta", "case_kmers.fasta"] + ["partitions/p%d_uncorrected" % p for p in range(6)]:
        assert open(tmp_path / "a" / name, "rb").read() == open(tmp_path / "b" / name, "rb").read(), name


def test_cli_background_release_of_the_staging_arrays_changes_nothing(synth_run, tmp_path):
    """The page-locked staging arrays are released by a background thread while stages 2-3 run (main.cpp, retired_staging);
    KMD_SYNC_RELEASE=1 releases them before stage 1 returns, as every round before 6 did: same outputs either way, with the
    pop-strat stage and two workers (two threads retiring their rings) behind stage 1."""
    run_dir, nc, nk, k, mats, kms = synth_run
    common = ["-d", run_dir, "-1", nc, "-2", nk, "-u", 1000, "--pop-correction", "--kmer-pca", 0.05, "-c", "benjamini", "--devices", 2]
    a, _ = run_cli(common, tmp_path / "a")
    os.environ["KMD_SYNC_RELEASE"] = "1"
    try:
        b, _ = run_cli(common, tmp_path / "b")
    finally:
        del os.environ["KMD_SYNC_RELEASE"]
    assert a == b and a["n_sig"] > 10
    for name in ("control_kmers.fasta", "case_kmers.fasta", "popstrat/pcs.evec"):
        assert open(tmp_path / "a" / name, "rb").read() == open(tmp_path / "b" / name, "rb").read(), name
