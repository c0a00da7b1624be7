#!/usr/bin/env python3
"""GPU box: `kmdiff-hip diff` on random kmtricks run directories against the oracle's pipeline (per-partition merge +
PoissonLikelihood::process + threshold, then the corrector), beyond the fixed-seed cases of tests/test_gpu_cli.py:
1..12 samples a side, 1..6 partitions of 0..20 000 k-mers (some samples or whole partitions empty), k = 11..64 (one and two limbs), counts to
70 000, all five corrections, thresholds, -t, --devices 1|2, packed or --raw-transfer, fused or --matrix-path.
Held: summary counts, and both FASTA files record by record (k-mer, rank, means; p as printed, 6 digits).
usage: python3 tests/soak_cli.py [--seconds 300] [--seed N]"""
import argparse
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kmtricks_files as KF                                   # noqa: E402
import oracle_lib as OL                                       # noqa: E402
from test_gpu_cli import run_cli, read_fasta, fmt_shortest, oracle_pipeline     # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=300)
ap.add_argument("--seed", type=int, default=int(time.time()))
ap.add_argument("--cases", type=int, default=0, help="stop after this many runs (0: by --seconds alone): the same runs on every box")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
o = OL.load()
print("cli soak seed", a.seed, flush=True)
t0, n_runs, n_kept = time.time(), 0, 0
tmp = tempfile.mkdtemp(prefix="kmd_soak_")
try:
    while time.time() - t0 < a.seconds and (a.cases == 0 or n_runs < a.cases):
        nc, nk = int(rng.integers(1, 13)), int(rng.integers(1, 13))
        S = nc + nk
        k = int(rng.choice([11, 15, 20, 27, 31, 32, 33, 47, 63, 64]))
        two = k > 32
        n_parts = int(rng.integers(1, 7))
        count_hi = int(rng.choice([3, 40, 255, 256, 1000, 70000]))
        effect = float(rng.choice([1.0, 2.0, 6.0]))
        parts, mats, kms, his = [], [], [], []
        for p in range(n_parts):
            n = 0 if rng.random() < 0.1 else int(rng.integers(1, 20001))
            lo = np.unique(rng.integers(0, 1 << (2 * k), n, dtype=np.uint64)) if 2 * k < 64 else np.unique(rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64))
            n = len(lo)
            hi = None
            if two:                                           # the high limb holds the 2 (k - 32) bits above the low 64; sorted (hi, lo)
                top = 1 << (2 * (k - 32))
                hi = np.sort(rng.integers(0, min(top, int(rng.choice([2, 7, top]))), n, dtype=np.uint64) if top < (1 << 63) else rng.integers(0, 1 << 63, n, dtype=np.uint64) * np.uint64(2))
                order = np.lexsort((lo, hi))
                lo, hi = lo[order], hi[order]
            host = rng.integers(1, count_hi + 1, (n, S)).astype(np.uint32)
            if effect > 1.0:
                up = rng.random(n) < 0.02
                host[up, nc:] = np.minimum(host[up, nc:].astype(np.float64) * effect, 4e9).astype(np.uint32)
            host[rng.random((n, S)) < rng.uniform(0, 0.9)] = 0
            for s in rng.choice(S, size=int(rng.integers(0, max(1, S // 3))), replace=False):
                if rng.random() < 0.3:
                    host[:, s] = 0                            # a sample without k-mers in this partition
            keep = host.sum(axis=1) > 0                       # a k-mer nobody holds is in no file
            host, lo = host[keep], lo[keep]
            if two:
                hi = hi[keep]
                parts.append([(lo[host[:, s] > 0], host[host[:, s] > 0, s], hi[host[:, s] > 0]) for s in range(S)])
            else:
                parts.append([(lo[host[:, s] > 0], host[host[:, s] > 0, s]) for s in range(S)])
            mats.append(host)
            kms.append(lo)
            his.append(hi)
        totals = np.sum([m.sum(axis=0, dtype=np.uint64) for m in mats], axis=0) if mats else np.zeros(S, np.uint64)
        if int(totals[:nc].sum()) == 0 or int(totals[nc:].sum()) == 0:
            continue
        run = os.path.join(tmp, "run%d" % n_runs)
        ids = ["C%d" % i for i in range(nc)] + ["K%d" % i for i in range(nk)]
        KF.write_run_dir(run, k, ids, parts)
        correction = str(rng.choice(["bonferroni", "benjamini", "sidak", "holm", "disabled"]))
        alpha, cutoff = float(rng.choice([0.05, 0.5, 1e-3])), int(rng.choice([1, 100, 100000]))
        args = ["-d", run, "-1", nc, "-2", nk, "-c", correction, "-s", alpha, "-u", cutoff, "-t", int(rng.choice([1, 3, 16]))]
        if rng.random() < 0.3:
            args += ["--devices", 2]
        if rng.random() < 0.3:
            args += ["--raw-transfer"]
        if rng.random() < 0.25:
            args += ["--matrix-path"]
        out = os.path.join(tmp, "out%d" % n_runs)
        s, _ = run_cli(args, out)
        # (oracle_pipeline carries one limb per survivor: give it the row's index in the job, look both limbs up afterwards)
        base_of = np.cumsum([0] + [len(x) for x in kms])
        surv, keep, total = oracle_pipeline(o, nc, nk, mats, [np.arange(base_of[i], base_of[i + 1], dtype=np.uint64) for i in range(len(kms))], alpha / cutoff, correction, alpha)
        all_lo = np.concatenate(kms) if kms else np.zeros(0, np.uint64)
        all_hi = np.concatenate(his) if two else None
        tag = (a.seed, n_runs, args)
        assert s["total_kmers"] == total and s["n_sig"] == len(surv["p"]) and s["kept"] == int(keep.sum()), (tag, s, total, len(surv["p"]), int(keep.sum()))
        order = list(range(len(keep)))
        if correction in ("benjamini", "holm"):                 # sorted_aggregator (aggregator.hpp:359-361): output in p order
            order.sort(key=lambda i: surv["p"][i])
        want = {"control": [], "case": []}
        for i in order:
            if keep[i]:
                want["control" if surv["sign"][i] == 0 else "case"].append(i)
        for name in ("control", "case"):
            got = read_fasta(os.path.join(out, "%s_kmers.fasta" % name))
            assert len(got) == len(want[name]), (tag, name, len(got), len(want[name]))
            for j, (i, (hdr, seq)) in enumerate(zip(want[name], got)):
                at = int(surv["kmer"][i])
                assert seq == (KF.kmer_to_string2(all_hi[at], all_lo[at], k) if two else KF.kmer_to_string(all_lo[at], k)), (tag, name, j)
                f = hdr[1:].split("_")
                assert f[0] == str(j) and f[2] == "control=%d" % int(surv["mc"][i]) and f[3] == "case=" + fmt_shortest(surv["mk"][i]), (tag, hdr)
                pv = float(f[1].split("=")[1])
                assert abs(pv - surv["p"][i]) <= 1e-5 * surv["p"][i] + 1e-300, (tag, hdr, surv["p"][i])
        n_kept += int(keep.sum())
        n_runs += 1
        shutil.rmtree(run, ignore_errors=True)
        shutil.rmtree(out, ignore_errors=True)
        if n_runs % 20 == 0:
            print("  %.0f s: %d runs, %d k-mers kept in all" % (time.time() - t0, n_runs, n_kept), flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
print("cli soak ok: %d runs, %d kept k-mers compared, %.0f s (seed %d)" % (n_runs, n_kept, time.time() - t0, a.seed))
