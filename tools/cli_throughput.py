#!/usr/bin/env python3
"""End-to-end `kmdiff-hip diff` on a fabricated kmtricks run directory (host side included: LZ4
decode of the per-sample files, H2D copies): how the command scales with -t and --devices.
usage: python3 tools/cli_throughput.py [--rows 1000000] [--parts 4] [--nc 20 --nk 20]"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kmdiff_amd as K          # noqa: E402
import kmtricks_files as KF     # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1_000_000)
ap.add_argument("--parts", type=int, default=4)
ap.add_argument("--nc", type=int, default=20)
ap.add_argument("--nk", type=int, default=20)
ap.add_argument("--keep", default="", help="write the run directory here, keep it, and stop")
ap.add_argument("--matrix", action="store_true", help="also write <run>/matrices: the pre-merged feed (matrix_proxy)")
ap.add_argument("--cpu-baseline", action="store_true", help="also: the CPU doing the same job on the same run directory (oracle/cpu_pipeline: liblz4 decode + the oracle's merge + the oracle's test, one task per partition on as many threads as the quota gives)")
ap.add_argument("--only", default="", help="instead of the -t sweep: this one set of flags, e.g. \"-t 16\"")
ap.add_argument("--ab-preload", default="", help="with --ab N: instead of packed against --raw-transfer, the shipped library against THIS libkmdiff_hip.so "
                "(LD_PRELOAD: an older build of the host packer, say), alternating, at -t 16")
ap.add_argument("--json", default="", help="write what was measured to this file as JSON (bench.py --e2e reads it)")
ap.add_argument("--ab", type=int, default=0, help="instead of the -t sweep: this many rounds of packed transfer against --raw-transfer at -t 16 and -t 64, alternating")
a = ap.parse_args()
S = a.nc + a.nk
root = tempfile.mkdtemp(prefix="kmrun_")
if a.keep:
    os.makedirs(a.keep, exist_ok=True)
    root = a.keep
t0 = time.time()
parts, records, merged = [], 0, []
for p in range(a.parts):
    mat = K.synth_matrix(0x6B6D64696666, p, a.rows, a.nc, a.nk, 4, K.LAYOUT_ROWS)
    host, lo = mat.to_host(), mat.kmers_to_host()[0]
    streams = []
    for s in range(S):
        sel = host[:, s] > 0
        streams.append((lo[sel], host[sel, s]))
        records += int(sel.sum())
    parts.append(streams)
    if a.matrix:
        merged.append((lo, host))
ids = ["C%d" % i for i in range(a.nc)] + ["K%d" % i for i in range(a.nk)]
KF.write_run_dir(os.path.join(root, "km"), 31, ids, parts)
for p, (lo, host) in enumerate(merged):
    KF.write_matrix_file(os.path.join(root, "km", "matrices", "matrix_%d.count.lz4" % p), 31, p, lo, host)
size = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(root) for f in fs)
print("run dir: %d partitions x %d rows, %d samples, %d records, %.1f MB on disk, written in %.0f s"
      % (a.parts, a.rows, S, records, size / 1e6, time.time() - t0), flush=True)
if a.keep:
    sys.exit(0)
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/proc/loadavg"):
    if os.path.exists(f):
        print(f, open(f).read().strip(), flush=True)
print("cpus in affinity mask:", len(os.sched_getaffinity(0)), flush=True)
cli = os.path.join(ROOT, "kmdiff_amd", "bin", "kmdiff-hip")


def cpu_quota():
    """threads the container may really run at once: the affinity mask cut by the cgroup's CPU quota"""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    return n


cpu_line = None
if a.cpu_baseline:
    import json
    exe = os.path.join(ROOT, "oracle", "cpu_pipeline")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "cpu_pipeline"])
    T = cpu_quota()
    r = subprocess.run([exe, os.path.join(root, "km"), str(a.nc), str(a.nk), str(T)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    cpu_line = json.loads(r.stdout)
    assert cpu_line["rows"] == a.parts * a.rows, cpu_line
    print("CPU baseline (oracle/cpu_pipeline, %d threads: %s): stage 1 %.2f s = %.3e rows/s, %.3e records/s, %d survivors"
          % (T, cpu_line["what"], cpu_line["seconds"], cpu_line["rows_per_s"], cpu_line["records_per_s"], cpu_line["survivors"]), flush=True)
configs = (["-t", "1"], ["-t", "8"], ["-t", "16"], ["-t", "64"], ["-t", "256"], ["-t", "64", "--devices", "2"])
if a.only:
    configs = (a.only.split(),)
preload = {}
if a.ab and a.ab_preload:
    configs = [["-t", "16"] + m for _ in range(a.ab) for m in ([], ["#preload"])]
    preload = {"LD_PRELOAD": os.path.abspath(a.ab_preload)}
elif a.ab:
    configs = [c + m for _ in range(a.ab) for c in (["-t", "16"], ["-t", "64"]) for m in ([], ["--raw-transfer"])]
gpu_lines = []
for extra in configs:
    out = os.path.join(root, "out")
    shutil.rmtree(out, ignore_errors=True)
    t0 = time.time()
    env = dict(os.environ, KMD_HOST_TIMING="1")
    if "#preload" in extra:
        env.update(preload)
    r = subprocess.run([cli, "diff", "-d", os.path.join(root, "km"), "-1", str(a.nc), "-2", str(a.nk), "-o", out] + [x for x in extra if x != "#preload"],
                       capture_output=True, text=True, env=env)
    dt = time.time() - t0
    assert r.returncode == 0, r.stderr
    stage1 = [l for l in r.stderr.split("\n") if "Partitions processed" in l][0].split("(")[1].split(" s")[0]
    for l in r.stderr.split("\n"):
        if "waited" in l or "steady state" in l or "last partition" in l or "worker ready" in l or "workers done" in l or "Done in" in l:
            print("   ", l)
    try:
        import json
        tr = json.load(open(os.path.join(out, "summary.json"))).get("transfer")
        if tr:
            print("    transfer: %s, %.3f bytes per record across the link" % (tr["format"], tr["bytes_per_record"]))
    except Exception:
        pass
    print("kmdiff-hip diff %-24s total %.2f s, stage 1 %s s = %.3e rows/s, %.3e records/s%s"
          % (" ".join(extra).replace("#preload", "(LD_PRELOAD " + os.path.basename(os.path.dirname(preload.get("LD_PRELOAD", "/x/y"))) + ")"), dt, stage1, a.parts * a.rows / float(stage1), records / float(stage1),
             "  (%.1f x the CPU baseline's stage 1)" % (cpu_line["seconds"] / float(stage1)) if cpu_line else ""), flush=True)
    gpu_lines.append({"flags": " ".join(extra), "total_s": dt, "stage1_s": float(stage1), "rows_per_s": a.parts * a.rows / float(stage1),
                      "records_per_s": records / float(stage1)})
if a.json:
    import json
    with open(a.json, "w") as f:
        json.dump({"run_dir": {"partitions": a.parts, "rows_per_partition": a.rows, "samples": S, "records": records, "bytes_on_disk": size},
                   "host_threads_available": cpu_quota(), "kmdiff_hip_diff": gpu_lines, "cpu_baseline_e2e": cpu_line}, f)
shutil.rmtree(root, ignore_errors=True)
